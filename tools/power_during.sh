#!/bin/bash
# rocm-smi power / sclk samples while a command runs:  bash tools/power_during.sh <delay_s> <samples> <cmd...>
DELAY=$1; N=$2; shift 2
"$@" > /tmp/power_cmd.out 2>/dev/null &
BP=$!
sleep $DELAY
for i in $(seq 1 $N); do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk" | sed 's/.*: //' | paste -sd' '
  sleep 0.3
done
wait $BP
tail -3 /tmp/power_cmd.out
