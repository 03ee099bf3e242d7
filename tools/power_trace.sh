#!/bin/bash
# Samples board power, clocks and temperature (rocm-smi, read-only) while the default benchmark's training steps run.
#   bash tools/power_trace.sh > gpurun_out/power_trace.txt
python3 bench.py --lean --steps 40 --warmup 5 > /tmp/power_bench.json 2>/dev/null &
BP=$!
sleep 12                                   # model build + first import
for i in $(seq 1 40); do
  rocm-smi --showpower --showclocks --showtemp --showuse 2>/dev/null | grep -E "Power|sclk|mclk|Temperature \(Sensor (edge|junction|hotspot)|GPU use" | tr -s ' ' | paste -sd';'
  sleep 0.5
done
wait $BP
python3 tools/show_bench.py /tmp/power_bench.json | head -1
