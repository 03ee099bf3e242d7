"""GPU idle time inside the timed steps of a rocprofv3 *_kernel_trace.csv: union of all kernels' busy intervals against
the span they cover, plus the largest gaps and what ran around them.  usage: trace_idle.py CSV [SKIP_FRACTION]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5      # ignore the first half (build, warm-up, data synthesis)
t0, t1 = int(rows[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows)
lo = t0 + (t1 - t0) * skip
rows = [r for r in rows if int(r["Start_Timestamp"]) >= lo]
def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").replace("at::native::", "")[:50]
busy, gaps, end, prev = 0, [], None, None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if end is None:
        end = s
    if s > end:
        gaps.append((s - end, short(prev["Kernel_Name"]), short(r["Kernel_Name"])))
        busy += e - s
        end = e
    elif e > end:
        busy += e - end
        end = e
    if prev is None or e >= int(prev["End_Timestamp"]):
        prev = r
span = end - int(rows[0]["Start_Timestamp"])
print(f"span {span/1e6:.1f} ms, busy {busy/1e6:.1f} ms, idle {(span-busy)/1e6:.1f} ms = {100*(span-busy)/span:.1f} %  ({len(gaps)} gaps)")
from collections import defaultdict
agg = defaultdict(lambda: [0, 0])
for g, a, b in gaps:
    agg[(a, b)][0] += 1; agg[(a, b)][1] += g
for (a, b), (n, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"  {ns/1e6:7.2f} ms in {n:5d} gaps  after {a}  before {b}")
