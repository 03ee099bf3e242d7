"""Where do the framework's own small kernels (fills, copies, element-wise adds) come from?  Reads a rocprofv3
*_kernel_trace.csv and groups every dispatch whose name matches PATTERN by (name, grid size, previous kernel, next
kernel).  usage: trace_glue.py KERNEL_TRACE_CSV STEPS [PATTERN]"""
import csv, re, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
pat = re.compile(sys.argv[3] if len(sys.argv) > 3 else r"at::native|rocclr")
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "").replace("at::native::", "")
    return n[:60]
agg = defaultdict(lambda: [0, 0])
for i, r in enumerate(rows):
    if not pat.search(r["Kernel_Name"]):
        continue
    grid = int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)
    prev = short(rows[i - 1]["Kernel_Name"]) if i else ""
    nxt = short(rows[i + 1]["Kernel_Name"]) if i + 1 < len(rows) else ""
    a = agg[(short(r["Kernel_Name"]), grid, prev, nxt)]
    a[0] += 1
    a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, (n, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{ns/1e6/steps:7.3f} ms/step {n/steps:6.1f} calls/step grid={k[1]:>10d}  {k[0]}\n          after {k[2]}\n          before {k[3]}")
