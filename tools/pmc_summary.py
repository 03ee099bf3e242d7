"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel (mean per dispatch).  usage: pmc_summary.py CSV [filter]"""
import csv, collections, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
def short(n):
    m = re.search(r"(\w+_kernel\w*)(<[^>]*>)?", n)
    return (m.group(1) + (m.group(2) or "")) if m else n[:60]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for r in rows:
    k = short(r["Kernel_Name"])
    if flt and flt not in k:
        continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[k].add(r["Dispatch_Id"])
for k, v in agg.items():
    n = len(disp[k])
    print(f"{k}  ({n} dispatches)")
    for c, x in sorted(v.items()):
        print(f"   {c:24s} {x / n:12.4g}")
