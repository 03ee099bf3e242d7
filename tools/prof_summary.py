"""Summarise a rocprofv3 *_kernel_stats.csv: per-step milliseconds per kernel.  usage: prof_summary.py CSV STEPS [OUT]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return (n[:n.index("(")] if "(" in n else n)[:72]
tot = sum(int(r["TotalDurationNs"]) for r in rows)
lines = [f"# total kernel time {tot/1e6/steps:.2f} ms/step over {steps:.0f} steps", "kernel,calls,ms_per_step,avg_us,pct"]
for r in rows:
    lines.append(f"{short(r['Name'])},{r['Calls']},{int(r['TotalDurationNs'])/1e6/steps:.3f},{float(r['AverageNs'])/1e3:.1f},{r['Percentage']}")
out = "\n".join(lines)
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write(out + "\n")
print("\n".join(lines[:int(sys.argv[4]) if len(sys.argv) > 4 else 40]))
