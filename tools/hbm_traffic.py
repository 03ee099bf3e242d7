"""HBM-side bytes per launch from two rocprofv3 PMC passes of the same command (one with --pmc FETCH_SIZE, one with
--pmc WRITE_SIZE; counters are collected in their own runs, with --kernel-trace only).
bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: gfx950 tallies the 128-byte read requests of wide streaming loads at 64 B
(MI355X_MICROARCH.md, HBM section), so the read side is doubled.
usage: hbm_traffic.py FETCH_counter_collection.csv WRITE_counter_collection.csv OUT.csv ["command line for the header"]"""
import collections, csv, re, sys

def per_kernel(path, counter):
    tot, n = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        k = k.replace("(anonymous namespace)::", "").replace("void ", "")
        k = re.sub(r"\(.*$", "", k)[:90]
        tot[k] += float(r["Counter_Value"])
        n[k].add(r["Dispatch_Id"])
    return tot, {k: len(v) for k, v in n.items()}

f, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
w, nw = per_kernel(sys.argv[2], "WRITE_SIZE")
rows = []
for k in f:
    if k not in w or nf[k] == 0 or nw[k] == 0:
        continue
    fm = 2.0 * f[k] / nf[k] * 1024 / 1e6
    wm = w[k] / nw[k] * 1024 / 1e6
    rows.append((nf[k] * (fm + wm), k, nf[k], fm, wm))
rows.sort(reverse=True)
with open(sys.argv[3], "w") as out:
    out.write(f"# HBM-side traffic per launch from rocprofv3 PMC (separate FETCH_SIZE and WRITE_SIZE passes, {sys.argv[4] if len(sys.argv) > 4 else ''})\n")
    out.write("# bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: gfx950 tallies 128-byte read requests at 64 B (MI355X_MICROARCH.md, HBM section)\n")
    out.write("kernel,launches,fetch_MB_per_launch(corrected),write_MB_per_launch,total_MB_per_launch\n")
    for _, k, n, fm, wm in rows[:40]:
        out.write(f"{k},{n},{fm:.1f},{wm:.1f},{fm + wm:.1f}\n")
print(open(sys.argv[3]).read()[:3000])
