#!/bin/bash
# What the vendor library launches for the step's GEMM shapes, next to this repo's kernels: kernel names (Tensile names
# encode macro tile / MFMA / wave grid / LDS use), registers, LDS, grid, duration -- and the effective shader clock of each
# launch (GRBM_GUI_ACTIVE / 8 XCDs / duration; guide 'DVFS give-back').   usage (GPU box, repo root): bash tools/vendor_probe.sh
set -e -o pipefail
OUT=$PWD/gpurun_out
REPO=$PWD
export B=${B:-32} R=2
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_vp
rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_vp/kt -o kt -- python3 $REPO/tools/gemm_vs_vendor.py > $OUT/vp_kt.txt 2> $OUT/vp_kt.err
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/prof_vp/pmc -o pmc -- python3 $REPO/tools/gemm_vs_vendor.py > $OUT/vp_pmc.txt 2> $OUT/vp_pmc.err
python3 - $(find $OUT/prof_vp/kt -name '*kernel_trace.csv' | head -1) $(find $OUT/prof_vp/pmc -name '*counter_collection.csv' | head -1) <<'PY'
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    k = r["Kernel_Name"]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg.setdefault(k, {"n": 0, "us": [], "r": r})
    a["n"] += 1; a["us"].append(d)
print("# kernel-trace: name | launches | median us | workgroup | grid | LDS | VGPR | AGPR | SGPR")
for k, a in agg.items():
    r = a["r"]; us = sorted(a["us"])
    if us[len(us) // 2] < 20: continue
    print(f"{k[:150]} | {a['n']} | {us[len(us)//2]:.1f} | {r.get('Workgroup_Size_X')} | {r.get('Grid_Size_X')}x{r.get('Grid_Size_Y')} | "
          f"{r.get('LDS_Block_Size')} | {r.get('VGPR_Count')} | {r.get('Accum_VGPR_Count')} | {r.get('SGPR_Count')}")
print("# effective clock = GRBM_GUI_ACTIVE / 8 / duration (reads high on launches < 0.3 ms)")
c = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[2])):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE": continue
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if d < 100: continue
    a = c.setdefault(r["Kernel_Name"], [])
    a.append(float(r["Counter_Value"]) / 8 / d)          # cycles per us = MHz
for k, v in c.items():
    v = sorted(v)
    print(f"{k[:110]} | launches {len(v)} | clock MHz median {v[len(v)//2]:.0f} min {v[0]:.0f} max {v[-1]:.0f}")
PY
rm -rf $OUT/prof_vp
