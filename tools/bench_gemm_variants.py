"""Run tools/bench_gemm.py under several gemm3 variants (env switches), one subprocess each, and tabulate TFLOP/s."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIAG = os.path.join(ROOT, "ct-clip-ut_amd", "ctclip_hip", "libctclip_hip_diag.so")
variants = [
    ("4 stages", {}),
    ("5 stages (160 KiB)", {"CTCLIP_HIP_LIB": DIAG}),
    ("4 stages (2)", {}),
    ("5 stages (2)", {"CTCLIP_HIP_LIB": DIAG}),
]
sel = os.environ.get("VARIANTS")
if sel:
    variants = [v for i, v in enumerate(variants) if str(i) in sel.split(",")]
only = os.environ.get("ONLY", "sq4096,ff1 fwd,ff2 fwd,kv fwd,q fwd,out fwd,ff2 dgrad,ff1 dgrad")
rows = {}
for name, env in variants:
    e = dict(os.environ, ONLY=only, B=os.environ.get("B", "32"), **env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_gemm.py")], env=e, capture_output=True, text=True)
    if out.returncode != 0:
        print(name, "FAILED", out.stderr[-500:], flush=True)
        continue
    for line in out.stdout.splitlines():
        if "TFLOP/s" in line:
            shape = line[:20].strip()
            rows.setdefault(shape, {})[name] = float(line.split()[-2])
    print(f"# {name} done", flush=True)
names = [n for n, _ in variants]
print("shape".ljust(16) + "".join(n[:24].rjust(26) for n in names))
for shape, r in rows.items():
    print(shape.ljust(16) + "".join((f"{r[n]:.0f}" if n in r else "-").rjust(26) for n in names))
