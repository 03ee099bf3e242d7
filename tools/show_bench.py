"""Print the headline fields of a bench.py JSON line.  usage: show_bench.py FILE"""
import json, sys
d = json.load(open(sys.argv[1]))
c = d["config"]
print(f"{d['value']:.2f} {d['unit']}  {d['ms_per_step']:.2f} ms/step (median {d['ms_per_step_median']:.2f})  batch {c['per_gpu_batch']}  "
      f"peak {c['peak_hbm_gib']} GiB  loss {c['final_loss']:.4f}")
r = d["roofline"]
print(f"GEMM family: {r['achieved']:.0f} TFLOP/s = {r['frac']:.3f} of peak, {r['gemm_ms_per_step']:.1f} ms/step, "
      f"{r['per_shape_bound']['frac_of_bound']:.2f} of its per-shape bound; event timing overhead {r['event_timing']['overhead_frac']:+.4f}")
if r.get("step_hbm"):
    print("step HBM:", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in r["step_hbm"].items() if k != "what"})
for k, v in r["kernels"].items():
    keys = ("ms_per_step", "ms_forward_chain", "achieved", "unit", "frac", "traffic", "algorithmic_bytes_per_launch")
    print(f"  {k}: " + ", ".join(f"{a}={v[a]:.4g}" if isinstance(v[a], float) else f"{a}={v[a]}" for a in keys if a in v and v[a] is not None))
if "attribution" in d:
    a = d["attribution"]
    print(f"attribution: occlusion {a['occlusion']['value']:.1f} windows/s (frac {a['occlusion']['roofline']['frac']:.3f}), "
          f"IG {a['integrated_gradients']['value']:.1f} points/s")
if "cpu_baseline" in d:
    b = d["cpu_baseline"]
    print(f"cpu_baseline: {b['value']:.4f} {b['unit']} on {b['cores']} cores ({b['kind']}); config1 {b['config1']['value']:.1f} pairs/s")
