"""Throughput of the batched occlusion-sensitivity scan at the production shape: windows (= forwards)
per second on one MI355X; the reference does one B=1 forward per window (~10 forwards/s)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ct-clip-ut_amd"))
import torch
import bench
from utils.visualizations import Visualizations

class Acc:
    is_main_process, process_index, num_processes, device = True, 0, 1, torch.device("cuda")

model = bench.build_model(dict(bench.VIT), dict(bench.TEXT)).cuda().eval()
vol, txt = bench.synthetic_batch(1, 240, 480, 128, bench.TEXT["vocab_size"], torch.device("cuda"), 0)
vol = vol.float()
n = int(os.environ.get("WINDOWS", 512))
for batch in (int(b) for b in os.environ.get("BATCHES", "32,64").split(",")):
    vis = Visualizations(model, Acc(), occlusion_batch=batch, max_windows=n)
    vis.maybe_print = lambda *a, **k: None
    vis._compute_occlusion(vol, txt, None, (20, 40, 40), (10, 20, 20), 0.0)      # warm-up (shadows, allocator)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    vis._compute_occlusion(vol, txt, None, (20, 40, 40), (10, 20, 20), 0.0)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"occlusion_batch={batch}: {n} windows in {dt:.2f} s = {n / dt:.1f} windows/s "
          f"(full 12167-window scan: {12167 / (n / dt):.0f} s); peak HBM {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)

# integrated gradients (reference visualizations.py:851-910): 50 interpolation points, forward + backward to the volume
for ig_batch in (10, 25):
    vis = Visualizations(model, Acc())
    vis._integrated_gradients(vol, txt, steps=ig_batch, ig_batch=ig_batch)        # warm-up
    torch.cuda.synchronize(); t0 = time.perf_counter()
    vis._integrated_gradients(vol, txt, steps=50, ig_batch=ig_batch)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"integrated gradients, 50 steps, ig_batch={ig_batch}: {dt:.2f} s ({50 / dt:.1f} fwd+bwd points/s); "
          f"peak HBM {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
