"""Micro-benchmark of the PEG kernels at the CT-ViT shape (b, 24, 24, 24, 512)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ct-clip-ut_amd"))
import torch
from ctclip_hip.lib import hip
B = int(os.environ.get("B", 32))
dev = "cuda"
def timeit(fn, n=5):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
T = H = W = 24; d = 512
x = torch.randn(B, T, H, W, d, device=dev)
dy = torch.randn_like(x)
w27 = torch.randn(27, d, device=dev) * 0.1
bias = torch.randn(d, device=dev)
y = torch.empty_like(x); y16 = torch.empty(x.shape, device=dev, dtype=torch.bfloat16)
dw = torch.zeros(27, d, device=dev); db = torch.zeros(d, device=dev)
gb = x.numel() * 4 / 1e9
t = timeit(lambda: hip.peg_fwd(x, w27, bias, y, y16, B, T, H, W, d, 1)); print(f"peg_fwd        {t:9.1f} us  {(2.5 * gb) / t * 1e6 / 1e3:6.2f} TB/s (x read + y f32 + y bf16)")
t = timeit(lambda: hip.peg_bwd_data(dy, w27, y, y16, B, T, H, W, d, 1)); print(f"peg_bwd_data   {t:9.1f} us  {(2.5 * gb) / t * 1e6 / 1e3:6.2f} TB/s")
t = timeit(lambda: hip.peg_bwd_weight(dy, x, dw, db, B, T, H, W, d)); print(f"peg_bwd_weight {t:9.1f} us  {(2.0 * gb) / t * 1e6 / 1e3:6.2f} TB/s (x + dy read)")
