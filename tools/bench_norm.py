"""Micro-benchmark of the LayerNorm kernels at the CT-ViT shape (884 736 x 512 at 64 pairs)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ct-clip-ut_amd"))
import torch
from ctclip_hip.lib import hip
B = int(os.environ.get("B", 64))
M, dim = 13824 * B, 512
dev = "cuda"
x = torch.randn(M, dim, device=dev); gamma = torch.ones(dim, device=dev); beta = torch.zeros(dim, device=dev)
mean = x.mean(-1).contiguous(); rstd = (x.var(-1, unbiased=False) + 1e-5).rsqrt().contiguous()
dy16 = torch.randn(M, dim, device=dev).to(torch.bfloat16); dres = torch.randn(M, dim, device=dev)
dres2 = torch.randn(M, dim, device=dev).to(torch.bfloat16)
dx = torch.empty(M, dim, device=dev); dx16 = torch.empty(M, dim, device=dev, dtype=torch.bfloat16)
dg = torch.zeros(dim, device=dev); db = torch.zeros(dim, device=dev)
def timeit(fn, n=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
gb = M * dim / 1e9
t = timeit(lambda: hip.layernorm_bwd_bf16(dy16, x, gamma, mean, rstd, dres, dres2, dx, None, dg, None, M, dim))
print(f"layernorm_bwd_bf16 (dy16, x, dres, dres2 -> dx)        {t*1e3:8.1f} us  {gb * (2 + 4 + 4 + 2 + 4) / t:6.2f} TB/s")
t = timeit(lambda: hip.layernorm_bwd_bf16(dy16, x, gamma, mean, rstd, dres, None, dx, dx16, dg, db, M, dim))
print(f"layernorm_bwd_bf16 (dy16, x, dres -> dx, dx16)         {t*1e3:8.1f} us  {gb * (2 + 4 + 4 + 4 + 2) / t:6.2f} TB/s")
