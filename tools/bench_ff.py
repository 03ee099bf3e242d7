"""Micro-benchmark of the fused feed-forward products (gemm3 epilogues 1, 2, 3) at the CT-ViT shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ct-clip-ut_amd"))
import torch
from ctclip_hip.lib import hip
B = int(os.environ.get("B", 32))
M, dim, Ip = 13824 * B, 512, 1408
dev = "cuda"
bf = torch.bfloat16
x = torch.randn(M, dim, device=dev).to(bf)
w1 = (torch.randn(2 * Ip, dim, device=dev) * 0.04).to(bf)
w2 = (torch.randn(dim, Ip, device=dev) * 0.04).to(bf)
w2T = w2.t().contiguous()
h = torch.empty(M, 2 * Ip, device=dev, dtype=bf)
g = torch.empty(M, Ip, device=dev, dtype=bf)
resid = torch.randn(M, dim, device=dev)
y = torch.empty(M, dim, device=dev)
dy = torch.randn(M, dim, device=dev).to(bf)
def timeit(fn, n=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
ff1 = lambda: hip.gemm_bf16_geglu(x, w1, h, g, M, Ip, dim, dim, dim, 2 * Ip, Ip)
ff2 = lambda: hip.gemm_bf16(g, w2, y, None, resid, M, dim, Ip, Ip, Ip, dim, dim, 1, 1, 1, 1, 0, 1.0, 0)
ff2b = lambda: hip.gemm_bf16_geglu_bwd(dy, w2T, h, None, M, Ip, dim, dim, dim, 2 * Ip, Ip)
for name, fn, flops, gbytes in (("ff1 + geglu (epi 2)", ff1, 2.0 * M * 2 * Ip * dim, (M * dim * 2 + M * 3 * Ip * 2) / 1e9),
                                ("ff2 f32 + resid (epi 1)", ff2, 2.0 * M * dim * Ip, (M * Ip * 2 + 2 * M * dim * 4) / 1e9),
                                ("ff2 dgrad + geglu bwd (epi 3)", ff2b, 2.0 * M * Ip * dim, (M * dim * 2 + 2 * M * 2 * Ip * 2) / 1e9)):
    ms = timeit(fn)
    print(f"{name:32s} {ms*1e3:9.1f} us  {flops/ms/1e9:7.1f} TFLOP/s  {gbytes/ms:6.2f} TB/s algorithmic", flush=True)
