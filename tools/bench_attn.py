"""Micro-benchmark of the fused attention kernels at the CT-ViT spatial / temporal shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ct-clip-ut_amd"))
import torch
from ctclip_hip.lib import hip
B = int(os.environ.get("B", 8))
dev = "cuda"
def timeit(fn, n=5):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, nseq, n, H, D, gh, gw in (("spatial", 24 * B, 576, 8, 32, 24, 24), ("temporal", 576 * B, 24, 8, 32, 0, 0)):
    ld = H * D
    q, k, v, do = (torch.nn.functional.normalize(torch.randn(nseq * n, H, D, device=dev), dim=-1).reshape(nseq * n, ld).to(torch.bfloat16) for _ in range(4))
    q = (q.float() * 8).to(torch.bfloat16)
    bias = torch.randn(H, n, n, device=dev) if gh else None
    o = torch.empty_like(q); lse = torch.empty(nseq, H, n, device=dev)
    dq, dk, dv = (torch.empty_like(q) for _ in range(3)); delta = torch.empty_like(lse)
    R = (2 * gh - 1) * (2 * gw - 1) if gh else 0
    dt = torch.zeros(H, max(R, 1), device=dev)
    fwd = lambda: hip.attn_fwd(q, k, v, o, lse, bias, None, nseq, n, H, D, ld, ld, ld, ld, 1.0)
    def bwd(table):
        hip.attn_bwd(q, k, v, o, do, lse, delta, dq, dk, dv, bias, None, None, None, dt if table else None, R if table else 0,
                     gh if table else 0, gw if table else 0, nseq, n, H, D, ld, ld, ld, ld, ld, ld, ld, ld, 1.0)
    flops = 4.0 * nseq * H * n * n * D
    t = timeit(fwd); print(f"{name:9s} fwd            {t:9.1f} us  {flops / t / 1e6:7.1f} TFLOP/s")
    t = timeit(lambda: bwd(False)); print(f"{name:9s} bwd (no dbias) {t:9.1f} us  {2.5 * flops / t / 1e6:7.1f} TFLOP/s")
    if gh:
        keep = bias
        bias = None
        t = timeit(fwd); print(f"{name:9s} fwd  (bias=None)      {t:9.1f} us")
        t = timeit(lambda: bwd(False)); print(f"{name:9s} bwd  (bias=None)      {t:9.1f} us")
        bias = keep
        t = timeit(lambda: bwd(True)); print(f"{name:9s} bwd (+table)   {t:9.1f} us  {2.5 * flops / t / 1e6:7.1f} TFLOP/s")
