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

# the same spatial shape on head-major operands (csrc/attention_hm.hip): log2-domain logits, static shift, f16 bias MFMA
if True:
    nseq, n, H, D, gh, gw = 24 * B, 576, 8, 32, 24, 24
    LOG2E = 1.4426950408889634
    ld = H * D
    unit = lambda: torch.nn.functional.normalize(torch.randn(nseq, H, n, D, device=dev), dim=-1)
    q, k = (unit() * 8 * LOG2E).to(torch.bfloat16), unit().to(torch.bfloat16)
    v, do = (torch.randn(nseq, H, n, D, device=dev).to(torch.bfloat16) for _ in range(2))
    bias = torch.randn(H, n, n, device=dev)
    o = torch.empty(nseq * n, ld, device=dev, dtype=torch.bfloat16); lse = torch.empty(nseq, H, n, device=dev)
    dq, dk, dv = (torch.empty_like(o) for _ in range(3)); delta = torch.empty_like(lse)
    R = (2 * gh - 1) * (2 * gw - 1)
    dt = torch.zeros(H, R, device=dev)
    ones = torch.ones(D, device=dev)
    shift = torch.empty(H + 1, device=dev)
    hip.attn_shift(ones, ones, D, 8 * LOG2E, bias, n * n, n * n, 1, H, shift)
    flops = 4.0 * nseq * H * n * n * D
    for label, sh in (("static shift", shift), ("online softmax", None)):
        t = timeit(lambda: hip.attn_hm_fwd(q, k, v, o, lse, bias, sh, nseq, n, H, ld))
        print(f"head-major fwd ({label:14s}) {t:9.1f} us  {flops / t / 1e6:7.1f} TFLOP/s")
    t = timeit(lambda: hip.attn_hm_fwd(q, k, v, o, lse, None, None, nseq, n, H, ld))
    print(f"head-major fwd (bias=None, online)  {t:9.1f} us")
    hip.attn_hm_fwd(q, k, v, o, lse, bias, shift, nseq, n, H, ld)
    def bwd(table, b=bias):
        hip.attn_hm_bwd(q, k, v, o, do, lse, delta, dq, dk, dv, b, None, None, dt if table else None, R if table else 0,
                        gh if table else 0, gw if table else 0, nseq, n, H, ld, ld, ld, ld)
    t = timeit(lambda: bwd(False)); print(f"head-major bwd (no dbias) {t:9.1f} us  {2.5 * flops / t / 1e6:7.1f} TFLOP/s")
    t = timeit(lambda: bwd(True)); print(f"head-major bwd (+table)   {t:9.1f} us  {2.5 * flops / t / 1e6:7.1f} TFLOP/s")
    t = timeit(lambda: bwd(False, None)); print(f"head-major bwd (bias=None) {t:9.1f} us")
