import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "ct-clip-ut_amd"))
import torch
from ctclip_hip.lib import hip
B = 64; dev = "cuda"
def timeit(fn, n=5):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, nseq, H in (("token-major, 8 heads interleaved", 24 * B, 8), ("one head per row (timing stand-in for head-major)", 24 * B * 8, 1)):
    n, D, gh, gw = 576, 32, 24, 24
    ld = H * D
    q, k, v, do = (torch.nn.functional.normalize(torch.randn(nseq * n, H, D, device=dev), dim=-1).reshape(nseq * n, ld).to(torch.bfloat16) for _ in range(4))
    bias = torch.randn(H, n, n, device=dev)
    o = torch.empty_like(q); lse = torch.empty(nseq, H, n, device=dev)
    dq, dk, dv = (torch.empty_like(q) for _ in range(3)); delta = torch.empty_like(lse)
    R = (2 * gh - 1) * (2 * gw - 1)
    dt = torch.zeros(H, R, device=dev)
    fwd = lambda: hip.attn_fwd(q, k, v, o, lse, bias, None, nseq, n, H, D, ld, ld, ld, ld, 1.0)
    bwd = lambda table: hip.attn_bwd(q, k, v, o, do, lse, delta, dq, dk, dv, bias, None, None, None, dt if table else None, R if table else 0,
                                     gh if table else 0, gw if table else 0, nseq, n, H, D, ld, ld, ld, ld, ld, ld, ld, ld, 1.0)
    print(f"{name}: fwd {timeit(fwd):8.1f} us   bwd (no dbias) {timeit(lambda: bwd(False)):8.1f} us   bwd (+table) {timeit(lambda: bwd(True)):8.1f} us", flush=True)
    del q, k, v, do, o, dq, dk, dv
