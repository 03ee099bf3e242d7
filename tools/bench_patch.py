"""Micro-benchmark of the tubelet gather + LayerNorm kernels at the production shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ct-clip-ut_amd"))
import torch
from ctclip_hip.lib import hip
B = int(os.environ.get("B", 16))
vol = (torch.randn(B, 1, 240, 480, 480, device="cuda") * 0.5).clamp_(-1, 1).to(torch.bfloat16)
F_, M = 4000, B * 24 * 24 * 24
gm, bt = torch.ones(F_, device="cuda"), torch.zeros(F_, device="cuda")
A = torch.empty(M, F_, device="cuda", dtype=torch.bfloat16)
mean, rstd = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
dA = torch.randn(M, F_, device="cuda").to(torch.bfloat16)
dg, db = torch.zeros(F_, device="cuda"), torch.zeros(F_, device="cuda")
def timeit(fn, n=5):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
gb = (vol.numel() * 2 + A.numel() * 2) / 1e9
t = timeit(lambda: hip.patch_ln_fwd(vol, 1, gm, bt, A, mean, rstd, B, 1, 240, 480, 480, 10, 20, F_, 1e-5))
print(f"patch_ln_fwd {t:9.1f} us  {gb / t * 1e6 / 1e3:5.2f} TB/s")
if os.environ.get("NULL_AFFINE", "1") == "1":
    try:
        t = timeit(lambda: hip.patch_ln_fwd(vol, 1, None, None, A, mean, rstd, B, 1, 240, 480, 480, 10, 20, F_, 1e-5))
        print(f"patch_ln_fwd (no affine: folded into the projection) {t:9.1f} us  {gb / t * 1e6 / 1e3:5.2f} TB/s")
    except Exception as e:                                     # an older library build
        print("patch_ln_fwd without affine: not supported by this build", type(e).__name__)
G, W = torch.randn(512, F_, device="cuda"), torch.randn(512, F_, device="cuda")
dW, dbv = torch.zeros(512, F_, device="cuda"), torch.randn(512, device="cuda")
t = timeit(lambda: hip.patch_affine_bwd(G, dbv, W, gm, bt, dW, dg, db, 512, F_, F_, 0))
print(f"patch_affine_bwd {t:9.1f} us (folded d(W) / d(gamma) / d(beta) of LayerNorm(4000): no pass over the tokens)")
