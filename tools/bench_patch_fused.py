"""The tubelet embedding at the production shape: the unfused chain (gather + LayerNorm kernel -> [tokens, 4000] bf16 operand ->
K = 4000 GEMM) against ctclip_patch_embed_fused (one pass over the volume, csrc/patch_gemm.hip), forward; and the weight-gradient
product from the materialised operand against ctclip_patch_wgrad_fused (operand recomputed from the volume).   B=32 python3 ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ct-clip-ut_amd"))
import torch
from ctclip_hip.lib import hip
from ctclip_hip import ops
B = int(os.environ.get("B", 32))
dev = "cuda"
vol = (torch.randn(B, 1, 240, 480, 480, device=dev) * 0.5).clamp_(-1, 1).to(torch.bfloat16)
F_, N, M = 4000, 512, B * 24 * 24 * 24
Wg = (torch.randn(N, F_, device=dev) * F_ ** -0.5).to(torch.bfloat16)
bfold = torch.randn(N, device=dev) * 0.05
wsum = Wg.float().sum(1).contiguous()
Z = torch.empty(M, N, device=dev)
mean, rstd = (torch.empty(M, device=dev) for _ in range(2))
tstat = torch.empty(M, 4, device=dev)


def timeit(fn, n=5):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


flops = 2.0 * M * N * F_
alg = vol.numel() * 2 + Wg.numel() * 2 + M * N * 4
A = torch.empty(M, F_, device=dev, dtype=torch.bfloat16)


def chain():
    hip.patch_ln_fwd(vol, 1, None, None, A, mean, rstd, B, 1, 240, 480, 480, 10, 20, F_, 1e-5)
    hip.gemm_bf16(A, Wg, Z, bfold, None, M, N, F_, F_, F_, N, 0, 1, 1, 1, 1, 0, 1.0, 0)


def fused():
    hip.patch_embed_fused(vol, Wg, F_, wsum, bfold, Z, N, tstat, B, 1, 240, 480, 480, 10, 20, N, 1e-5)


for rep in range(2):
    t = timeit(chain)
    print(f"forward, unfused chain (gather + LN -> operand -> GEMM), {B} pairs: {t:9.1f} us  {flops / t / 1e6:7.1f} TFLOP/s  "
          f"{alg / t / 1e3:6.2f} GB/s-algorithmic x1e-3")
    t = timeit(fused)
    print(f"forward, fused (one pass over the volume),               {B} pairs: {t:9.1f} us  {flops / t / 1e6:7.1f} TFLOP/s  "
          f"{alg / t / 1e3:6.2f} GB/s-algorithmic x1e-3")
dz = torch.randn(M, N, device=dev).to(torch.bfloat16)
G = torch.zeros(N, F_, device=dev)
chain()
fused()
Gx = torch.zeros(N, F_ + 2, device=dev)
for rep in range(2):
    t = timeit(lambda: ops.wgrad(dz, A, N, F_, M, out=G))
    print(f"weight gradient from the materialised operand,       {B} pairs: {t:9.1f} us  {flops / t / 1e6:7.1f} TFLOP/s")
    t = timeit(lambda: hip.patch_wgrad_fused(vol, dz, N, tstat, Gx, F_ + 2, B, 1, 240, 480, 480, 10, 20, N))
    print(f"weight gradient, operand recomputed from the volume, {B} pairs: {t:9.1f} us  {flops / t / 1e6:7.1f} TFLOP/s")
