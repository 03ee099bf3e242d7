"""csrc/gemm5.hip through its direct entry (ctclip_gemm5_bf16) on the step's long-K / wide-N shapes.   B=32 python3 tools/bench_gemm5.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ct-clip-ut_amd"))
import torch
from ctclip_hip.lib import hip
B = int(os.environ.get("B", 32))
REP = int(os.environ.get("REP", 5))
ONLY = os.environ.get("ONLY")
T = 13824 * B
dev = "cuda"
shapes = [("sq4096", 4096, 4096, 4096, 0, 0), ("ff1 fwd", T, 2816, 512, 0, 0), ("ff1 + geglu", T, 2816, 512, 0, 2),
          ("ff2 fwd f32", T, 512, 1408, 1, 0), ("ff1 dgrad f32", T, 512, 2816, 1, 0), ("kv fwd", T, 512, 512, 0, 0),
          ("ff2 dgrad+geglu'", T, 1408, 512, 0, 3)]
for name, M, N, K, cf, act in shapes:
    if ONLY and not any(o in name for o in ONLY.split(",")):
        continue
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    Bm = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    C = torch.empty(M, N, device=dev, dtype=torch.float32 if cf else torch.bfloat16)
    G = torch.empty(M, N // 2, device=dev, dtype=torch.bfloat16) if act == 2 else (torch.randn(M, 2 * N, device=dev).to(torch.bfloat16) if act == 3 else None)
    fn = lambda: hip.gemm5_bf16(A, Bm, C, None, None, M, N, K, K, K, N, 0, cf, 1.0, act, G, N // 2 if act == 2 else (2 * N if act == 3 else 0))
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REP):
        fn()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / REP * 1e3
    print(f"{name:16s} M={M:8d} N={N:5d} K={K:5d}  {t:9.1f} us  {2.0 * M * N * K / t / 1e6:8.1f} TFLOP/s")
    del A, Bm, C, G
