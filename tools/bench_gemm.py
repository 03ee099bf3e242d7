"""Micro-benchmark of ctclip_gemm_bf16 on the shapes of the CT-CLIP step (per-GPU batch B) + square references."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ct-clip-ut_amd"))
import torch
from ctclip_hip.lib import hip

B = int(os.environ.get("B", 8))
T = 13824 * B
shapes = [  # name, M, N, K, akm, bkm, c_fp32, split, accumulate   (I = 1365, padded Ip = 1408)
    ("sq4096", 4096, 4096, 4096, 1, 1, 0, 1, 0),
    ("sq8192", 8192, 8192, 8192, 1, 1, 0, 1, 0),
    ("ff1 fwd", T, 2816, 512, 1, 1, 0, 1, 0),
    ("ff2 fwd f32", T, 512, 1408, 1, 1, 1, 1, 0),
    ("q fwd", T, 256, 512, 1, 1, 0, 1, 0),
    ("kv fwd", T, 512, 512, 1, 1, 0, 1, 0),
    ("out fwd f32", T, 512, 256, 1, 1, 1, 1, 0),
    ("patch fwd f32", T, 512, 4000, 1, 1, 1, 1, 0),
    ("ff2 dgrad", T, 1408, 512, 1, 1, 0, 1, 0),
    ("ff1 dgrad f32", T, 512, 2816, 1, 1, 1, 1, 0),
    ("ff1 dgrad bf16", T, 512, 2816, 1, 1, 0, 1, 0),
    ("out dgrad", T, 256, 512, 1, 1, 0, 1, 0),
    ("q dgrad f32", T, 512, 256, 1, 1, 1, 1, 0),
    ("kv dgrad f32", T, 512, 512, 1, 1, 1, 1, 0),
    ("attn dgrad f32 k768", T, 512, 768, 1, 1, 1, 1, 0),
    ("ff1 wgrad", 2816, 512, T, 0, 0, 1, 0, 1),
    ("ff2 wgrad", 512, 1408, T, 0, 0, 1, 0, 1),
    ("out wgrad", 512, 256, T, 0, 0, 1, 0, 1),
    ("kv wgrad", 512, 512, T, 0, 0, 1, 0, 1),
    ("q wgrad", 256, 512, T, 0, 0, 1, 0, 1),
    ("patch wgrad", 512, 4000, T, 0, 0, 1, 0, 1),
]
dev = "cuda"
only = os.environ.get("ONLY")
if only:
    shapes = [s_ for s_ in shapes if any(o in s_[0] for o in only.split(","))]
for name, M, N, K, akm, bkm, cf, split, acc in shapes:
    A = torch.randn((M, K) if akm else (K, M), device=dev).to(torch.bfloat16)
    Bm = torch.randn((N, K) if bkm else (K, N), device=dev).to(torch.bfloat16)
    C = torch.zeros(M, N, device=dev, dtype=torch.float32 if cf else torch.bfloat16)
    if split == 0:
        from ctclip_hip.ops import _splits_for
        split = _splits_for(M, N, K)
        if os.environ.get("WG"):                       # experiment: workgroups (tiles x splits) of a weight-gradient product
            tiles = ((M + 255) // 256) * ((N + 255) // 256)
            split = max(1, int(os.environ["WG"]) // tiles)
    ACT = int(os.environ.get("ACT", 0))
    LDC = N if not os.environ.get("LDC0") else 0
    def run():
        if os.environ.get("ATOMICS"):                  # split-K with f32 atomics instead of the workspace
            hip.gemm_bf16(A, Bm, C, None, None, M, N, K, A.stride(0), Bm.stride(0), LDC, 0, akm, bkm, cf, split, acc, 1.0, ACT, None, 0)
        else:
            hip.gemm_bf16(A, Bm, C, None, None, M, N, K, A.stride(0), Bm.stride(0), LDC, 0, akm, bkm, cf, split, acc, 1.0, ACT)
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"{name:20s} M={M:7d} N={N:5d} K={K:7d} ({akm},{bkm}) split={split:3d}  {ms*1e3:9.1f} us  {2.0*M*N*K/ms/1e9:8.1f} TFLOP/s", flush=True)
    del A, Bm, C
