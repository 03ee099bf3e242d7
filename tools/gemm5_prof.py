"""Where a K-step of gemm5.hip spends its cycles: per-wave s_memtime sums of {MFMA + issue block, vmcnt wait, lgkmcnt wait, barrier}
and the epilogue, from a -DCTCLIP_G5_PROF build.
    CTCLIP_EXTRA_HIPCC_FLAGS="-DCTCLIP_TUNING_KNOBS -DCTCLIP_G5_PROF" python -m ctclip_hip.build   (from ct-clip-ut_amd/)
    CTCLIP_HIP_LIB=.../libctclip_hip_diag.so CTCLIP_GEMM5_MINK=128 python3 tools/gemm5_prof.py
"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ct-clip-ut_amd"))
import torch
from ctclip_hip.lib import hip, library_path

hip.symbols()
dll = hip._dll
B = int(os.environ.get("B", 32))
T = 13824 * B
dev = "cuda"
for name, M, N, K, cf in (("sq4096", 4096, 4096, 4096, 0), ("ff2 fwd f32", T, 512, 1408, 1), ("ff1 dgrad f32", T, 512, 2816, 1),
                          ("kv fwd", T, 512, 512, 0), ("ff1 fwd", T, 2816, 512, 0)):
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = torch.randn(N, K, device=dev).to(torch.bfloat16)
    C = torch.zeros(M, N, device=dev, dtype=torch.float32 if cf else torch.bfloat16)
    run = lambda: hip.gemm_bf16(A, W, C, None, None, M, N, K, K, K, N, 0, 1, 1, cf, 1, 0, 1.0, 0)
    for _ in range(5):
        run()
    prof = torch.zeros(256 * 4 * 8, dtype=torch.int64, device=dev)
    assert dll.ctclip_debug_gemm5_prof(ctypes.c_void_p(prof.data_ptr())) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    assert dll.ctclip_debug_gemm5_prof(ctypes.c_void_p(0)) == 0
    p = prof.view(256, 4, 8).double().cpu()
    steps = p[:, :, 5].clamp(min=1)
    tiles = (steps / (K // 32)).mean().item()
    print(f"{name:14s} {us:8.1f} us  {2.0*M*N*K/us/1e6:7.0f} TFLOP/s   K-steps per wave {steps.mean().item():.0f} ({tiles:.1f} tiles)   life {p[:, :, 7].mean().item():.0f} cycles = "
          f"{p[:, :, 7].mean().item()/us/1e3:.2f} GHz")
    per = lambda i: (p[:, :, i] / steps).mean().item()
    print(f"      per K-step (64 MFMAs = 1024 matrix cycles): issue block {per(0):7.0f}   vmcnt wait {per(1):6.0f}   lgkmcnt wait {per(2):5.0f}   barrier {per(3):6.0f}"
          f"   sum {per(0)+per(1)+per(2)+per(3):7.0f};   epilogue per tile {(p[:, :, 4].sum() / (steps.sum() / (K // 32))).item():8.0f}")
    for w in range(4):
        q = p[:, w]
        st = q[:, 5].clamp(min=1)
        print(f"        wave {w}: issue {(q[:, 0]/st).mean().item():7.0f}  vm {(q[:, 1]/st).mean().item():6.0f}  lgkm {(q[:, 2]/st).mean().item():5.0f}  bar {(q[:, 3]/st).mean().item():6.0f}")
    del A, W, C
