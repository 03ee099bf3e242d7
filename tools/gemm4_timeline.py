"""Phase timeline of the weight-gradient kernel gemm4 (diagnostic build):

    CTCLIP_EXTRA_HIPCC_FLAGS="-DCTCLIP_G4_STAMPS" python -m ctclip_hip.build        (from ct-clip-ut_amd/)
    CTCLIP_HIP_LIB=ct-clip-ut_amd/ctclip_hip/libctclip_hip_diag.so python tools/gemm4_timeline.py

Every workgroup (one 256 x 256 tile of one split of K) stamps s_memrealtime (10 ns) at start, when its first K-step has landed,
at the end of the matrix loop and after its partial tile is stored; waves 0 and 4 (the two role groups) sum the shader cycles of
every segment of the loop.  Prints where a workgroup's time goes, against the 1024 cycles of pure matrix work per K-step."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ct-clip-ut_amd"))
import numpy as np
import torch
from ctclip_hip.lib import hip, library_path
from ctclip_hip.ops import _splits_for

B = int(os.environ.get("B", 64))
T = 13824 * B
shapes = {"kv": (512, 512), "ff1": (2816, 512), "ff2": (512, 1408), "q": (256, 512), "patch": (512, 4000)}
dll = ctypes.CDLL(library_path())
dll.ctclip_debug_gemm4_stamps.argtypes = [ctypes.c_void_p, ctypes.c_long]
dll.ctclip_debug_gemm4_prof.argtypes = [ctypes.c_void_p]
for name in os.environ.get("ONLY", "kv,ff1,ff2,q").split(","):
    M, N = shapes[name]
    K = T
    A = torch.randn(K, M, device="cuda").to(torch.bfloat16)
    Bm = torch.randn(K, N, device="cuda").to(torch.bfloat16)
    C = torch.zeros(M, N, device="cuda")
    split = _splits_for(M, N, K)
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    nk_total = K // 32
    per = (nk_total + split - 1) // split
    split_eff = (nk_total + per - 1) // per
    nblk = tiles * split_eff
    stamps = torch.zeros(nblk, 8, dtype=torch.int64, device="cuda")
    prof = torch.zeros(nblk, 16, dtype=torch.int64, device="cuda")
    run = lambda: hip.gemm_bf16(A, Bm, C, None, None, M, N, K, M, N, N, 0, 0, 0, 1, split, 1, 1.0, 0)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    assert dll.ctclip_debug_gemm4_prof(ctypes.c_void_p(prof.data_ptr())) == 0
    assert dll.ctclip_debug_gemm4_stamps(ctypes.c_void_p(stamps.data_ptr()), nblk) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    dll.ctclip_debug_gemm4_stamps(None, 0)
    dll.ctclip_debug_gemm4_prof(None)
    ms = e0.elapsed_time(e1)
    s = stamps.cpu().numpy().astype(np.int64)
    hw, xcc = s[:, 0], s[:, 1] & 0xf
    cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 0x1) << 4) | (((hw >> 13) & 0x7) << 5)
    cuid = xcc * 1024 + cu
    t = (s[:, 2:7] - s[:, 2].min()) * 0.01                                                 # us
    start, landed, loop_end, done, issued = t[:, 0], t[:, 1], t[:, 2], t[:, 3], t[:, 4]
    q = lambda x: f"{np.percentile(x, 10):7.2f} / {np.median(x):7.2f} / {np.percentile(x, 90):7.2f}"
    print(f"== {name} wgrad: dW[{M},{N}] over K = {K} tokens, split {split_eff} x {tiles} tiles = {nblk} workgroups of {per} K-steps; "
          f"product + partial-tile sum {ms * 1e3:.0f} us, {2.0 * M * N * K / ms / 1e9:.0f} TFLOP/s; distinct CUs {len(np.unique(cuid))}")
    print(f"   fill (start -> first K-step landed)      p10/p50/p90 us: {q(landed - start)}")
    print(f"   matrix loop                               p10/p50/p90 us: {q(loop_end - landed)}")
    print(f"   partial-tile stores issued / drained      p10/p50/p90 us: {q(issued - loop_end)}  |  {q(done - loop_end)}")
    print(f"   whole workgroup                           p10/p50/p90 us: {q(done - start)}")
    print(f"   workgroup start times (us): p50 {np.median(start):.1f}, p90 {np.percentile(start, 90):.1f}, last {start.max():.1f};"
          f" last workgroup done at {done.max():.1f}")
    pr = prof.cpu().numpy().astype(np.float64)
    names = ["load block issue", "lgkm wait", "barrier(R)", "MFMA block", "vmcnt wait", "barrier(M)"]
    for w, lab in ((0, "wave 0 (reads first)"), (1, "wave 4 (MFMAs first)")):
        med = np.median(pr[:, w * 8:w * 8 + 6], axis=0) / per
        print(f"   {lab}: shader cycles per K-step: " + ", ".join(f"{n} {v:.0f}" for n, v in zip(names, med)) + f"  = {med.sum():.0f} (1024 of matrix work)")
    loop_us = np.median(loop_end - landed)
    print(f"   matrix loop per K-step: {loop_us / per * 1e3:.0f} ns; loop share of the workgroup {np.median((loop_end - landed) / (done - start)):.2f}")
    tmax = done.max()
    grid = np.linspace(0, tmax, 21)[:-1]
    line = [str(int(((landed <= g0) & (loop_end > g0)).sum())) for g0 in grid]
    print(f"   workgroups inside their matrix loop at 20 instants of the launch: " + " ".join(line))
    del A, Bm, C
