"""Phase timeline of gemm3 workgroups (diagnostic build):

    CTCLIP_EXTRA_HIPCC_FLAGS=-DCTCLIP_G3_STAMPS python -m ctclip_hip.build        (from ct-clip-ut_amd/)
    CTCLIP_HIP_LIB=ct-clip-ut_amd/ctclip_hip/libctclip_hip_diag.so python tools/gemm_timeline.py

Every workgroup stamps s_memrealtime (10 ns) at start, when its first K-step has landed, at the end of the matrix loop and
after its stores have drained; the script prints phase durations and how many workgroups are in each phase over time."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ct-clip-ut_amd"))
import numpy as np
import torch
from ctclip_hip.lib import hip, library_path

B = int(os.environ.get("B", 32))
T = 13824 * B
shapes = {"ff1": (T, 2816, 512, 0), "kv": (T, 512, 512, 0), "ff2": (T, 512, 1408, 1), "sq4096": (4096, 4096, 4096, 0),
          "tiny8": (2048, 256, 512, 0), "tiny64": (4096, 1024, 512, 0), "tiny256": (8192, 2048, 512, 0),
          "tiny8f": (2048, 256, 512, 1), "tiny256f": (8192, 2048, 512, 1)}
dll = ctypes.CDLL(library_path())
dll.ctclip_debug_gemm3_stamps.argtypes = [ctypes.c_void_p, ctypes.c_long]
for name in os.environ.get("ONLY", "ff1,kv").split(","):
    M, N, K, cf = shapes[name]
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    Bm = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    C = torch.zeros(M, N, device="cuda", dtype=torch.float32 if cf else torch.bfloat16)
    bn = int(os.environ.get("CTCLIP_GEMM3_BN", 256))
    nblk = ((M + 255) // 256) * ((N + bn - 1) // bn)
    stamps = torch.zeros(nblk, 8, dtype=torch.int64, device="cuda")
    prof = torch.zeros(nblk, 16, dtype=torch.int64, device="cuda")
    dll.ctclip_debug_gemm3_prof.argtypes = [ctypes.c_void_p]
    assert dll.ctclip_debug_gemm3_prof(ctypes.c_void_p(prof.data_ptr())) == 0
    run = lambda: hip.gemm_bf16(A, Bm, C, None, None, M, N, K, K, K, N, 0, 1, 1, cf, 1, 0, 1.0, 0)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    assert dll.ctclip_debug_gemm3_stamps(ctypes.c_void_p(stamps.data_ptr()), nblk) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    dll.ctclip_debug_gemm3_stamps(None, 0)
    s = stamps.cpu().numpy().astype(np.int64)
    hw, xcc = s[:, 0], s[:, 1] & 0xf
    cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 0x1) << 4) | (((hw >> 13) & 0x7) << 5)       # cu_id, sh_id, se_id
    cuid = xcc * 1024 + cu
    t = (s[:, 2:6] - s[:, 2].min()) * 0.01                                                 # us
    start, landed, loop_end, done = t[:, 0], t[:, 1], t[:, 2], t[:, 3]
    print(f"== {name}: M={M} N={N} K={K} BN={bn}: {nblk} workgroups, kernel {e0.elapsed_time(e1)*1e3:.0f} us, "
          f"{2.0*M*N*K/e0.elapsed_time(e1)/1e9:.0f} TFLOP/s; distinct CUs seen {len(np.unique(cuid))}")
    q = lambda x: f"{np.percentile(x,10):6.2f} / {np.median(x):6.2f} / {np.percentile(x,90):6.2f}"
    print(f"   fill (start -> first K-step landed)   p10/p50/p90 us: {q(landed-start)}")
    print(f"   matrix loop                            p10/p50/p90 us: {q(loop_end-landed)}")
    print(f"   epilogue (stores drained)              p10/p50/p90 us: {q(done-loop_end)}")
    print(f"   whole workgroup                        p10/p50/p90 us: {q(done-start)}")
    pr = prof.cpu().numpy().astype(np.float64)
    nk = K // 32
    names = ["load issue", "lgkm wait", "barrier(R)", "MFMA block", "vmcnt wait", "barrier(M)"]
    for w, lab in ((0, "wave 0 (group 0)"), (1, "wave 4 (group 1)")):
        med = np.median(pr[:, w * 8:w * 8 + 6], axis=0) / nk
        print(f"   {lab}: shader cycles per K-step: " + ", ".join(f"{n} {v:.0f}" for n, v in zip(names, med)) + f"  = {med.sum():.0f}")
    issued = (s[:, 6] - s[:, 2].min()) * 0.01
    print(f"   epilogue: stores of wave 0 issued       p10/p50/p90 us: {q(issued-loop_end)}   (the rest is the drain)")
    print(f"   occupancy API: {dll.ctclip_debug_gemm3_occupancy(bn, 0)} workgroups per CU")
    # census: the largest number of workgroups simultaneously resident on one CU
    worst = 0
    ev = {}
    for i in range(nblk):
        ev.setdefault(cuid[i], []).append((start[i], 1)); ev[cuid[i]].append((done[i], -1))
    for v in ev.values():
        c = m = 0
        for _, d in sorted(v):
            c += d; m = max(m, c)
        worst = max(worst, m)
    print(f"   census: at most {worst} workgroups resident on one CU at a time")
    # gaps between consecutive workgroups on one CU slot
    order = np.argsort(start)
    per_cu = {}
    for i in order:
        per_cu.setdefault(cuid[i], []).append(i)
    conc = [len(v) for v in per_cu.values()]
    print(f"   workgroups per CU over the launch: min {min(conc)} median {int(np.median(conc))} max {max(conc)}")
    # occupancy of phases over time
    end = done.max()
    bins = np.arange(0, end, max(end / 40, 0.5))
    print("   t(us)   fill  loop  epi   (workgroups in each phase, chip-wide)")
    for b in bins:
        f = int(((start <= b) & (landed > b)).sum()); l = int(((landed <= b) & (loop_end > b)).sum()); e = int(((loop_end <= b) & (done > b)).sum())
        print(f"   {b:6.1f}  {f:5d} {l:5d} {e:5d}")
    # per-CU phase overlap: time during which a CU has at least one workgroup in the matrix loop
    busy = 0.0
    for v in per_cu.values():
        iv = sorted((landed[i], loop_end[i]) for i in v)
        cur_s, cur_e = iv[0]
        for a, b_ in iv[1:]:
            if a > cur_e:
                busy += cur_e - cur_s; cur_s, cur_e = a, b_
            else:
                cur_e = max(cur_e, b_)
        busy += cur_e - cur_s
    print(f"   a CU has a workgroup inside its matrix loop {100*busy/len(per_cu)/end:.0f} % of the launch")
    del A, Bm, C, stamps
