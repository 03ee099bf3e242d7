"""Head-major spatial attention only (csrc/attention_hm.hip) at the production shape: forward with the static shift, the
two backward forms.  B = pairs (24 sequences each).  Prints one line per pass; used for A/B runs of kernel variants
(CTCLIP_HIP_LIB selects the build, tuning knobs come from the environment)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ct-clip-ut_amd"))
import torch
from ctclip_hip.lib import hip
B = int(os.environ.get("B", 64))
REP = int(os.environ.get("REP", 8))
dev = "cuda"


def timeit(fn, n=REP):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


nseq, n, H, D, gh, gw = 24 * B, 576, 8, 32, 24, 24
LOG2E = 1.4426950408889634
ld = H * D
torch.manual_seed(0)
unit = lambda: torch.nn.functional.normalize(torch.randn(nseq, H, n, D, device=dev), dim=-1)
q, k = (unit() * 8 * LOG2E).to(torch.bfloat16), unit().to(torch.bfloat16)
v, do = (torch.randn(nseq, H, n, D, device=dev).to(torch.bfloat16) for _ in range(2))
bias = torch.randn(H, n, n, device=dev)
STAMPS = int(os.environ.get('STAMPS', 0))
o = torch.empty(nseq * n, ld, device=dev, dtype=torch.bfloat16)
lse_buf = torch.zeros(nseq * H * n + (1 << 20 if STAMPS else 0), device=dev)
lse = lse_buf[:nseq * H * n].view(nseq, H, n)
dq, dk, dv = (torch.empty_like(o) for _ in range(3)); delta = torch.empty_like(lse)
R = (2 * gh - 1) * (2 * gw - 1)
dt = torch.zeros(H, R, device=dev)
ones = torch.ones(D, device=dev)
shift = torch.empty(H + 1, device=dev)
hip.attn_shift(ones, ones, D, 8 * LOG2E, bias, n * n, n * n, 1, H, shift)
flops = 4.0 * nseq * H * n * n * D
what = os.environ.get("WHAT", "fwd,bwd,bwdt").split(",")
tag = os.environ.get("TAG", "")
if "fwd" in what:
    t = timeit(lambda: hip.attn_hm_fwd(q, k, v, o, lse, bias, shift, nseq, n, H, ld))
    print(f"{tag} hm fwd (static)   {t:9.1f} us  {flops / t / 1e6:7.1f} TFLOP/s  checksum {float(o.float().abs().mean()):.6f}")
if STAMPS:
    st = lse_buf[nseq * H * n:].view(-1, 8).cpu()
    st = st[st[:, 3] > 0]
    full = st[st[:, 3] == st[:, 3].max()]
    print(f"{tag} stamps: {len(st)} waves, {len(full)} with {int(full[0, 3])} sequences; per sequence (cycles): prologue {float((full[:, 0] / full[:, 3]).mean()):.0f}"
          f"  loop {float((full[:, 1] / full[:, 3]).mean()):.0f}  epilogue {float((full[:, 2] / full[:, 3]).mean()):.0f}; wave life {float(full[:, 5].mean()):.0f} cycles"
          f" (min {float(full[:, 5].min()):.0f} max {float(full[:, 5].max()):.0f}); shader clock {float((full[:, 5] / full[:, 6]).median()) * 100:.0f} MHz;"
          f" wave life {float(full[:, 6].mean()) / 100:.1f} us; starts span {float((st[:, 7].max() - st[:, 7].min())) / 100:.1f} us, last end {float(((st[:, 7] + st[:, 6]).max() - st[:, 7].min())) / 100:.1f} us")
    import numpy as np
    starts = np.sort(((st[:, 7] - st[:, 7].min()) / 100).numpy()[::8])
    print(f"{tag} workgroup start times (us), deciles: {np.percentile(starts, [0, 10, 20, 30, 40, 50, 60, 70, 80, 90, 100]).round(0)}")
if "fwdnb" in what:
    t = timeit(lambda: hip.attn_hm_fwd(q, k, v, o, lse, None, shift, nseq, n, H, ld))
    print(f"{tag} hm fwd (no bias)  {t:9.1f} us  {flops / t / 1e6:7.1f} TFLOP/s")
hip.attn_hm_fwd(q, k, v, o, lse, bias, shift, nseq, n, H, ld)


def bwd(table, b=bias):
    hip.attn_hm_bwd(q, k, v, o, do, lse, delta, dq, dk, dv, b, None, None, dt if table else None, R if table else 0,
                    gh if table else 0, gw if table else 0, nseq, n, H, ld, ld, ld, ld)


if "bwd" in what:
    t = timeit(lambda: bwd(False))
    print(f"{tag} hm bwd (no dbias) {t:9.1f} us  {2.5 * flops / t / 1e6:7.1f} TFLOP/s  checksum {float(dq.float().abs().mean()):.6f} {float(dk.float().abs().mean()):.6f}")
if "bwdt" in what:
    t = timeit(lambda: bwd(True))
    print(f"{tag} hm bwd (+table)   {t:9.1f} us  {2.5 * flops / t / 1e6:7.1f} TFLOP/s")
