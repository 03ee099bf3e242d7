"""Micro-benchmark of the PEG kernels at the CT-ViT shape (b, 24, 24, 24, 512)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ct-clip-ut_amd"))
import torch
from ctclip_hip.lib import hip
B = int(os.environ.get("B", 32))
dev = "cuda"
def timeit(fn, n=5):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
T = H = W = 24; d = 512
x = torch.randn(B, T, H, W, d, device=dev)
dy = torch.randn_like(x)
w27 = torch.randn(27, d, device=dev) * 0.1
bias = torch.randn(d, device=dev)
STAMPS = int(os.environ.get("STAMPS", 0))
ybuf = torch.zeros(x.numel() + (1 << 20 if STAMPS else 0), device=dev)
y = ybuf[:x.numel()].view_as(x); y16 = torch.empty(x.shape, device=dev, dtype=torch.bfloat16)
def show_stamps(tag):
    st = ybuf[x.numel():].view(-1, 8).cpu()
    st = st[st[:, 5] > 0]
    if not len(st): return
    names = ["issue loads", "LDS reads + FMAs", "stores", "rotate + stage (waits for the loads)", "barrier"]
    per = st[:, :5].mean(0) / T
    clk = float((st[:, 5] / st[:, 6]).median()) * 100
    print(f"  {tag} stamps: {len(st)} waves; shader clock {clk:.0f} MHz; wave life {float(st[:, 5].mean()):.0f} cycles = {float(st[:, 6].mean()) / 100:.1f} us;"
          f" starts span {float(st[:, 7].max() - st[:, 7].min()) / 100:.1f} us; per plane (cycles): " + ", ".join(f"{n} {float(v):.0f}" for n, v in zip(names, per))
          + f" = {float(per.sum()):.0f}")
    ybuf[x.numel():].zero_()
dw = torch.zeros(27, d, device=dev); db = torch.zeros(d, device=dev)
gb = x.numel() * 4 / 1e9
t = timeit(lambda: hip.peg_fwd(x, w27, bias, y, y16, B, T, H, W, d, 1)); print(f"peg_fwd        {t:9.1f} us  {(2.5 * gb) / t * 1e6 / 1e3:6.2f} TB/s (x read + y f32 + y bf16)")
if STAMPS: show_stamps("fwd")
t = timeit(lambda: hip.peg_bwd_data(dy, w27, y, y16, B, T, H, W, d, 1)); print(f"peg_bwd_data   {t:9.1f} us  {(2.5 * gb) / t * 1e6 / 1e3:6.2f} TB/s")
if STAMPS: show_stamps("bwd")
t = timeit(lambda: hip.peg_bwd_weight(dy, x, dw, db, B, T, H, W, d)); print(f"peg_bwd_weight {t:9.1f} us  {(2.0 * gb) / t * 1e6 / 1e3:6.2f} TB/s (x + dy read)")
t = timeit(lambda: hip.peg_bwd_fused(dy, x, w27, y, y16, dw, db, B, T, H, W, d, 1)); print(f"peg_bwd_fused  {t:9.1f} us  {(3.5 * gb) / t * 1e6 / 1e3:6.2f} TB/s (x + dy read, dx f32 + bf16 written: data and weight gradient in one pass)")
