"""Micro-benchmark of the VQ nearest-code search at the CT-ViT shape (8192 codes x 512, 13824 tokens per pair): the MFMA sweep
(ctclip_vq_topk_grouped) for 1 / 2 / 4 / 8 code groups and the exact re-rank (ctclip_vq_select) on its candidate lists."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ct-clip-ut_amd"))
import torch
from ctclip_hip.lib import hip

B = int(os.environ.get("B", 8))
GROUPS = [int(v) for v in os.environ.get("VQ_GROUPS", "1,2,4,8").split(",")]
T, C, K = 13824 * B, 8192, 512
Ef = torch.nn.functional.normalize(torch.randn(C, K, device="cuda"), dim=-1)
Xf = torch.randn(T, K, device="cuda")
inv = 1.0 / Xf.norm(dim=-1)
E = Ef.to(torch.bfloat16)
X = (Xf * inv[:, None]).to(torch.bfloat16)


def timeit(fn, n=5):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


ref = None
for g in GROUPS:
    pv = torch.empty(T, 16 * g, device="cuda")
    pi = torch.empty(T, 16 * g, device="cuda", dtype=torch.int32)
    idx = torch.empty(T, dtype=torch.long, device="cuda")
    quant = torch.empty(T, K, device="cuda")
    ms = timeit(lambda: hip.vq_topk_grouped(E, X, pv, pi, C, T, K, K, K, g))
    ms2 = timeit(lambda: hip.vq_select(pv, pi, 16 * g, Xf, inv, Ef, idx, quant, T, K, 2.0 ** -7))
    if ref is None:
        ref = idx.clone()
    same = float((idx == ref).float().mean())
    print(f"vq search B={B} code groups {g}: sweep {ms*1e3:.1f} us  {2.0*T*C*K/ms/1e9:.1f} TFLOP/s; select {ms2*1e3:.1f} us; "
          f"codes equal to the first row's {same:.6f}", flush=True)
