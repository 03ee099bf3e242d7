"""Micro-benchmark of ctclip_vq_topk at the CT-ViT shape (8192 codes x 512, 13824 tokens per pair)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ct-clip-ut_amd"))
import torch
from ctclip_hip.lib import hip

B = int(os.environ.get("B", 8))
T, C, K = 13824 * B, 8192, 512
E = torch.nn.functional.normalize(torch.randn(C, K, device="cuda"), dim=-1).to(torch.bfloat16)
X = torch.nn.functional.normalize(torch.randn(T, K, device="cuda"), dim=-1).to(torch.bfloat16)
pv = torch.empty(T, 16, device="cuda")
pi = torch.empty(T, 16, device="cuda", dtype=torch.int32)
run = lambda: hip.vq_topk(E, X, pv, pi, C, T, K, K, K)
for _ in range(2): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"vq_topk B={B}: {ms*1e3:.1f} us  {2.0*T*C*K/ms/1e9:.1f} TFLOP/s")
