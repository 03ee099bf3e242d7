"""Sustained MFMA rate of the part under a register-only load (no memory traffic)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ct-clip-ut_amd"))
import torch
from ctclip_hip.lib import hip
out = torch.zeros(1, device="cuda")
for blocks, iters in ((256, 20000), (512, 20000), (256, 200000)):
    hip.probe_mfma(out, blocks, 100)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); hip.probe_mfma(out, blocks, iters); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    flops = blocks * 8 * iters * 16 * (2.0 * 32 * 32 * 16)
    print(f"blocks={blocks} iters={iters}: {ms:8.2f} ms  {flops / ms / 1e9:8.1f} TFLOP/s")
