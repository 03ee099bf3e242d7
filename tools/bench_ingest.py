"""Volume ingest at a realistic scan size: raw [512,512,300] int16 -> [1,240,480,480] bf16."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ct-clip-ut_amd"))
import torch
from utils.preprocess import process_volume
raw = torch.randint(-1000, 2000, (512, 512, 300), dtype=torch.int16, device="cuda")
for _ in range(2):
    out = process_volume(raw, 1.0, -1024.0, 0.7, 1.0)
torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    out = process_volume(raw, 1.0, -1024.0, 0.7, 1.0)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"ingest 512x512x300 i16 -> {tuple(out.shape)} {out.dtype}: {ms:.3f} ms per scan "
      f"({(raw.numel() * 2 + out.numel() * 2) / ms / 1e6:.1f} GB/s of raw-in + model-input-out)")
