"""Run-to-run reproducibility of the gradients of one small CT-CLIP step (BASELINE config 1 shapes).  Atomics order alone
moves gradients by ~1e-6 of their peak; a race between streams or inside a kernel shows up orders of magnitude above.
usage: flake_check.py [RUNS]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ct-clip-ut_amd"))
os.environ.setdefault("CTCLIP_GEMM_V2_ALL", "1")          # as tests/conftest.py: every kernel variant on small shapes
import torch
from transformers import BertConfig, BertModel
from models.ctclip import CTCLIP
from utils.ctvit import CTViT
from ctclip_hip import ops

torch.manual_seed(0)
vit_cfg = dict(dim=64, codebook_size=256, image_size=64, patch_size=16, temporal_patch_size=16, spatial_depth=2,
               temporal_depth=2, dim_head=32, heads=2)
bcfg = dict(hidden_size=64, num_hidden_layers=2, num_attention_heads=2, intermediate_size=128, vocab_size=211,
            max_position_embeddings=64, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
clip = CTCLIP(text_encoder=BertModel(BertConfig(**bcfg)), image_encoder=CTViT(**vit_cfg), dim_text=64,
              dim_image=4 * 4 * 64, dim_latent=32).cuda().train()
clip.visual_transformer.eval()
gen = torch.Generator().manual_seed(1234)
vol = (torch.randn(4, 1, 64, 64, 64, generator=gen) * 0.5).clamp(-1, 1).cuda()
ids = torch.randint(0, 211, (4, 32), generator=gen)
txt = {"input_ids": ids.cuda(), "token_type_ids": torch.zeros_like(ids).cuda(), "attention_mask": torch.ones_like(ids).cuda()}
ref = None
for run in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    clip.zero_grad(set_to_none=True)
    sim, *_ = clip(txt, vol)
    loss = ops.InfoNCEFn.apply(sim)
    loss.backward()
    torch.cuda.synchronize()
    grads = {k: p.grad.detach().clone() for k, p in clip.named_parameters() if p.grad is not None}
    if ref is None:
        ref = grads
        print(f"run 0: loss {float(loss):.6f}, {len(grads)} gradients")
        continue
    worst = max(((float((grads[k] - ref[k]).abs().max() / (ref[k].abs().max() + 1e-20)), k) for k in ref), key=lambda t: t[0])
    print(f"run {run}: loss {float(loss):.6f}  worst run-to-run gradient difference {worst[0]:.2e} at {worst[1]}", flush=True)
