"""Diagnostic (not a test): per-stage and per-tensor parity of the HIP path vs oracle / golden vectors."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ct-clip-ut_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from conftest import load_golden, sub
from oracle import ctclip_oracle as O
DEV = "cuda"

def perr(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))

def cos(a, b):
    a, b = a.detach().float().cpu().reshape(-1), b.detach().float().cpu().reshape(-1)
    return float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30))

def grads_report(named, ref):
    rows = []
    for k, gr in ref.items():
        if gr is None or gr.numel() < 2 or float(gr.abs().max()) == 0: continue
        g = named[k].grad
        if g is None: rows.append((k, float('nan'), 0, 0)); continue
        rows.append((k, cos(g, gr), float(g.norm()), float(gr.norm())))
    rows.sort(key=lambda r: r[1])
    for r in rows[:25]: print(f"   cos {r[1]:.4f}  |hip| {r[2]:.3e} |ref| {r[3]:.3e}  {r[0]}")
    print(f"   ... {len(rows)} tensors, median cos {sorted(r[1] for r in rows)[len(rows)//2]:.5f}")

def golden():
    from test_hip_model import build_clip, batches
    from utils.CTClipTrainer import CTClipTrainer
    g = load_golden("ctclip")
    clip = build_clip(g)
    clip.train()
    vol, txt = batches(g)[0]
    sim, *_ = clip(txt, vol)
    from ctclip_hip import ops
    loss = ops.InfoNCEFn.apply(sim)
    loss.backward()
    print("golden step0 loss", float(loss), float(g["step0.loss"]))
    grads_report(dict(clip.named_parameters()), sub(g, "step0.grad."))

def cfg1():
    from transformers import BertConfig, BertModel
    from models.ctclip import CTCLIP
    from utils.ctvit import CTViT
    from ctclip_hip import ops
    from ctclip_hip.text import bert_last_hidden_state
    torch.manual_seed(0)
    vit_cfg = dict(dim=64, codebook_size=256, image_size=64, patch_size=16, temporal_patch_size=16, spatial_depth=2,
                   temporal_depth=2, dim_head=32, heads=2)
    bcfg = dict(hidden_size=64, num_hidden_layers=2, num_attention_heads=2, intermediate_size=128, vocab_size=211,
                max_position_embeddings=64, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    clip = CTCLIP(text_encoder=BertModel(BertConfig(**bcfg)), image_encoder=CTViT(**vit_cfg), dim_text=64,
                  dim_image=4 * 4 * 64, dim_latent=32)
    st = {k: v.clone() for k, v in clip.state_dict().items()}
    gen = torch.Generator().manual_seed(1234)
    vol = (torch.randn(4, 1, 64, 64, 64, generator=gen) * 0.5).clamp(-1, 1)
    ids = torch.randint(0, 211, (4, 32), generator=gen)
    lens = torch.randint(8, 33, (4,), generator=gen)
    mask = (torch.arange(32)[None] < lens[:, None]).long()
    txt = {"input_ids": ids, "token_type_ids": torch.zeros_like(ids), "attention_mask": mask}
    cfg = dict(vit_cfg, text_layers=2, text_heads=2)
    # oracle stages
    frozen = {k for k in st if k.endswith(".beta") or "vq._codebook." in k or not st[k].is_floating_point()}
    sto = {k: (v.clone().requires_grad_(True) if (v.is_floating_point() and k not in frozen) else v) for k, v in st.items()}
    pt_o = O.patch_embed(vol, sto, "visual_transformer.to_patch_emb.", 16, 16)
    enc_o = O.ctvit_encode(pt_o, sto, "visual_transformer.", cfg)
    out_o = O.ctclip_forward(txt, vol, sto, cfg, training=True)
    loss_o = O.symmetric_info_nce(out_o["sim"])
    loss_o.backward()
    clip = clip.to(DEV).train()
    vit = clip.visual_transformer
    pt_h = vit.patch_embed(vol.to(DEV))
    print("patch tokens", perr(pt_h, pt_o))
    enc_h = vit.encode(pt_o.detach().to(DEV))
    print("encoded (from oracle patch tokens)", perr(enc_h, enc_o))
    tx = {k: v.to(DEV) for k, v in txt.items()}
    sim, il, tl, temp, toks = clip(tx, vol.to(DEV))
    idx_h = None
    print("text cls", perr(bert_last_hidden_state(clip.text_transformer, **tx)[:, 0], out_o["text_cls"]))
    print("image tokens", perr(toks, out_o["image_tokens"]), "  tokens differing:",
          int(((toks.cpu() - out_o["image_tokens"]).abs().amax(-1) > 1e-4).sum()), "of", toks[..., 0].numel())
    print("text latents", perr(tl, out_o["text_latents"]), " image latents", perr(il, out_o["image_latents"]))
    print("sim", perr(sim, out_o["sim"]))
    loss = ops.InfoNCEFn.apply(sim)
    print("loss", float(loss), float(loss_o), abs(float(loss) - float(loss_o)) / float(loss_o))
    loss.backward()
    grads_report(dict(clip.named_parameters()), {k: v.grad for k, v in sto.items() if isinstance(v, torch.Tensor) and v.requires_grad})

if __name__ == "__main__":
    print("=== golden"); golden()
    print("=== cfg1"); cfg1()
