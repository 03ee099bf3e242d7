#!/usr/bin/env python3
"""Generate golden input/output vectors by running the REFERENCE modules on CPU.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

It imports the reference's own, unmodified source files from /root/reference/src and stores
seeded inputs, the state dict and the reference outputs as small .npz files next to this
script.  Nothing from the reference's source text is stored: the fixtures are tensors.

Two import shims are needed and neither touches arithmetic that is
pinned here:
  * `beartype` is not installed; it is used as a bare decorator (attention.py:61,312), so a
    pass-through decorator is registered under that name.
  * `vector_quantize_pytorch` is not installed (third party, unpinned).  ctvit.py:6 imports
    `VectorQuantize` from it; a stand-in built on oracle.vq_cosine is registered so the
    *glue* of ctvit.py (patch-embed, encode, rearranges) runs.  VQ arithmetic itself is
    therefore NOT pinned by these vectors ("parity unpinned", see oracle header).
ContinuousPositionBias.forward hard-codes device 'cuda' (attention.py:261); as SURVEY 8c
describes, `cache_rel_pos=True` plus a pre-registered `rel_pos` buffer makes the reference's
own MLP lines (:272-277) run on CPU.
"""
import os
import sys
import types

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/src"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

from oracle import ctclip_oracle as O  # noqa: E402

# ---- shims -------------------------------------------------------------------------------
bt = types.ModuleType("beartype")
bt.beartype = lambda f: f
sys.modules["beartype"] = bt


class _Codebook(nn.Module):
    def __init__(self, codebook_size, dim):
        super().__init__()
        e = torch.empty(1, codebook_size, dim)
        nn.init.kaiming_uniform_(e)
        self.register_buffer("initted", torch.tensor([True]))
        self.register_buffer("cluster_size", torch.zeros(1, codebook_size))
        self.register_buffer("embed", O.unit_rows(e))


class _VQ(nn.Module):
    def __init__(self, dim, codebook_size, use_cosine_sim=True, freeze_codebook=False, **kw):
        super().__init__()
        self._codebook = _Codebook(codebook_size, dim)

    def forward(self, x, freeze_codebook=False):
        out, idx, ncs, nemb = O.vq_cosine(x, self._codebook.embed, self._codebook.cluster_size,
                                           freeze_codebook=freeze_codebook)
        if nemb is not None:
            self._codebook.embed.copy_(nemb)
            self._codebook.cluster_size.copy_(ncs)
        return out, idx, torch.zeros(())


vqmod = types.ModuleType("vector_quantize_pytorch")
vqmod.VectorQuantize = _VQ
sys.modules["vector_quantize_pytorch"] = vqmod

from utils import attention as RA  # noqa: E402  (reference)
from utils.ctvit import CTViT as RefCTViT  # noqa: E402
from models.ctclip import CTCLIP as RefCTCLIP  # noqa: E402
from utils.optimizer import get_optimizer as ref_get_optimizer  # noqa: E402
from transformers import BertConfig, BertModel  # noqa: E402


def npd(d):
    return {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in d.items()}


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **npd(arrays))
    print(f"{name}: {os.path.getsize(path)/1024:.1f} KiB, {len(arrays)} arrays")


def sd(module, prefix=""):
    return {f"sd.{prefix}{k}": v.clone() for k, v in module.state_dict().items()}


def enable_cpb_on_cpu(cpb, h, w):
    cpb.cache_rel_pos = True
    cpb.register_buffer("rel_pos", O.cpb_relpos(h, w), persistent=False)


def randomize(module, gen, scale=0.5):
    """Make every parameter non-trivial (ones/zeros inits hide bugs)."""
    with torch.no_grad():
        for n, p in module.named_parameters():
            if p.numel() == 0:
                continue
            if n.endswith(("gamma", "q_scale", "k_scale")) or (".norm" in n.lower() and n.endswith("weight")):
                p.copy_(1.0 + 0.3 * torch.randn(p.shape, generator=gen))
            elif p.ndim == 1:
                p.copy_(0.2 * torch.randn(p.shape, generator=gen))


# ---- A. building blocks (reference src/utils/attention.py) --------------------------------
def blocks():
    g = torch.Generator().manual_seed(11)
    out = {}
    dim, heads, dh = 56, 4, 8

    ln = RA.LayerNorm(dim)
    randomize(ln, g)
    x = torch.randn(3, 5, dim, generator=g)
    out.update({"ln.x": x, "ln.gamma": ln.gamma, "ln.y": ln(x)})

    ff = RA.FeedForward(dim=dim)
    randomize(ff, g)
    x = torch.randn(2, 7, dim, generator=g, requires_grad=True)
    r = torch.randn(2, 7, dim, generator=g)
    y = ff(x)
    (y * r).sum().backward()
    out.update({"ff.x": x, "ff.r": r, "ff.y": y, "ff.dx": x.grad})
    out.update({f"ff.{k}": v for k, v in ff.state_dict().items()})
    out.update({f"ff.grad.{k}": p.grad for k, p in ff.named_parameters()})

    pg = RA.PEG(dim=8, causal=True)
    randomize(pg, g)
    shape = (2, 3, 4, 5)
    xs = torch.randn(2 * 3, 4 * 5, 8, generator=g, requires_grad=True)      # (b t) (h w) d
    xt = torch.randn(2 * 4 * 5, 3, 8, generator=g, requires_grad=True)      # (b h w) t d : the quirk
    rs = torch.randn_like(xs)
    rt = torch.randn_like(xt)
    ys, yt = pg(xs, shape=shape), pg(xt, shape=shape)
    ((ys * rs).sum() + (yt * rt).sum()).backward()
    out.update({"peg.w": pg.dsconv.weight, "peg.b": pg.dsconv.bias, "peg.shape": np.array(shape),
                "peg.xs": xs, "peg.ys": ys, "peg.xt": xt, "peg.yt": yt, "peg.rs": rs, "peg.rt": rt,
                "peg.dxs": xs.grad, "peg.dxt": xt.grad,
                "peg.dw": pg.dsconv.weight.grad, "peg.db": pg.dsconv.bias.grad})

    at = RA.Attention(dim=dim, dim_head=dh, heads=heads)
    randomize(at, g)
    x = torch.randn(3, 12, dim, generator=g, requires_grad=True)
    bias = torch.randn(heads, 12, 12, generator=g, requires_grad=True)
    r = torch.randn(3, 12, dim, generator=g)
    y, probs = at(x, attn_bias=bias)
    (y * r).sum().backward()
    out.update({"attn.x": x, "attn.bias": bias, "attn.r": r, "attn.y": y, "attn.probs": probs,
                "attn.dx": x.grad, "attn.dbias": bias.grad})
    out.update({f"attn.{k}": v for k, v in at.state_dict().items()})
    out.update({f"attn.grad.{k}": p.grad for k, p in at.named_parameters() if p.grad is not None})
    y2, probs2 = at(x.detach())
    out.update({"attn.y_nobias": y2, "attn.probs_nobias": probs2})

    cpb = RA.ContinuousPositionBias(dim=16, heads=heads)
    enable_cpb_on_cpu(cpb, 3, 4)
    b = cpb(3, 4)
    out.update({"cpb.bias": b})
    out.update({f"cpb.{k}": v for k, v in cpb.state_dict().items()})

    tr = RA.Transformer(dim=dim, depth=2, dim_head=dh, heads=heads, peg=True, peg_causal=True)
    randomize(tr, g)
    shape = (2, 3, 2, 3)
    xsp = torch.randn(2 * 3, 2 * 3, dim, generator=g)
    bsp = torch.randn(heads, 6, 6, generator=g)
    xtm = torch.randn(2 * 2 * 3, 3, dim, generator=g)
    out.update({"tr.shape": np.array(shape), "tr.xs": xsp, "tr.bias": bsp, "tr.xt": xtm,
                "tr.ys": tr(xsp, video_shape=shape, attn_bias=bsp), "tr.yt": tr(xtm, video_shape=shape)})
    out.update({f"tr.{k}": v for k, v in tr.state_dict().items()})
    save("blocks.npz", **out)


# ---- B. CT-ViT glue (reference src/utils/ctvit.py) -----------------------------------------
VIT_CFG = dict(dim=32, codebook_size=64, image_size=16, patch_size=4, temporal_patch_size=2,
               spatial_depth=1, temporal_depth=1, dim_head=8, heads=4)


def make_ref_vit(gen):
    torch.manual_seed(int(torch.randint(0, 10_000, (1,), generator=gen)))
    vit = RefCTViT(**VIT_CFG)
    randomize(vit, gen)
    enable_cpb_on_cpu(vit.spatial_rel_pos_bias, vit.patch_height, vit.patch_width)
    return vit


def ctvit():
    g = torch.Generator().manual_seed(23)
    vit = make_ref_vit(g).eval()
    vol = torch.randn(2, 1, 8, 16, 16, generator=g).clamp(-1, 1)
    tok_pre = vit.to_patch_emb(vol)
    enc = vit.encode(tok_pre)
    tokens = vit(vol)
    ids = vit(vol, return_only_codebook_ids=True)
    out = {"volume": vol, "patch_tokens": tok_pre, "encoded": enc, "tokens": tokens, "indices": ids}
    out.update(sd(vit))
    save("ctvit.npz", **out)


# ---- C. BERT CLS (transformers.BertModel, local library) -----------------------------------
BERT_CFG = dict(hidden_size=32, num_hidden_layers=2, num_attention_heads=4, intermediate_size=64,
                vocab_size=97, max_position_embeddings=40, hidden_dropout_prob=0.0,
                attention_probs_dropout_prob=0.0)


def text_batch(gen, B, L, vocab):
    ids = torch.randint(0, vocab, (B, L), generator=gen)
    lens = torch.randint(L // 4, L + 1, (B,), generator=gen)
    lens[0] = L
    mask = (torch.arange(L)[None] < lens[:, None]).long()
    return {"input_ids": ids, "token_type_ids": torch.zeros_like(ids), "attention_mask": mask}


def bert():
    g = torch.Generator().manual_seed(31)
    torch.manual_seed(31)
    m = BertModel(BertConfig(**BERT_CFG)).eval()
    randomize(m, g)
    txt = text_batch(g, 3, 16, BERT_CFG["vocab_size"])
    hid = m(**txt).last_hidden_state
    out = {"input_ids": txt["input_ids"], "token_type_ids": txt["token_type_ids"],
           "attention_mask": txt["attention_mask"], "last_hidden_state": hid}
    out.update(sd(m))
    save("bert.npz", **out)


# ---- D. CTCLIP forward/backward + 2 optimizer steps (ctclip.py, CTClipTrainer.py:164-204) ---
def ctclip():
    g = torch.Generator().manual_seed(47)
    torch.manual_seed(47)
    text = BertModel(BertConfig(**BERT_CFG))
    randomize(text, g)
    vit = make_ref_vit(g)
    grid = VIT_CFG["image_size"] // VIT_CFG["patch_size"]
    dim_image = grid * grid * VIT_CFG["dim"]
    clip = RefCTCLIP(text_encoder=text, image_encoder=vit, dim_text=BERT_CFG["hidden_size"],
                     dim_image=dim_image, dim_latent=16)
    out = sd(clip)
    B = 3
    batches = []
    for s in range(2):
        vol = torch.randn(B, 1, 8, 16, 16, generator=g).clamp(-1, 1)
        txt = text_batch(g, B, 16, BERT_CFG["vocab_size"])
        batches.append((txt, vol))
        out[f"step{s}.volume"] = vol
        for k, v in txt.items():
            out[f"step{s}.{k}"] = v

    # forward-only record in eval mode (frozen codebook)
    clip.eval()
    sim, il, tl, temp, toks = clip(batches[0][0], batches[0][1])
    out.update({"eval.sim": sim, "eval.image_latents": il, "eval.text_latents": tl, "eval.temp": temp,
                "eval.image_tokens": toks})

    # two training steps: CTClipTrainer.train_step order without Accelerate (fp32, 1 process)
    clip.train()
    opt = ref_get_optimizer(clip.parameters(), lr=1.25e-5, wd=0.0)       # CTClipTrainer.py:50-51,107
    for s, (txt, vol) in enumerate(batches):
        opt.zero_grad()
        sim, *_ = clip(txt, vol)
        tgt = torch.arange(sim.size(0))
        loss = (torch.nn.functional.cross_entropy(sim, tgt)
                + torch.nn.functional.cross_entropy(sim.t(), tgt)) / 2    # CTClipTrainer.py:171-173
        loss.backward()
        if s == 0:
            for k, p in clip.named_parameters():
                if p.grad is not None:
                    out[f"step0.grad.{k}"] = p.grad.clone()
        norm = torch.nn.utils.clip_grad_norm_(clip.parameters(), 0.5)     # CTClipTrainer.py:199-200
        opt.step()
        out[f"step{s}.loss"] = loss.detach()
        out[f"step{s}.grad_norm"] = norm
        out[f"step{s}.sim"] = sim.detach()
    for k in ("to_text_latent.weight", "temperature",
              "visual_transformer.enc_spatial_transformer.layers.0.1.to_q.weight",
              "visual_transformer.to_patch_emb.2.bias",
              "text_transformer.encoder.layer.1.output.dense.weight",
              "visual_transformer.vq._codebook.embed", "visual_transformer.vq._codebook.cluster_size"):
        out[f"final.{k}"] = clip.state_dict()[k]
    save("ctclip.npz", **out)


# ---- E. optimizer factory (reference src/utils/optimizer.py) --------------------------------
def optimizer():
    g = torch.Generator().manual_seed(5)
    out = {}
    for tag, wd in (("adam", 0.0), ("adamw", 1e-2)):
        w = torch.randn(6, 5, generator=g)
        b = torch.randn(5, generator=g)
        pw, pb = nn.Parameter(w.clone()), nn.Parameter(b.clone())
        opt = ref_get_optimizer([pw, pb], lr=1e-2, wd=wd)
        out[f"{tag}.w0"], out[f"{tag}.b0"] = w, b
        out[f"{tag}.class"] = np.array(type(opt).__name__)
        for s in range(3):
            gw = torch.randn(6, 5, generator=g)
            gb = torch.randn(5, generator=g)
            pw.grad, pb.grad = gw.clone(), gb.clone()
            opt.step()
            out[f"{tag}.gw{s}"], out[f"{tag}.gb{s}"] = gw, gb
            out[f"{tag}.w{s+1}"], out[f"{tag}.b{s+1}"] = pw.detach().clone(), pb.detach().clone()
    save("optimizer.npz", **out)


if __name__ == "__main__":
    torch.set_num_threads(4)
    blocks()
    ctvit()
    bert()
    ctclip()
    optimizer()
