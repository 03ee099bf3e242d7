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
    torch.manual_seed(11)          # module constructors draw their initial weights from the GLOBAL generator: seed it too,
    out = {}                       # so that this fixture regenerates bit for bit (every other generator already does)
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

    # the nearest-code decisions of every forward below (the VQ output's second element): an arg-max, so a reduced-precision
    # implementation may flip genuine near-ties; the HIP parity tests pin the decisions to these to separate the continuous
    # arithmetic from the flips (and count the flips on their own)
    codes = []
    hook = vit.vq.register_forward_hook(lambda m, i, o: codes.append(o[1].detach().clone()))

    # forward-only record in eval mode (frozen codebook)
    clip.eval()
    sim, il, tl, temp, toks = clip(batches[0][0], batches[0][1])
    out.update({"eval.sim": sim, "eval.image_latents": il, "eval.text_latents": tl, "eval.temp": temp,
                "eval.image_tokens": toks, "eval.indices": codes.pop()})

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
        out[f"step{s}.indices"] = codes.pop()
    hook.remove()
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


# ---- F. volume ingest (reference src/utils/preprocess.py) ------------------------------------
def preprocess():
    """`nibabel` is not installed; preprocess.py:3 imports it only for `nib.load(path).get_fdata()` (:12-14).  A stand-in
    module whose load() hands back arrays registered here lets the reference's own `process_file`, `resize_array` and
    `crop_and_pad` run unmodified on synthetic scans."""
    import pandas as pd
    scans = {}

    class _Img:
        def __init__(self, a):
            self.a = a

        def get_fdata(self):
            return self.a

    nb = types.ModuleType("nibabel")
    nb.load = lambda path: _Img(scans[str(path)])
    sys.modules["nibabel"] = nb
    from utils import preprocess as RP  # noqa: E402  (reference)

    g = torch.Generator().manual_seed(71)
    out = {}
    # resize_array (:20-37) and crop_and_pad (:38-82) directly, crop / pad / identity on different axes
    for i, (shape, cur, tgt) in enumerate([((9, 14, 11), (2.0, 0.9, 0.9), (1.5, 0.75, 0.75)),
                                            ((12, 10, 16), (1.0, 0.6, 0.6), (1.5, 0.75, 0.75)),
                                            ((8, 8, 8), (1.5, 0.75, 0.75), (1.5, 0.75, 0.75))]):
        x = torch.randn(1, 1, *shape, generator=g)
        out[f"resize{i}.x"], out[f"resize{i}.cur"], out[f"resize{i}.tgt"] = x, np.array(cur), np.array(tgt)
        out[f"resize{i}.y"] = RP.resize_array(x, cur, tgt)
    for i, (shape, tgt) in enumerate([((10, 7, 12), (6, 11, 12)), ((5, 9, 4), (8, 4, 9)), ((6, 6, 6), (6, 6, 6)),
                                      ((7, 8, 9), (4, 5, 6)), ((3, 4, 5), (8, 9, 10))]):
        x = torch.randn(*shape, generator=g)
        out[f"crop{i}.x"], out[f"crop{i}.tgt"] = x, np.array(tgt)
        out[f"crop{i}.y"] = RP.crop_and_pad(x, tgt, pad_value=-1)
    # the whole process_file pipeline (:84-157) for model_type "ctclip": target 480 x 480 x 240 is hard-coded there, so the
    # output is [1, 240, 480, 480]; only its non-pad box is stored (everything outside it is the pad value -1, asserted)
    for i, (hwd, xy, z, slope, icpt) in enumerate([((40, 36, 20), 1.0, 2.0, 1.0, -1024.0),
                                                   ((30, 50, 24), 0.6, 1.2, 2.0, -2048.0)]):
        raw = torch.randint(-200, 3000, hwd, generator=g).double().numpy()
        name = f"scan{i}.nii.gz"
        scans[f"/fake/{name}"] = raw
        df = pd.DataFrame({"VolumeName": [name], "RescaleSlope": [slope], "RescaleIntercept": [icpt],
                           "XYSpacing": [f"[{xy}, {xy}]"], "ZSpacing": [z]})
        y = RP.process_file(f"/fake/{name}", name, df, "ctclip")
        assert tuple(y.shape) == (1, 240, 480, 480)
        nz = (y[0] != -1).nonzero()
        lo, hi = nz.min(0).values, nz.max(0).values + 1
        box = y[0, lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]]
        masked = y.clone()
        masked[0, lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]] = -1
        assert bool((masked == -1).all())
        out[f"file{i}.raw"] = raw.astype(np.float32)
        out[f"file{i}.meta"] = np.array([slope, icpt, xy, z])
        out[f"file{i}.lo"], out[f"file{i}.hi"], out[f"file{i}.box"] = lo, hi, box
    save("preprocess.npz", **out)


# ---- G. attribution loops (reference src/utils/visualizations.py) ----------------------------
def attribution():
    """`Visualizations._compute_occlusion` (:335-424) and the numeric part of `visualize_integrated_gradients` (:851-898)
    run as the reference wrote them, on the reference CTCLIP (tiny, CPU), through a stand-in `self` that supplies only what
    those two methods read (model, rank, world_size, accelerator.device / is_main_process; GIF rendering and result
    directories stubbed out).  Importing the module seeds the global RNGs and switches deterministic algorithms on
    (:29-39); the latter is switched back off afterwards."""
    import tempfile
    from pathlib import Path
    if "nibabel" not in sys.modules:
        sys.modules["nibabel"] = types.ModuleType("nibabel")
    from utils import visualizations as RV  # noqa: E402  (reference)
    torch.use_deterministic_algorithms(False)

    g = torch.Generator().manual_seed(83)
    torch.manual_seed(83)
    vit_cfg = dict(dim=32, codebook_size=512, image_size=16, patch_size=4, temporal_patch_size=2, spatial_depth=1,
                   temporal_depth=1, dim_head=8, heads=4)     # 6 x 4 x 4 tokens, 512 codes: occlusions do move codes
    text = BertModel(BertConfig(**BERT_CFG)).eval()
    randomize(text, g)
    vit = RefCTViT(**vit_cfg)
    randomize(vit, g)
    enable_cpb_on_cpu(vit.spatial_rel_pos_bias, vit.patch_height, vit.patch_width)
    clip = RefCTCLIP(text_encoder=text, image_encoder=vit, dim_text=BERT_CFG["hidden_size"], dim_image=4 * 4 * 32,
                     dim_latent=16).eval()
    image = (torch.randn(1, 1, 12, 16, 16, generator=g) * 0.5).clamp(-1, 1)
    txt = text_batch(g, 1, 12, BERT_CFG["vocab_size"])
    txt["attention_mask"][:] = 1
    out = sd(clip)
    out.update({"image": image, **{f"txt.{k}": v for k, v in txt.items()}})

    tmp = Path(tempfile.mkdtemp())

    class Acc:
        device = torch.device("cpu")
        is_main_process = True

    class Self:                                          # what the two methods read from `self`
        model, accelerator, rank, world_size = clip, Acc(), 0, 1

        def _results_subdirectory(self, name):
            return tmp

        def visualize_overlay(self, *a, **k):
            pass

    patch, stride, thr = (4, 8, 8), (4, 4, 8), 0.2          # 3 x 3 x 2 = 18 windows, overlapping along h
    hm = RV.Visualizations._compute_occlusion(Self(), image, txt, None, patch, stride, thr)
    assert float((hm > 0).mean()) > 0.2, "degenerate occlusion fixture: no window moved the score"
    out.update({"occ.patch": np.array(patch), "occ.stride": np.array(stride), "occ.threshold": np.array(thr),
                "occ.heatmap": np.ascontiguousarray(hm)})
    # two ranks: each returns nothing on rank != 0 before the reduce; only the window slicing is reference code worth
    # pinning there, and it is covered by the single-process list above (ranks take contiguous equal slices, :352-362)
    RV.Visualizations.visualize_integrated_gradients(Self(), image, txt, None, "scan", None, steps=5)
    out["ig.steps"] = np.array(5)
    out["ig.map"] = np.load(tmp / "scan.npy")
    save("attribution.npz", **out)


if __name__ == "__main__":
    torch.set_num_threads(4)
    which = sys.argv[1:] or ["blocks", "ctvit", "bert", "ctclip", "optimizer", "preprocess", "attribution"]
    for name in which:
        globals()[name]()
