"""CPU oracle for the CT-CLIP contrastive training step -- TEST INFRASTRUCTURE ONLY.

This file is a plain fp32 PyTorch-CPU restatement of the reference's algorithm for the
hot path named in BASELINE.json (`north_star`).  It exists so that the hand-written HIP
path can be checked against something that does not share any code with it.

Rules (see DESIGN.md "Oracle"):
  * Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import
    this module.  Nothing under `ct-clip-ut_amd/` imports it; the product path raises when
    the HIP library is missing instead of falling back to this code.
  * Every function cites the reference `file:line` (relative to /root/reference) it follows.
  * It is pinned against the reference itself: `tests/golden/make_golden.py` imports the
    reference modules in the build container, runs them on seeded inputs and stores
    inputs + outputs under `tests/golden/*.npz`; `tests/test_oracle_golden.py` replays
    those vectors through this file -- including the volume ingest (`preprocess_*`, pinned by the reference's own
    `resize_array` / `crop_and_pad` / `process_file` run behind a `nibabel` stand-in) and the attribution loops
    (`occlusion_heatmap`, `integrated_gradients`, pinned by the reference's own `Visualizations._compute_occlusion` /
    `visualize_integrated_gradients` run through a stand-in `self`).  Two pieces are *parity unpinned* because their
    arithmetic lives in third-party packages that are absent from /root/reference and
    from the image:
      - vector-quantize-pytorch (lucidrains; version not pinned by the reference, call sites
        src/utils/ctvit.py:66,117-118): `vq_cosine` restates the library's published
        cosine-similarity codebook (l2-normalise input, arg-max of dot products against the
        l2-normalised codebook, straight-through estimator, EMA update with decay 0.8).
      - the pretrained CXR-BERT weights / tokenizer (src/train_ctclip.py:17): `bert_cls`
        restates `transformers.BertModel` (present in the image, 5.15.0) and IS pinned by
        golden vectors produced with that local library on random weights.

All functions are written over a flat ``state`` mapping that uses exactly the reference's
state-dict keys, so one seeded state dict can be loaded into the
reference modules, this oracle, and the HIP-backed modules alike.
"""
from __future__ import annotations

import math
from typing import Dict, Mapping, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
State = Mapping[str, Tensor]


# --------------------------------------------------------------------------------------
# small pieces
# --------------------------------------------------------------------------------------

def gammanorm(x: Tensor, gamma: Tensor, eps: float = 1e-5) -> Tensor:
    """Bias-less LayerNorm: reference src/utils/attention.py:27-34 (beta is a zero buffer)."""
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps) * gamma


def affine_layernorm(x: Tensor, w: Tensor, b: Tensor, eps: float = 1e-5) -> Tensor:
    """nn.LayerNorm with bias: reference src/utils/attention.py:46, src/utils/ctvit.py:49,51."""
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps) * w + b


def unit_rows(x: Tensor, eps: float = 1e-12) -> Tensor:
    """F.normalize(dim=-1): reference src/utils/attention.py:21-22."""
    return x / x.norm(dim=-1, keepdim=True).clamp_min(eps)


def gelu_erf(x: Tensor) -> Tensor:
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


# --------------------------------------------------------------------------------------
# K1  tubelet patch embedding            reference src/utils/ctvit.py:44-52,112
# --------------------------------------------------------------------------------------

def tubelet_features(volume: Tensor, patch: int, tpatch: int) -> Tensor:
    """'b c (t pt) (h p1) (w p2) -> b t h w (c pt p1 p2)'  (ctvit.py:45-48).

    Feature index inside a tubelet is ((c*pt + pt_i)*p1 + p1_i)*p2 + p2_i.
    """
    b, c, D, Hh, Ww = volume.shape
    t, h, w = D // tpatch, Hh // patch, Ww // patch
    v = volume.reshape(b, c, t, tpatch, h, patch, w, patch)
    v = v.permute(0, 2, 4, 6, 1, 3, 5, 7)          # b t h w c pt p1 p2
    return v.reshape(b, t, h, w, c * tpatch * patch * patch)


def patch_embed(volume: Tensor, st: State, prefix: str, patch: int, tpatch: int) -> Tensor:
    """Rearrange -> LayerNorm(c*pt*p1*p2) -> Linear(+bias) -> LayerNorm(dim)  (ctvit.py:44-52)."""
    f = tubelet_features(volume, patch, tpatch)
    f = affine_layernorm(f, st[prefix + "1.weight"], st[prefix + "1.bias"])
    f = f @ st[prefix + "2.weight"].t() + st[prefix + "2.bias"]
    return affine_layernorm(f, st[prefix + "3.weight"], st[prefix + "3.bias"])


# --------------------------------------------------------------------------------------
# K8  continuous position bias           reference src/utils/attention.py:230-277
# --------------------------------------------------------------------------------------

def cpb_relpos(h: int, w: int) -> Tensor:
    """Signed log-distance grid, attention.py:262-268: [h*w, h*w, 2] float32."""
    ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    grid = torch.stack((ys, xs)).reshape(2, -1).t()                 # [(h w), 2]
    rel = grid[:, None, :] - grid[None, :, :]                        # i - j
    rel = rel.to(torch.float32)
    return torch.sign(rel) * torch.log(rel.abs() + 1.0)


def cpb_mlp(rows: Tensor, st: State, prefix: str) -> Tensor:
    """The MLP of attention.py:247-253 applied row-wise (fp32, attention.py:272-275)."""
    x = rows @ st[prefix + "net.0.0.weight"].t() + st[prefix + "net.0.0.bias"]
    x = F.leaky_relu(x, 0.1)
    i = 1
    while (prefix + f"net.{i}.0.weight") in st:
        x = x @ st[prefix + f"net.{i}.0.weight"].t() + st[prefix + f"net.{i}.0.bias"]
        x = F.leaky_relu(x, 0.1)
        i += 1
    return x @ st[prefix + f"net.{i}.weight"].t() + st[prefix + f"net.{i}.bias"]


def cpb_bias(h: int, w: int, st: State, prefix: str) -> Tensor:
    """[heads, h*w, h*w] additive attention bias (attention.py:258-277)."""
    out = cpb_mlp(cpb_relpos(h, w), st, prefix)                      # [i, j, heads]
    return out.permute(2, 0, 1).contiguous()


# --------------------------------------------------------------------------------------
# K3  PEG depthwise causal conv          reference src/utils/attention.py:55-83
# --------------------------------------------------------------------------------------

def peg(x: Tensor, weight: Tensor, bias: Tensor, shape: Sequence[int], causal: bool = True) -> Tensor:
    """Depthwise 3x3x3 conv over the *memory order* of x viewed as (b,t,h,w,d).

    attention.py:69 reshapes whatever flat [n_seq, seq, d] tensor it is given to
    (b, t, h, w, d); in the temporal transformer the tokens are ordered (b h w) t, so the
    convolution runs over a scrambled grid.  The restatement keeps that behaviour: it only
    ever looks at flat memory.
    """
    orig = x.shape
    d = x.shape[-1]
    g = x.reshape(*shape, d).permute(0, 4, 1, 2, 3)                  # b d t h w
    tpad = (2, 0) if causal else (1, 1)
    g = F.pad(g, (1, 1, 1, 1, *tpad), value=0.0)                      # attention.py:73-75
    g = F.conv3d(g, weight, bias, groups=d)                           # attention.py:59,76
    return g.permute(0, 2, 3, 4, 1).reshape(orig)


# --------------------------------------------------------------------------------------
# K2,K4,K5,K6  cosine-sim attention      reference src/utils/attention.py:126-182
# --------------------------------------------------------------------------------------

def attention(x: Tensor, st: State, prefix: str, heads: int,
              attn_bias: Optional[Tensor] = None, scale: float = 8.0) -> Tuple[Tensor, Tensor]:
    """Self-attention branch used by CT-ViT (no context, num_null_kv = 0, not causal).

    Returns (to_out(attn @ v), attn probabilities) as attention.py:182 does.
    """
    n_seq, n, _ = x.shape
    # QUIRK: `kv_input = default(context, x)` is bound at :138, BEFORE `x = self.norm(x)` at :140,
    # so keys/values are projected from the UN-normalised input; only the query sees the LayerNorm.
    y = gammanorm(x, st[prefix + "norm.gamma"])                      # :140
    q = y @ st[prefix + "to_q.weight"].t()                           # :142
    kv = x @ st[prefix + "to_kv.weight"].t()                         # :138,142
    k, v = kv.chunk(2, dim=-1)
    split = lambda t: t.reshape(n_seq, n, heads, -1).permute(0, 2, 1, 3)   # b h n d  (:144)
    q, k, v = split(q), split(k), split(v)
    null_kv = st.get(prefix + "null_kv")
    if null_kv is not None and null_kv.shape[1] > 0:                 # :146-149 (empty for CT-ViT)
        nk, nv = null_kv[:, 0::2], null_kv[:, 1::2]
        k = torch.cat((nk.expand(n_seq, -1, -1, -1), k), dim=2)
        v = torch.cat((nv.expand(n_seq, -1, -1, -1), v), dim=2)
    q = unit_rows(q) * st[prefix + "q_scale"]                        # :151-153
    k = unit_rows(k) * st[prefix + "k_scale"]
    sim = torch.einsum("bhid,bhjd->bhij", q, k) * scale              # :155
    if attn_bias is not None:
        sim = sim + attn_bias                                        # :159-161
    probs = sim.softmax(dim=-1)                                      # :174
    out = torch.einsum("bhij,bhjd->bhid", probs, v)                  # :178
    out = out.permute(0, 2, 1, 3).reshape(n_seq, n, -1)              # :180
    return out @ st[prefix + "to_out.weight"].t(), probs             # :182


# --------------------------------------------------------------------------------------
# K7  GEGLU feed-forward                 reference src/utils/attention.py:38-51
# --------------------------------------------------------------------------------------

def feed_forward(x: Tensor, st: State, prefix: str) -> Tensor:
    y = affine_layernorm(x, st[prefix + "0.weight"], st[prefix + "0.bias"])
    hdn = y @ st[prefix + "1.weight"].t()
    val, gate = hdn.chunk(2, dim=-1)                                  # :40  (x first, gate second)
    return (gelu_erf(gate) * val) @ st[prefix + "4.weight"].t()      # :41,50


# --------------------------------------------------------------------------------------
# Transformer stack                      reference src/utils/attention.py:313-336
# --------------------------------------------------------------------------------------

def transformer(x: Tensor, st: State, prefix: str, depth: int, heads: int,
                video_shape: Sequence[int], attn_bias: Optional[Tensor] = None,
                collect_probs: Optional[list] = None) -> Tensor:
    for i in range(depth):
        lp = f"{prefix}layers.{i}."
        x = peg(x, st[lp + "0.dsconv.weight"], st[lp + "0.dsconv.bias"], video_shape) + x   # :325
        a, probs = attention(x, st, lp + "1.", heads, attn_bias)                           # :327
        if collect_probs is not None:
            collect_probs.append(probs)
        x = a + x                                                                           # :328
        x = feed_forward(x, st, lp + "3.") + x                                              # :334
    return gammanorm(x, st[prefix + "norm_out.gamma"])                                      # :336


# --------------------------------------------------------------------------------------
# K9  cosine-similarity vector quantiser (third-party; PARITY UNPINNED, see module docstring)
#     call sites: reference src/utils/ctvit.py:66,117-118
# --------------------------------------------------------------------------------------

def vq_cosine(x: Tensor, embed: Tensor, cluster_size: Optional[Tensor] = None,
              freeze_codebook: bool = True, decay: float = 0.8):
    """x: [b, n, d]; embed: [1, C, d] (already unit rows, as the library keeps it).

    Returns (quantised with straight-through gradient, indices [b, n], new_cluster_size,
    new_embed).  new_* are None when the codebook is frozen.
    The module is always in .train() mode in the reference (ctvit.py:117) so the
    straight-through branch is always taken; the EMA update runs only when
    freeze_codebook is False (ctvit.py:118: freeze_codebook = not self.training).
    """
    code = embed[0]
    xn = unit_rows(x.float())
    flat = xn.reshape(-1, xn.shape[-1])
    scores = flat @ code.t()
    idx = scores.argmax(dim=-1)
    quant = code[idx].reshape(xn.shape)
    out = xn + (quant - xn).detach()
    new_cluster, new_embed = None, None
    if not freeze_codebook:
        with torch.no_grad():
            C = code.shape[0]
            bins = torch.bincount(idx, minlength=C).to(flat.dtype)
            new_cluster = cluster_size.clone() if cluster_size is not None else torch.zeros(1, C)
            new_cluster = new_cluster * decay + bins[None] * (1 - decay)
            empty = bins == 0
            esum = torch.zeros_like(code).index_add_(0, idx, flat.detach())
            mean = esum / bins.masked_fill(empty, 1.0)[:, None]
            mean = unit_rows(mean)
            mean = torch.where(empty[:, None], code, mean)
            new_embed = (code * decay + mean * (1 - decay))[None]
    return out, idx.reshape(x.shape[:-1]), new_cluster, new_embed


# --------------------------------------------------------------------------------------
# CT-ViT                                 reference src/utils/ctvit.py:88-125
# --------------------------------------------------------------------------------------

def ctvit_encode(tokens: Tensor, st: State, prefix: str, cfg: dict,
                 collect_probs: Optional[list] = None) -> Tensor:
    b, t, h, w, d = tokens.shape
    shape = (b, t, h, w)
    bias = cpb_bias(h, w, st, prefix + "spatial_rel_pos_bias.")                    # ctvit.py:89
    x = tokens.reshape(b * t, h * w, d)                                              # :94
    x = transformer(x, st, prefix + "enc_spatial_transformer.", cfg["spatial_depth"],
                    cfg["heads"], shape, bias, collect_probs)                        # :95
    x = x.reshape(b, t, h * w, d).permute(0, 2, 1, 3).reshape(b * h * w, t, d)       # :96,99
    x = transformer(x, st, prefix + "enc_temporal_transformer.", cfg["temporal_depth"],
                    cfg["heads"], shape, None, collect_probs)                        # :100
    return x.reshape(b, h, w, t, d).permute(0, 3, 1, 2, 4).contiguous()              # :101


def ctvit_forward(volume: Tensor, st: State, prefix: str, cfg: dict,
                  training: bool = False, collect_probs: Optional[list] = None):
    """Returns (tokens [b,t,h,w,d], indices [b,t,h,w], new_cluster_size, new_embed)."""
    tok = patch_embed(volume, st, prefix + "to_patch_emb.", cfg["patch_size"], cfg["temporal_patch_size"])
    tok = ctvit_encode(tok, st, prefix, cfg, collect_probs)
    b, t, h, w, d = tok.shape
    q, idx, ncs, nemb = vq_cosine(tok.reshape(b, t * h * w, d), st[prefix + "vq._codebook.embed"],
                                  st.get(prefix + "vq._codebook.cluster_size"),
                                  freeze_codebook=not training)                      # :117-118
    return q.reshape(b, t, h, w, d), idx.reshape(b, t, h, w), ncs, nemb               # :124


# --------------------------------------------------------------------------------------
# K11  BERT text encoder -> CLS          reference call site src/models/ctclip.py:107
#      (arithmetic: transformers.BertModel, absolute positions, post-LN, erf-GELU)
# --------------------------------------------------------------------------------------

def bert_cls(input_ids: Tensor, token_type_ids: Optional[Tensor], attention_mask: Optional[Tensor],
             st: State, prefix: str, n_layers: int, n_heads: int, ln_eps: float = 1e-12,
             return_all: bool = False) -> Tensor:
    B, L = input_ids.shape
    if token_type_ids is None:
        token_type_ids = torch.zeros_like(input_ids)
    e = prefix + "embeddings."
    x = (st[e + "word_embeddings.weight"][input_ids]
         + st[e + "position_embeddings.weight"][:L][None]
         + st[e + "token_type_embeddings.weight"][token_type_ids])
    x = affine_layernorm(x, st[e + "LayerNorm.weight"], st[e + "LayerNorm.bias"], ln_eps)
    add_mask = None
    if attention_mask is not None:
        add_mask = (1.0 - attention_mask.to(x.dtype))[:, None, None, :] * torch.finfo(x.dtype).min
    hd = x.shape[-1] // n_heads
    for i in range(n_layers):
        lp = f"{prefix}encoder.layer.{i}."
        lin = lambda t, name: t @ st[lp + name + ".weight"].t() + st[lp + name + ".bias"]
        split = lambda t: t.reshape(B, L, n_heads, hd).permute(0, 2, 1, 3)
        q = split(lin(x, "attention.self.query"))
        k = split(lin(x, "attention.self.key"))
        v = split(lin(x, "attention.self.value"))
        s = q @ k.transpose(-1, -2) / math.sqrt(hd)
        if add_mask is not None:
            s = s + add_mask
        p = s.softmax(dim=-1)
        ctx = (p @ v).permute(0, 2, 1, 3).reshape(B, L, -1)
        a = lin(ctx, "attention.output.dense")
        x = affine_layernorm(a + x, st[lp + "attention.output.LayerNorm.weight"],
                             st[lp + "attention.output.LayerNorm.bias"], ln_eps)
        m = gelu_erf(lin(x, "intermediate.dense"))
        o = lin(m, "output.dense")
        x = affine_layernorm(o + x, st[lp + "output.LayerNorm.weight"],
                             st[lp + "output.LayerNorm.bias"], ln_eps)
    return x if return_all else x[:, 0, :]


# --------------------------------------------------------------------------------------
# K10,K12  CTCLIP tail + loss            reference src/models/ctclip.py:107-129,
#                                        src/utils/CTClipTrainer.py:164-175
# --------------------------------------------------------------------------------------

def clip_latents(text_cls: Tensor, image_tokens: Tensor, st: State, prefix: str = ""):
    img = image_tokens.mean(dim=1)                                    # ctclip.py:111
    img = img.reshape(img.shape[0], -1)                               # :112
    tl = text_cls @ st[prefix + "to_text_latent.weight"].t()          # :115
    il = img @ st[prefix + "to_visual_latent.weight"].t()             # :116
    tl = tl / tl.norm(dim=-1, keepdim=True)                           # :119
    il = il / il.norm(dim=-1, keepdim=True)                           # :120
    return tl, il


def sim_matrix(il: Tensor, tl: Tensor, temperature: Tensor) -> Tensor:
    return il @ tl.t() * temperature.exp()                            # ctclip.py:127


def symmetric_info_nce(sim: Tensor, targets: Optional[Tensor] = None) -> Tensor:
    """CTClipTrainer.py:164-175."""
    if targets is None:
        targets = torch.arange(sim.shape[0])
    li = F.cross_entropy(sim, targets)
    lt = F.cross_entropy(sim.t(), targets)
    return (li + lt) / 2


def ctclip_forward(text_inputs: Mapping[str, Tensor], volume: Tensor, st: State, cfg: dict,
                   training: bool = False, text_embeds: Tensor = None):
    """Single-process CTCLIP.forward (ctclip.py:99-129). Returns dict of everything.
    `text_embeds` replaces the text encoder's CLS output when text_inputs is None/empty (ctclip.py:107)."""
    if text_inputs:
        cls = bert_cls(text_inputs["input_ids"], text_inputs.get("token_type_ids"),
                       text_inputs.get("attention_mask"), st, "text_transformer.",
                       cfg["text_layers"], cfg["text_heads"])
    else:
        cls = text_embeds
    tokens, idx, ncs, nemb = ctvit_forward(volume, st, "visual_transformer.", cfg, training)
    tl, il = clip_latents(cls, tokens, st)
    sim = sim_matrix(il, tl, st["temperature"])
    return dict(sim=sim, image_latents=il, text_latents=tl, temp=st["temperature"].exp(),
                image_tokens=tokens, indices=idx, text_cls=cls, new_cluster_size=ncs, new_embed=nemb)


# --------------------------------------------------------------------------------------
# K13  grad clip + Adam                  reference src/utils/CTClipTrainer.py:199-202,
#                                        src/utils/optimizer.py:42-54
# --------------------------------------------------------------------------------------

def clip_grad_norm(grads: Sequence[Tensor], max_norm: float) -> Tuple[float, float]:
    """torch.nn.utils.clip_grad_norm_ semantics: returns (total_norm, applied coefficient)."""
    total = math.sqrt(sum(float((g.double() ** 2).sum()) for g in grads))
    coef = min(1.0, max_norm / (total + 1e-6))
    for g in grads:
        g.mul_(coef)
    return total, coef


def adam_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float,
              betas=(0.9, 0.99), eps: float = 1e-8, weight_decay: float = 0.0,
              decoupled: bool = False) -> None:
    """torch.optim.Adam / AdamW single-tensor update (optimizer.py:42-54), in place."""
    b1, b2 = betas
    if weight_decay != 0.0:
        if decoupled:
            p.mul_(1.0 - lr * weight_decay)
        else:
            g = g + weight_decay * p
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


def train_steps(st0: Dict[str, Tensor], batches, cfg: dict, lr: float = 1.25e-5,
                max_grad_norm: float = 0.5, frozen: Sequence[str] = ()):
    """CTClipTrainer.train_step order (CTClipTrainer.py:181-204) on CPU in fp32.

    batches: list of (text_inputs, volume).  Returns (losses, grad_norms, final state).
    Parameters are every floating tensor in st0 that is not a buffer listed in `frozen`.
    """
    st = {k: v.clone() for k, v in st0.items()}
    names = [k for k, v in st.items() if v.is_floating_point() and k not in frozen]
    for k in names:
        st[k].requires_grad_(True)
    ms = {k: torch.zeros_like(st[k]) for k in names}
    vs = {k: torch.zeros_like(st[k]) for k in names}
    losses, norms = [], []
    for step, (txt, vol) in enumerate(batches, start=1):
        out = ctclip_forward(txt, vol, st, cfg, training=True)
        loss = symmetric_info_nce(out["sim"])
        grads = torch.autograd.grad(loss, [st[k] for k in names], allow_unused=True)
        live = [(k, g.clone()) for k, g in zip(names, grads) if g is not None]
        total, _ = clip_grad_norm([g for _, g in live], max_grad_norm)
        with torch.no_grad():
            for k, g in live:
                adam_step(st[k], g, ms[k], vs[k], step, lr)
            if out["new_embed"] is not None:
                st["visual_transformer.vq._codebook.embed"] = out["new_embed"]
                st["visual_transformer.vq._codebook.cluster_size"] = out["new_cluster_size"]
        losses.append(float(loss.detach()))
        norms.append(total)
    return losses, norms, {k: v.detach() for k, v in st.items()}


# ---------------------------------------------------------------------------------------------------------------------
# occlusion sensitivity, restated from reference src/utils/visualizations.py:335-424
# (`Visualizations._compute_occlusion`): serial B=1 forwards, one per voxel window.
# ---------------------------------------------------------------------------------------------------------------------
def occlusion_heatmap(text_inputs, image, st, cfg, patch_size, stride, threshold=0.0, text_embeds=None, rank=0, world_size=1):
    """visualizations.py:335-424 for one process (rank / world_size select the window slice exactly as :352-362).

    image [1,1,D,H,W]; returns the rank's UN-reduced (heatmap, count_map) and, for world_size == 1, the final map
    (normalised, thresholded, rot90) the reference returns on the main process."""
    import numpy as np
    _, _, D, H, W = image.shape
    coords = [(d, h, w)
              for d in range(0, D - patch_size[0] + 1, stride[0])
              for h in range(0, H - patch_size[1] + 1, stride[1])
              for w in range(0, W - patch_size[2] + 1, stride[2])]
    per_rank = len(coords) // world_size                                   # :352-353 (extra windows are dropped, :356)
    coords = coords[:per_rank * world_size][rank * per_rank:(rank + 1) * per_rank]
    heat = torch.zeros(D, H, W, dtype=torch.float64)
    count = torch.zeros(D, H, W, dtype=torch.float64)

    def score(img):
        with torch.no_grad():
            out = ctclip_forward(text_inputs, img, st, cfg, training=False, text_embeds=text_embeds)
        return float(out["sim"][0, 0])

    original = score(image)                                                # :370-375
    for d, h, w in coords:                                                 # :379-391
        occ = image.clone()
        occ[:, :, d:d + patch_size[0], h:h + patch_size[1], w:w + patch_size[2]] = -1
        importance = max(original - score(occ), 0.0)
        heat[d:d + patch_size[0], h:h + patch_size[1], w:w + patch_size[2]] += importance
        count[d:d + patch_size[0], h:h + patch_size[1], w:w + patch_size[2]] += 1
    final = None
    if world_size == 1:                                                    # :409-424
        c = count.clone()
        c[c == 0] = 1
        hm = (heat / c).float()
        hm = (hm - hm.min()) / (hm.max() - hm.min() + 1e-8)
        hm = torch.nn.functional.interpolate(hm[None, None], size=(D, H, W), mode="trilinear", align_corners=False)[0, 0]
        hm = hm.numpy().copy()
        hm[hm < threshold] = 0
        final = np.rot90(hm, k=-1, axes=(1, 2))
    return heat, count, final


# ---------------------------------------------------------------------------------------------------------------------
# SURVEY §8(f) row f1, second half: integrated gradients, restated from reference src/utils/visualizations.py:851-910
# ---------------------------------------------------------------------------------------------------------------------
def integrated_gradients(text_inputs, image, st, cfg, steps=50, text_embeds=None):
    """visualizations.py:851-910 for one process.  Returns (avg_grads [1,1,D,H,W] f32, final map [D,H,W] after the
    reference's normalise / 0.90-quantile / **0.05 / rot90 post-processing)."""
    import numpy as np
    baseline = torch.ones_like(image)                                      # :853-854 (baseline_value = 1)
    diff = image - baseline
    grads = []
    for alpha in torch.linspace(0, 1, steps):                              # :861
        x = (baseline + alpha * diff).detach().requires_grad_()
        out = ctclip_forward(text_inputs, x, st, cfg, training=False, text_embeds=text_embeds)
        out["sim"][0, 0].backward()                                        # :868-869 (rank 0, world size 1)
        grads.append(x.grad.detach().clone())
    avg = torch.stack(grads).mean(dim=0)                                   # :878
    return avg, ig_postprocess(diff, avg)


def ig_postprocess(diff, avg_grads):
    """visualizations.py:879-898."""
    import numpy as np
    ig = (diff * avg_grads).squeeze().relu()
    ig = (ig - ig.min()) / (ig.max() + 1e-8)
    ig = ig.cpu().numpy()
    q = np.quantile(ig, 0.90)
    ig = np.where(ig >= q, ig, 0.0)
    ig = ig ** 0.05
    ig = ig / (ig.max() + 1e-8)
    return np.rot90(ig, k=-1, axes=(1, 2))


# ---------------------------------------------------------------------------------------------------------------------
# volume ingest, restated from reference src/utils/preprocess.py:20-82,123-152
# ---------------------------------------------------------------------------------------------------------------------
def preprocess_resize_array(array, current_spacing, target_spacing):
    """preprocess.py:20-37"""
    shape = array.shape[2:]
    factors = [current_spacing[i] / target_spacing[i] for i in range(3)]
    new_shape = [int(shape[i] * factors[i]) for i in range(3)]
    return torch.nn.functional.interpolate(array, size=new_shape, mode="trilinear", align_corners=False)


def preprocess_crop_and_pad(array, target_shape, pad_value=-1):
    """preprocess.py:38-82 ([H, W, D] order; centre crop / symmetric pad per axis)"""
    current = array.shape
    out = array
    for i in range(3):
        size, target = current[i], target_shape[i]
        if size > target:
            start = (size - target) // 2
            out = out.narrow(i, start, target)
        elif size < target:
            total = target - size
            before = total // 2
            pad = [0, 0, 0, 0, 0, 0]
            pad[2 * (2 - i)] = before
            pad[2 * (2 - i) + 1] = total - before
            out = torch.nn.functional.pad(out, pad, mode="constant", value=pad_value)
    return out


def preprocess_volume(raw_hwd, slope, intercept, xy_spacing, z_spacing, target_shape=(480, 480, 240),
                      target_spacing=(1.5, 0.75, 0.75)):
    """preprocess.py:123-152 for model_type 'ctclip'.  raw_hwd [H, W, D] -> [1, D_t, H_t, W_t] f32."""
    img = torch.as_tensor(raw_hwd).float()
    img = slope * img + intercept                                          # :124-125
    img = img.permute(2, 0, 1)[None, None]                                 # :127-131
    img = preprocess_resize_array(img, (z_spacing, xy_spacing, xy_spacing), target_spacing)   # :133-137
    img = torch.clamp(img, -1000, 1000) / 1000.0                           # :139-141
    img = img[0, 0].permute(1, 2, 0)                                       # :145
    img = preprocess_crop_and_pad(img, target_shape, pad_value=-1)         # :147-149
    return img.permute(2, 0, 1).unsqueeze(0)                               # :151-152 (+ squeeze(0) of :157)
