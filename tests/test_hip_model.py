"""GPU parity of the HIP-backed modules against (a) the golden vectors produced by the reference itself
(tests/golden/*.npz) and (b) the CPU oracle on BASELINE config 1.

The HIP path computes GEMM/attention operands in bf16 with f32 accumulation; tolerances are stated per check:
activations 3e-2 of the tensor's peak, gradients 6e-2 of peak (bf16 grads through 2-8 layers), and the
contrastive loss 1e-3 relative (the north-star bar).
"""
import math

import pytest
import torch

from conftest import load_golden, sub

pytestmark = pytest.mark.gpu
DEV = "cuda"


def peak_err(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def check(name, got, ref, tol):
    e = peak_err(got, ref)
    print(f"  {name}: peak-rel err {e:.3e} (tol {tol:.0e})")
    assert math.isfinite(e) and e <= tol, f"{name}: {e} > {tol}"


def grad_parity(named, ref, rel_tol, label):
    """Magnitude-aware gradient comparison.  Tensors whose true gradient is (near) zero -- e.g. BERT key biases, to
    which softmax is invariant -- carry only rounding noise in either implementation, so a tensor is compared by
    ||hip - ref|| / ||ref|| only if its reference norm is at least 1e-3 of the largest tensor norm; all tensors
    together must agree as one concatenated vector."""
    items = [(k, named[k].grad, g) for k, g in ref.items() if g is not None and g.numel() > 0]
    big = max(float(g.norm()) for _, _, g in items)
    worst = ("", 0.0)
    num = den = dot = nh = 0.0
    for k, gh, gr in items:
        assert gh is not None, k
        gh, gr = gh.detach().float().cpu(), gr.detach().float().cpu()
        num += float((gh - gr).pow(2).sum()); den += float(gr.pow(2).sum())
        dot += float((gh * gr).sum()); nh += float(gh.pow(2).sum())
        if float(gr.norm()) >= 1e-3 * big:
            e = float((gh - gr).norm() / gr.norm())
            if e > worst[1]:
                worst = (k, e)
            assert e <= rel_tol, (k, e)
    glob_rel, glob_cos = (num / den) ** 0.5, dot / ((nh * den) ** 0.5)
    print(f"  {label}: {len(items)} tensors, worst significant-tensor rel err {worst[1]:.3e} ({worst[0]}), "
          f"global rel err {glob_rel:.3e}, global cosine {glob_cos:.6f}")
    assert glob_cos > 0.999 and glob_rel < rel_tol


def same_trajectory(ref_state, got_state, codes_ref, codes_got, lr, steps, label):
    """Two runs of the same training steps from the same state, executed by the same kernels: what may differ is the order of
    the f32 atomic adds inside the order-dependent gradient sums (include/ctclip_hip.h names them).  Two mechanisms turn
    that last-bit noise into visible differences, and both are bounded here instead of being ignored:
      * Adam moves every element by ~lr * sign(g) in its first steps, so an element whose gradient IS rounding noise
        (zero-initialised biases, BERT key biases the softmax is invariant to) may move the other way: allowed per element
        is 2 lr per step, or 1e-3 of the tensor's peak if that is larger;
      * the VQ nearest-code decision is an arg-max: weights that differ by ~lr can flip a genuine near-tie in a LATER step,
        and a flipped token moves its two codebook rows by (1 - decay) of a unit vector.  The code decisions of both runs
        are therefore compared first: if they are identical the codebook buffers must agree to 1e-4 of their peak; if a
        near-tie flipped (at most 2 % of the decisions may), only the rows no flipped token touched are compared.
    Returns the fraction of flipped decisions."""
    flipped_codes = set()
    n_dec = n_flip = 0
    for a, b in zip(codes_ref, codes_got):
        a, b = a.reshape(-1).cpu(), b.reshape(-1).cpu()
        diff = a != b
        n_dec += a.numel()
        n_flip += int(diff.sum())
        flipped_codes.update(a[diff].tolist())
        flipped_codes.update(b[diff].tolist())
    frac = n_flip / max(1, n_dec)
    worst, worst_name = 0.0, ""
    for k, v in got_state.items():
        if not (v.is_floating_point() and v.numel()):
            continue
        r = ref_state[k]
        d = (v - r).abs()
        peak = float(r.abs().max())
        if "vq._codebook." in k:
            allowed = 1e-4 * peak + 1e-7
            if flipped_codes:                                   # embed [1, C, d] / cluster_size [1, C]: drop the touched rows
                keep = torch.ones(r.shape[1], dtype=torch.bool, device=d.device)
                keep[torch.tensor(sorted(flipped_codes), device=d.device)] = False
                d = d[:, keep]
        else:
            allowed = max(2.0 * steps * lr, 1e-3 * peak)
        dev_ = float(d.max()) if d.numel() else 0.0
        if dev_ / allowed > worst:
            worst, worst_name = dev_ / allowed, f"{k}: |dev| {dev_:.2e}, peak {peak:.2e}"
    print(f"  {label}: {n_flip} of {n_dec} code decisions flipped; worst deviation in units of its allowance "
          f"{worst:.3f}  [{worst_name}]")
    assert frac <= 0.02, f"{label}: {n_flip} of {n_dec} nearest-code decisions differ between two runs of the same steps"
    assert worst <= 1.0, f"{label}: {worst_name}"
    return frac


def cos(a, b):
    a, b = a.detach().float().cpu().reshape(-1), b.detach().float().cpu().reshape(-1)
    return float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30))


def dev(d):
    return {k: (v.to(DEV) if isinstance(v, torch.Tensor) else v) for k, v in d.items()}


# ------------------------------------------------------------------------------------------- blocks
def test_feed_forward_golden():
    from utils.attention import FeedForward
    g = load_golden("blocks")
    ff = FeedForward(dim=56).to(DEV)
    ff.load_state_dict(dev({k: v for k, v in sub(g, "ff.").items() if k[0].isdigit()}))
    x = g["ff.x"].to(DEV).requires_grad_(True)
    y = ff(x)
    check("ff y", y, g["ff.y"], 3e-2)
    (y * g["ff.r"].to(DEV)).sum().backward()
    check("ff dx", x.grad, g["ff.dx"], 4e-2)
    for k in ("0.weight", "0.bias", "1.weight", "4.weight"):
        check("ff d" + k, dict(ff.named_parameters())[k].grad, g["ff.grad." + k], 4e-2)


def test_peg_golden_both_layouts():
    from utils.attention import PEG
    g = load_golden("blocks")
    peg = PEG(dim=8, causal=True).to(DEV)
    peg.dsconv.weight.data.copy_(g["peg.w"])
    peg.dsconv.bias.data.copy_(g["peg.b"])
    shape = tuple(int(v) for v in g["peg.shape"])
    xs = g["peg.xs"].to(DEV).requires_grad_(True)
    xt = g["peg.xt"].to(DEV).requires_grad_(True)          # temporal ordering: exercises the memory-order quirk
    ys, yt = peg(xs, shape=shape), peg(xt, shape=shape)
    check("peg spatial", ys, g["peg.ys"], 1e-5)
    check("peg temporal", yt, g["peg.yt"], 1e-5)
    ((ys * g["peg.rs"].to(DEV)).sum() + (yt * g["peg.rt"].to(DEV)).sum()).backward()
    check("peg dxs", xs.grad, g["peg.dxs"], 1e-5)
    check("peg dxt", xt.grad, g["peg.dxt"], 1e-5)
    check("peg dw", peg.dsconv.weight.grad, g["peg.dw"], 1e-5)
    check("peg db", peg.dsconv.bias.grad, g["peg.db"], 1e-5)


def test_attention_golden():
    from utils.attention import Attention
    g = load_golden("blocks")
    at = Attention(dim=56, dim_head=8, heads=4).to(DEV)
    names = ("null_kv", "q_scale", "k_scale", "norm.gamma", "norm.beta", "context_norm.gamma", "context_norm.beta",
             "to_q.weight", "to_kv.weight", "to_out.weight")
    at.load_state_dict(dev({k: g["attn." + k] for k in names}))
    at.return_attn = True
    x = g["attn.x"].to(DEV).requires_grad_(True)
    bias = g["attn.bias"].to(DEV).requires_grad_(True)
    y, probs = at(x, attn_bias=bias)
    check("attn y", y, g["attn.y"], 3e-2)
    check("attn probs", probs, g["attn.probs"], 3e-2)
    (y * g["attn.r"].to(DEV)).sum().backward()
    check("attn dx", x.grad, g["attn.dx"], 5e-2)
    check("attn dbias", bias.grad, g["attn.dbias"], 5e-2)
    for k in ("q_scale", "k_scale", "norm.gamma", "to_q.weight", "to_kv.weight", "to_out.weight"):
        check("attn d" + k, dict(at.named_parameters())[k].grad, g["attn.grad." + k], 5e-2)
    y2, p2 = at(x.detach())
    check("attn y (no bias)", y2, g["attn.y_nobias"], 3e-2)
    check("attn probs (no bias)", p2, g["attn.probs_nobias"], 3e-2)


@pytest.mark.parametrize("heads,n", [(3, 64), (4, 64), (4, 80), (2, 704)])
def test_attention_module_layout_choice_vs_oracle(heads, n):
    """d_head 32 with an odd head count, a ragged last tile, or more tiles than the head-major kernels hold must fall back to
    the row-major attention kernels; (4, 64) takes the head-major ones.  Same answers either way: output 3e-2 of peak,
    gradients 6e-2 of peak against the f32 oracle (reference src/utils/attention.py:126-182)."""
    from ctclip_hip import ops
    from oracle import ctclip_oracle as O
    from utils.attention import Attention
    dim = 64
    torch.manual_seed(heads * 1000 + n)
    at = Attention(dim=dim, dim_head=32, heads=heads)
    with torch.no_grad():
        at.q_scale.copy_(1.0 + 0.2 * torch.randn(32))
        at.k_scale.copy_(1.0 + 0.2 * torch.randn(32))
        at.norm.gamma.copy_(1.0 + 0.2 * torch.randn(dim))
    st = {k: v.detach().clone().requires_grad_(v.is_floating_point() and v.numel() > 0 and not k.endswith("beta"))
          for k, v in at.state_dict().items()}
    x = torch.randn(5, n, dim)
    bias = 0.5 * torch.randn(heads, n, n)
    r = torch.randn(5, n, dim)
    xo, bo = x.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    yo, _ = O.attention(xo, st, "", heads, attn_bias=bo)
    (yo * r).sum().backward()
    assert ops.attn_head_major_ok(n, 32, 32, dim, heads, False) == ((heads, n) == (4, 64))
    at = at.to(DEV)
    xh, bh = x.to(DEV).requires_grad_(True), bias.to(DEV).requires_grad_(True)
    out = at(xh, attn_bias=bh)
    yh = out[0] if isinstance(out, tuple) else out
    check("attention y", yh, yo, 3e-2)
    (yh * r.to(DEV)).sum().backward()
    check("attention dx", xh.grad, xo.grad, 6e-2)
    check("attention dbias", bh.grad, bo.grad, 6e-2)
    named = dict(at.named_parameters())
    for k in ("q_scale", "k_scale", "norm.gamma", "to_q.weight", "to_kv.weight", "to_out.weight"):
        check("attention d" + k, named[k].grad, st[k].grad, 6e-2)


def test_position_bias_golden():
    from utils.attention import ContinuousPositionBias
    g = load_golden("blocks")
    cpb = ContinuousPositionBias(dim=16, heads=4).to(DEV)
    cpb.load_state_dict(dev({k: v for k, v in sub(g, "cpb.").items() if k.startswith("net.")}))
    check("cpb dense bias", cpb(3, 4), g["cpb.bias"], 1e-5)


def test_transformer_golden():
    from utils.attention import Transformer
    g = load_golden("blocks")
    tr = Transformer(dim=56, depth=2, dim_head=8, heads=4, peg=True, peg_causal=True).to(DEV)
    tr.load_state_dict(dev({k: v for k, v in sub(g, "tr.").items() if k.startswith(("layers", "norm_out"))}))
    shape = tuple(int(v) for v in g["tr.shape"])
    check("transformer spatial", tr(g["tr.xs"].to(DEV), video_shape=shape, attn_bias=g["tr.bias"].to(DEV)), g["tr.ys"], 3e-2)
    check("transformer temporal", tr(g["tr.xt"].to(DEV), video_shape=shape), g["tr.yt"], 3e-2)


# ------------------------------------------------------------------------------------------- CT-ViT / BERT / CTCLIP
VIT_CFG = dict(dim=32, codebook_size=64, image_size=16, patch_size=4, temporal_patch_size=2, spatial_depth=1,
               temporal_depth=1, dim_head=8, heads=4)
BERT_CFG = dict(hidden_size=32, num_hidden_layers=2, num_attention_heads=4, intermediate_size=64, vocab_size=97,
                max_position_embeddings=40, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)


def test_ctvit_golden():
    from utils.ctvit import CTViT
    g = load_golden("ctvit")
    vit = CTViT(**VIT_CFG).to(DEV).eval()
    missing, unexpected = vit.load_state_dict(dev(sub(g, "sd.")), strict=True), None
    vol = g["volume"].to(DEV)
    pt = vit.patch_embed(vol)
    check("patch tokens", pt, g["patch_tokens"], 3e-2)
    check("encoded", vit.encode(g["patch_tokens"].to(DEV)), g["encoded"], 4e-2)
    idx = vit(vol, return_only_codebook_ids=True).cpu()
    agree = float((idx == g["indices"]).float().mean())
    print(f"  codebook index agreement {agree:.3f}")
    assert agree > 0.9
    same = (idx == g["indices"])
    tok = vit(vol).cpu()
    assert float((tok[same] - g["tokens"][same]).abs().max()) < 1e-5      # gathered codebook rows are exact


def test_bert_golden():
    from transformers import BertConfig, BertModel
    from ctclip_hip.text import bert_last_hidden_state
    g = load_golden("bert")
    m = BertModel(BertConfig(**BERT_CFG)).to(DEV).eval()
    m.load_state_dict(dev(sub(g, "sd.")))
    hid = bert_last_hidden_state(m, g["input_ids"].to(DEV), g["token_type_ids"].to(DEV), g["attention_mask"].to(DEV))
    mask = g["attention_mask"].bool()
    check("bert hidden (valid tokens)", hid.cpu()[mask], g["last_hidden_state"][mask], 3e-2)


def test_bert_layer_with_dropout_vs_torch_same_flags():
    """One encoder layer in train mode with the config's dropouts (transformers BertSelfAttention / BertSelfOutput /
    BertOutput) against plain torch f32 math with the same keep flags: the flags are the library's counter-based draws for
    (seed, layer, site), materialised here through ctclip_dropout_keep; the layer itself re-evaluates them in its kernels."""
    from transformers import BertConfig, BertModel
    from ctclip_hip.text import BertLayerFn, _layer_params, _layer_shadows
    torch.manual_seed(3)
    B, L, Hd, heads, I = 3, 32, 128, 2, 256
    p_hid, p_att = 0.2, 0.3
    cfg = BertConfig(hidden_size=Hd, num_hidden_layers=1, num_attention_heads=heads, intermediate_size=I, vocab_size=50,
                     max_position_embeddings=L, hidden_dropout_prob=p_hid, attention_probs_dropout_prob=p_att)
    layer = BertModel(cfg).to(DEV).train().encoder.layer[0]
    dh = Hd // heads
    x0 = torch.randn(B * L, Hd, device=DEV)
    lens = torch.tensor([L, L // 2, L - 5])
    mask_add = ((torch.arange(L)[None] >= lens[:, None]).float() * torch.finfo(torch.float32).min).to(DEV)
    dy = torch.randn(B * L, Hd, device=DEV)
    P = _layer_params(layer)

    # reference with the same flags: the library's counter-based draws for (seed, layer 5, site) -- ctclip_dropout_keep
    from ctclip_hip.lib import hip
    from ctclip_hip.text import _site_offset
    seed, layer_idx = 123456789012345, 5
    def flags(shape, p, site):
        k = torch.empty(*shape, dtype=torch.uint8, device=DEV)
        hip.dropout_keep(k, k.numel(), p, seed, _site_offset(layer_idx, site))
        return k
    ka, k1, k2 = flags((B, heads, L, L), p_att, 0), flags((B * L, Hd), p_hid, 1), flags((B * L, Hd), p_hid, 2)
    for k, pdrop in ((ka, p_att), (k1, p_hid), (k2, p_hid)):          # the draws are fair and the sites differ
        assert abs(float(k.float().mean()) - (1 - pdrop)) < 4 * math.sqrt(pdrop * (1 - pdrop) / k.numel()) + 1e-3
    assert not torch.equal(k1, k2)
    again = flags((B * L, Hd), p_hid, 1)
    assert torch.equal(again, k1)
    keep_a, keep1, keep2 = ka.float() / (1 - p_att), k1.float() / (1 - p_hid), k2.float() / (1 - p_hid)
    xr = x0.clone().requires_grad_(True)
    lin = torch.nn.functional.linear
    split = lambda t: t.reshape(B, L, heads, dh).permute(0, 2, 1, 3)
    q, k, v = split(lin(xr, P[0], P[1])), split(lin(xr, P[2], P[3])), split(lin(xr, P[4], P[5]))
    s_ = q @ k.transpose(-1, -2) / math.sqrt(dh) + mask_add[:, None, None, :]
    ctxv = ((s_.softmax(-1) * keep_a) @ v).permute(0, 2, 1, 3).reshape(B * L, Hd)
    a = torch.nn.functional.layer_norm(lin(ctxv, P[6], P[7]) * keep1 + xr, (Hd,), P[8], P[9], cfg.layer_norm_eps)
    h = torch.nn.functional.gelu(lin(a, P[10], P[11]))
    ref = torch.nn.functional.layer_norm(lin(h, P[12], P[13]) * keep2 + a, (Hd,), P[14], P[15], cfg.layer_norm_eps)
    ref.backward(dy)
    ref_grads = {i: P[i].grad.clone() for i in (0, 6, 7, 10, 12, 13, 14)}
    for t in P:
        t.grad = None

    xh = x0.clone().requires_grad_(True)
    lcfg = (B, L, heads, dh, ops_head_pad(dh), float(cfg.layer_norm_eps), p_hid, p_att, seed, layer_idx)
    out = BertLayerFn.apply(xh, mask_add, _layer_shadows(layer, heads, dh, ops_head_pad(dh)), lcfg, *P)
    out.backward(dy)
    torch.cuda.synchronize()
    valid = (torch.arange(L)[None] < lens[:, None]).reshape(-1).to(DEV)
    check("layer output", out[valid], ref[valid], 3e-2)
    check("input gradient", xh.grad, xr.grad, 5e-2)
    for i, g in ref_grads.items():
        check(f"parameter {i} gradient", P[i].grad, g, 6e-2)
    # eval mode of the whole model: dropouts off, two calls agree exactly
    from ctclip_hip.text import bert_last_hidden_state
    m = BertModel(cfg).to(DEV)
    ids = torch.randint(0, 50, (B, L), device=DEV)
    m.eval()
    e1 = bert_last_hidden_state(m, ids)
    e2 = bert_last_hidden_state(m, ids)
    assert torch.equal(e1, e2)
    m.train()
    torch.manual_seed(11)
    t1 = bert_last_hidden_state(m, ids)
    assert not torch.equal(t1, e1) and bool(torch.isfinite(t1).all())
    t2 = bert_last_hidden_state(m, ids)                       # a new seed per forward
    assert not torch.equal(t2, t1)
    torch.manual_seed(11)
    t3 = bert_last_hidden_state(m, ids)                       # torch.manual_seed makes it reproducible
    assert torch.equal(t3, t1)


def ops_head_pad(dh):
    from ctclip_hip import ops
    return ops.head_pad(dh)


def build_clip(g):
    from transformers import BertConfig, BertModel
    from models.ctclip import CTCLIP
    from utils.ctvit import CTViT
    text = BertModel(BertConfig(**BERT_CFG))
    vit = CTViT(**VIT_CFG)
    clip = CTCLIP(text_encoder=text, image_encoder=vit, dim_text=32, dim_image=4 * 4 * 32, dim_latent=16)
    clip.load_state_dict(sub(g, "sd."), strict=True)          # reference state dict, key for key
    return clip.to(DEV)


def batches(g):
    out = []
    for s in range(2):
        txt = {k: g[f"step{s}.{k}"].to(DEV) for k in ("input_ids", "token_type_ids", "attention_mask")}
        out.append((g[f"step{s}.volume"].to(DEV), txt))
    return out


def code_agreement(vq, ref_idx, label, floor):
    """Free-running nearest-code decisions against the reference's: an arg-max over bf16 encoder outputs flips genuine
    near-ties (the reference under fp16 autocast does too), so the CONTINUOUS arithmetic is compared with the decisions
    pinned to the reference's (VectorQuantize.forced_indices) and the flips are counted here."""
    got = vq.last_indices.reshape(-1).cpu()
    agree = float((got == ref_idx.reshape(-1)).float().mean())
    print(f"  {label}: {agree:.4f} of {got.numel()} nearest-code decisions equal the reference's")
    assert agree >= floor, (label, agree)
    return agree


def test_ctclip_eval_forward_golden():
    g = load_golden("ctclip")
    clip = build_clip(g).eval()
    vol, txt = batches(g)[0]
    vq = clip.visual_transformer.vq
    clip(txt, vol)                                                  # free-running: how many decisions flip
    code_agreement(vq, g["eval.indices"], "eval forward", 0.9)
    vq.forced_indices = g["eval.indices"].reshape(vol.shape[0], -1)
    sim, il, tl, temp, toks = clip(txt, vol)
    vq.forced_indices = None
    check("text latents", tl, g["eval.text_latents"], 3e-2)
    check("image tokens (pinned codes)", toks, g["eval.image_tokens"], 1e-5)
    check("image latents", il, g["eval.image_latents"], 2e-2)
    check("sim", sim, g["eval.sim"], 2e-2)
    check("exp(temp)", temp, g["eval.temp"], 1e-6)


def test_ctclip_training_steps_golden():
    """Two CTClipTrainer.train_step()s vs the reference's losses / grad-norm / first-step gradients."""
    from utils.CTClipTrainer import CTClipTrainer
    g = load_golden("ctclip")
    clip = build_clip(g)
    trainer = CTClipTrainer(clip, batch_size=3, lr=1.25e-5, wd=0.0, max_grad_norm=0.5, results_folder=None)
    ref_grads = sub(g, "step0.grad.")
    vq = clip.visual_transformer.vq
    for s, batch in enumerate(batches(g)):
        # the reference's nearest-code decisions of this step (an arg-max: see code_agreement); the step itself -- forward,
        # backward, clip, Adam, EMA codebook update -- runs on them
        vq.forced_indices = g[f"step{s}.indices"].reshape(batch[0].shape[0], -1)
        loss = trainer.train_step(batch)
        vq.forced_indices = None
        ref = float(g[f"step{s}.loss"])
        rel = abs(loss - ref) / abs(ref)
        print(f"  step {s}: loss {loss:.6f} ref {ref:.6f} rel {rel:.2e}; grad-norm {trainer.optim.grad_norm():.4f} "
              f"ref {float(g[f'step{s}.grad_norm']):.4f}")
        assert rel <= 1e-3
        assert abs(trainer.optim.grad_norm() - float(g[f"step{s}.grad_norm"])) <= 1e-2 * float(g[f"step{s}.grad_norm"])
        if s == 0:
            # gradients as left in .grad by the step: clipped by min(1, 0.5/norm) like the reference (:199-200)
            coef = min(1.0, 0.5 / (float(g["step0.grad_norm"]) + 1e-6))
            # 9e-2: the query / key path tensors of this toy's first spatial layer (to_q.weight, q_scale, k_scale: 16-token sequences,
            # d_head 8 padded to 32, logits of +-8) sit at 6.0-8.5e-2 depending on the bf16 rounding REALISATION -- 6.1-6.7e-2 with
            # the separate head-norm pass, 7.8-8.5e-2 with the normalisation in the GEMM epilogue, while over four seeds of config 1
            # (d_head 32) the two forms are indistinguishable (mean 9.91e-3 vs 9.90e-3, max 1.28e-2 vs 1.32e-2:
            # profiles/r05_headnorm_in_gemm.txt); every other tensor is below 4e-2
            grad_parity(dict(clip.named_parameters()), {k: v * 1.0 for k, v in ref_grads.items()}, 9e-2,
                        "step-0 gradients vs reference")
    # post-step weights (reference: clip_grad_norm_(0.5) + Adam(lr 1.25e-5), two steps).  Adam's first steps move every
    # weight by ~lr whatever the gradient's size, so the UPDATE (final - initial) is what is compared: a wrong bias
    # correction, eps placement, clip coefficient or step count changes its length or direction at once.
    sd0, final = sub(g, "sd."), clip.state_dict()
    for k, ref_w in sub(g, "final.").items():
        if "vq._codebook" in k:
            continue
        du, dr = (final[k].detach().cpu() - sd0[k]).double().reshape(-1), (ref_w - sd0[k]).double().reshape(-1)
        c = float(du @ dr / (du.norm() * dr.norm() + 1e-30))
        ratio = float(du.norm() / (dr.norm() + 1e-30))
        print(f"  update of {k}: cosine {c:.5f}, length ratio {ratio:.4f} (|update| {float(dr.norm()):.3e})")
        assert c >= 0.97 and 0.97 <= ratio <= 1.03, k
        # element-wise: Adam's first steps are ~lr * sign(g), so an element whose (near-zero) gradient has the other sign under
        # bf16 noise ends up to ~2 lr per step away; nothing may be further off than that
        worst = float((final[k].detach().cpu() - ref_w).abs().max())
        print(f"    largest element deviation {worst:.2e} (bound {6 * 1.25e-5:.2e})")
        assert worst <= 6 * 1.25e-5, k
    # EMA codebook after two training forwards (decay 0.8) on the pinned decisions: every row follows the reference's
    # (bf16 encoder outputs enter the running sums)
    emb, emb_ref = final["visual_transformer.vq._codebook.embed"].cpu(), g["final.visual_transformer.vq._codebook.embed"]
    row_err = float((emb - emb_ref).abs().amax(-1).max())
    print(f"  codebook after two EMA updates: worst row deviation {row_err:.2e}")
    assert row_err <= 2e-2
    cs, cs_ref = final["visual_transformer.vq._codebook.cluster_size"].cpu(), g["final.visual_transformer.vq._codebook.cluster_size"]
    assert float((cs - cs_ref).abs().max()) <= 1e-5


# ------------------------------------------------------------------------------------------- BASELINE config 1
def _config1():
    from transformers import BertConfig, BertModel
    from models.ctclip import CTCLIP
    from utils.ctvit import CTViT
    torch.manual_seed(0)
    vit_cfg = dict(dim=64, codebook_size=256, image_size=64, patch_size=16, temporal_patch_size=16, spatial_depth=2,
                   temporal_depth=2, dim_head=32, heads=2)
    bcfg = dict(hidden_size=64, num_hidden_layers=2, num_attention_heads=2, intermediate_size=128, vocab_size=211,
                max_position_embeddings=64, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    clip = CTCLIP(text_encoder=BertModel(BertConfig(**bcfg)), image_encoder=CTViT(**vit_cfg), dim_text=64,
                  dim_image=4 * 4 * 64, dim_latent=32)
    gen = torch.Generator().manual_seed(1234)
    data = []
    for _ in range(2):
        vol = (torch.randn(4, 1, 64, 64, 64, generator=gen) * 0.5).clamp(-1, 1)
        ids = torch.randint(0, 211, (4, 32), generator=gen)
        lens = torch.randint(8, 33, (4,), generator=gen)
        mask = (torch.arange(32)[None] < lens[:, None]).long()
        data.append(({"input_ids": ids, "token_type_ids": torch.zeros_like(ids), "attention_mask": mask}, vol))
    return clip, data, dict(vit_cfg, text_layers=2, text_heads=2)


def test_config1_vs_oracle():
    """BASELINE.json configs[0]: 4 synthetic 64^3 volumes + 32-token reports, 2-layer CT-ViT / 2-layer text encoder.

    (1) code decisions pinned to the oracle's: loss within 1e-3 relative and gradients within 6e-2 -- this isolates
        the continuous arithmetic (every kernel) from discrete nearest-code flips;
    (2) free-running: bf16 noise (~3e-3 on encoder outputs) legitimately flips genuine near-ties of the VQ arg-max, an
        effect any reduced-precision path has (the reference's fp16 autocast too); bounded here: >= 95 % of the codes
        agree and the loss stays within 2e-2."""
    from ctclip_hip import ops
    from oracle import ctclip_oracle as O
    clip, data, cfg = _config1()
    st0 = {k: v.clone() for k, v in clip.state_dict().items()}
    frozen = {k for k in st0 if k.endswith(".beta") or "vq._codebook." in k or not st0[k].is_floating_point()}
    sto = {k: (v.clone().requires_grad_(True) if (v.is_floating_point() and k not in frozen) else v) for k, v in st0.items()}
    txt, vol = data[0]
    out_o = O.ctclip_forward(txt, vol, sto, cfg, training=False)
    loss_o = O.symmetric_info_nce(out_o["sim"])
    loss_o.backward()
    clip = clip.to(DEV).train()
    txd = {k: v.to(DEV) for k, v in txt.items()}
    vq = clip.visual_transformer.vq
    # (2) free running
    clip.visual_transformer.eval()          # freeze the codebook for this comparison (oracle ran training=False)
    sim, *_ = clip(txd, vol.to(DEV))
    free = float(ops.InfoNCEFn.apply(sim))
    agree = float((vq.last_indices.cpu().reshape(-1) == out_o["indices"].reshape(-1)).float().mean())
    print(f"  free-running: loss {free:.6f} vs oracle {float(loss_o):.6f} (rel {abs(free - float(loss_o)) / float(loss_o):.2e}), "
          f"code agreement {agree:.4f}")
    assert agree >= 0.95 and abs(free - float(loss_o)) / float(loss_o) <= 2e-2
    # (1) pinned codes
    vq.forced_indices = out_o["indices"].reshape(vol.shape[0], -1)
    sim, *_ = clip(txd, vol.to(DEV))
    loss = ops.InfoNCEFn.apply(sim)
    rel = abs(float(loss) - float(loss_o)) / float(loss_o)
    print(f"  pinned codes: loss {float(loss):.6f} vs oracle {float(loss_o):.6f} rel {rel:.2e}")
    assert rel <= 1e-3
    loss.backward()
    grad_parity(dict(clip.named_parameters()), {k: v.grad for k, v in sto.items() if v.requires_grad}, 6e-2,
                "config-1 gradients vs oracle (pinned codes)")
    vq.forced_indices = None


def test_config1_training_trajectory_follows_the_oracle():
    """Sixteen CTClipTrainer.train_step()s on config 1 (its two batches in turn) at a learning rate 40x the default, against the
    f32 oracle trained the same way (forward, clip at 0.5, Adam, EMA codebook): the loss must FALL, and the two trajectories
    must stay together -- an end-to-end check of backward + fused clip / Adam + codebook update over many steps, where a wrong
    sign, a stale shadow weight or a lost moment shows as drift.  Free-running nearest-code decisions (bf16 near-ties flip,
    see test_config1_vs_oracle) and 16 steps of Adam bound how tight this can be; the bars are 3x what was observed."""
    from utils.CTClipTrainer import CTClipTrainer
    from oracle import ctclip_oracle as O
    clip, data, cfg = _config1()
    st0 = {k: v.clone() for k, v in clip.state_dict().items()}
    frozen = [k for k in st0 if k.endswith(".beta") or "vq._codebook." in k or not st0[k].is_floating_point()]
    steps, lr = 16, 5e-4
    batches = [data[s % 2] for s in range(steps)]
    ref_losses, _, _ = O.train_steps(st0, batches, cfg, lr=lr, max_grad_norm=0.5, frozen=frozen)
    trainer = CTClipTrainer(clip, batch_size=4, lr=lr, results_folder=None)
    losses = [trainer.train_step((vol, txt)) for txt, vol in batches]
    worst = max(abs(a - b) / abs(b) for a, b in zip(losses, ref_losses))
    print("  hip   : " + " ".join(f"{x:.4f}" for x in losses))
    print("  oracle: " + " ".join(f"{x:.4f}" for x in ref_losses))
    print(f"  worst relative distance over {steps} steps: {worst:.3e}; last-four mean {sum(losses[-4:]) / 4:.4f} vs first-two {sum(losses[:2]) / 2:.4f}")
    assert all(math.isfinite(x) for x in losses)
    assert sum(losses[-4:]) / 4 < 0.8 * sum(losses[:2]) / 2, "the loss does not fall"
    assert sum(ref_losses[-4:]) / 4 < 0.8 * sum(ref_losses[:2]) / 2, "the oracle's loss does not fall: test set-up"
    assert worst <= 0.12, "HIP and oracle training trajectories drift apart"       # observed 3.6e-2 (step 1: 1.4961 vs 1.4474)


def test_config1_two_training_steps():
    """Two full CTClipTrainer.train_step()s on config 1 (EMA codebook update, clip, fused Adam): finite, and the step-0
    loss agrees with the oracle's training-mode step within the free-running bound."""
    from utils.CTClipTrainer import CTClipTrainer
    from oracle import ctclip_oracle as O
    clip, data, cfg = _config1()
    st0 = {k: v.clone() for k, v in clip.state_dict().items()}
    frozen = [k for k in st0 if k.endswith(".beta") or "vq._codebook." in k or not st0[k].is_floating_point()]
    ref_losses, ref_norms, _ = O.train_steps(st0, data, cfg, lr=1.25e-5, max_grad_norm=0.5, frozen=frozen)
    trainer = CTClipTrainer(clip, batch_size=4, results_folder=None)
    for s, (txt, vol) in enumerate(data):
        loss = trainer.train_step((vol, txt))
        rel = abs(loss - ref_losses[s]) / abs(ref_losses[s])
        print(f"  cfg1 step {s}: hip {loss:.6f} oracle {ref_losses[s]:.6f} rel {rel:.2e}; "
              f"grad-norm {trainer.optim.grad_norm():.4f} / {ref_norms[s]:.4f}")
        assert math.isfinite(loss) and rel <= 5e-2      # free-running codes on a 64-token toy: see test_config1_vs_oracle
        assert abs(trainer.optim.grad_norm() - ref_norms[s]) <= 0.1 * ref_norms[s]       # free-running codes


def test_dropped_model_leaves_the_shadow_plan_and_frees_its_memory():
    """ops.PLAN refills every module's weight shadows with one launch per optimiser step; it must hold the sets weakly, or every
    model built in the process stays resident on the GPU and is refilled forever (two trainers in one process, sweeps, eval
    copies).  Build two models, run both once, drop one: its sets leave the plan, its memory is returned, the survivor's forward
    is unchanged."""
    import gc
    from ctclip_hip import ops
    gc.collect()
    torch.cuda.synchronize()
    base_sets, base_mem = len(ops.PLAN.sets), torch.cuda.memory_allocated()
    clip_a, data, _ = _config1()
    clip_b, _, _ = _config1()
    clip_a, clip_b = clip_a.to(DEV).eval(), clip_b.to(DEV).eval()
    txt, vol = data[0]
    txt = {k: v.to(DEV) for k, v in txt.items()}
    with torch.no_grad():
        sim_a = clip_a(txt, vol.to(DEV))[0].clone()
        clip_b(txt, vol.to(DEV))
    both_sets, both_mem = len(ops.PLAN.sets), torch.cuda.memory_allocated()
    per_model = (both_sets - base_sets) // 2
    assert per_model >= 4 and both_sets - base_sets == 2 * per_model
    del clip_b
    gc.collect()
    torch.cuda.synchronize()
    assert len(ops.PLAN.sets) == base_sets + per_model
    freed = both_mem - torch.cuda.memory_allocated()
    assert freed > 0.4 * (both_mem - base_mem - sim_a.numel() * 4), (freed, both_mem - base_mem)
    ops.bump_weight_epoch()                                              # as after an optimiser step: ONE launch over the live sets only
    with torch.no_grad():
        sim_again = clip_a(txt, vol.to(DEV))[0]
    assert torch.equal(sim_a, sim_again)
    del clip_a
    gc.collect()
    assert len(ops.PLAN.sets) == base_sets


def test_backward_joins_the_text_stream_by_itself():
    """The text tower runs (forward and backward) on its own HIP stream and accumulates its gradients in place; `.backward()`
    must leave them complete on the CALLER's stream without anyone calling ops.join_side_streams() -- a custom loop that clips
    or logs gradient norms between backward() and step() is the reference's own pattern (src/utils/CTClipTrainer.py:199-202).
    ops.JoinAfterBackwardFn queues the join as an autograd-engine callback: it is made exactly once per backward, and a norm
    read right after backward() equals the norm read after a device-wide synchronise."""
    from ctclip_hip import ops
    from ctclip_hip.optim import HipAdam
    if not ops._text_stream["on"]:
        pytest.skip("CTCLIP_TEXT_STREAM=0")
    clip, data, _ = _config1()
    clip = clip.to(DEV).train()
    opt = HipAdam([p for p in clip.parameters()], lr=1e-4)
    opt.zero_grad()
    txt, vol = data[0]
    sim = clip({k: v.to(DEV) for k, v in txt.items()}, vol.to(DEV))[0]
    loss = ops.InfoNCEFn.apply(sim)
    calls = []
    real = ops.join_side_streams

    def counted():
        calls.append(torch.cuda.current_stream().cuda_stream)
        return real()
    ops.join_side_streams = counted
    try:
        main = torch.cuda.current_stream().cuda_stream
        loss.backward()
    finally:
        ops.join_side_streams = real
    assert calls == [main], calls                       # once, and with the caller's stream current
    text = [p for n, p in clip.named_parameters() if n.startswith("text_transformer.") and p.grad is not None]
    norm_now = torch.stack([p.grad.float().square().sum() for p in text]).sum().sqrt()
    torch.cuda.synchronize()
    norm_later = torch.stack([p.grad.float().square().sum() for p in text]).sum().sqrt()
    assert float(norm_now) == float(norm_later) and float(norm_now) > 0
