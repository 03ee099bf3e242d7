"""Pin the CPU oracle (oracle/ctclip_oracle.py) against vectors produced by the reference.

Vectors: tests/golden/*.npz, written by tests/golden/make_golden.py from the unmodified
reference modules.  Tolerances are fp32 round-off (same math, different op order)."""
import numpy as np
import torch

from conftest import load_golden, sub
from oracle import ctclip_oracle as O

TOL = dict(rtol=2e-4, atol=2e-5)


def close(a, b, **kw):
    tol = {**TOL, **kw}
    torch.testing.assert_close(a, b, **tol)


def test_gammanorm():
    g = load_golden("blocks")
    close(O.gammanorm(g["ln.x"], g["ln.gamma"]), g["ln.y"])


def test_feed_forward_and_grads():
    g = load_golden("blocks")
    st = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sub(g, "ff.").items()
          if k[0].isdigit()}
    x = g["ff.x"].clone().requires_grad_(True)
    y = O.feed_forward(x, st, "")
    close(y, g["ff.y"])
    (y * g["ff.r"]).sum().backward()
    close(x.grad, g["ff.dx"])
    for k in ("0.weight", "0.bias", "1.weight", "4.weight"):
        close(st[k].grad, g["ff.grad." + k])
    assert st["1.weight"].shape[0] // 2 % 2 == 1, "fixture must exercise an odd GEGLU inner size"


def test_peg_memory_order_quirk():
    g = load_golden("blocks")
    shape = tuple(int(v) for v in g["peg.shape"])
    w = g["peg.w"].clone().requires_grad_(True)
    b = g["peg.b"].clone().requires_grad_(True)
    xs = g["peg.xs"].clone().requires_grad_(True)
    xt = g["peg.xt"].clone().requires_grad_(True)
    ys, yt = O.peg(xs, w, b, shape), O.peg(xt, w, b, shape)
    close(ys, g["peg.ys"])
    close(yt, g["peg.yt"])
    ((ys * g["peg.rs"]).sum() + (yt * g["peg.rt"]).sum()).backward()
    close(xs.grad, g["peg.dxs"])
    close(xt.grad, g["peg.dxt"])
    close(w.grad, g["peg.dw"])
    close(b.grad, g["peg.db"])


def test_attention_probs_and_grads():
    g = load_golden("blocks")
    names = ("null_kv", "q_scale", "k_scale", "norm.gamma", "to_q.weight", "to_kv.weight", "to_out.weight")
    st = {k: g["attn." + k].clone().requires_grad_(True) for k in names}
    x = g["attn.x"].clone().requires_grad_(True)
    bias = g["attn.bias"].clone().requires_grad_(True)
    y, probs = O.attention(x, st, "", heads=4, attn_bias=bias)
    close(y, g["attn.y"])
    close(probs, g["attn.probs"])
    (y * g["attn.r"]).sum().backward()
    close(x.grad, g["attn.dx"])
    close(bias.grad, g["attn.dbias"])
    for k in names:
        if ("attn.grad." + k) in g and st[k].numel():
            close(st[k].grad, g["attn.grad." + k])
    y2, p2 = O.attention(x.detach(), {k: v.detach() for k, v in st.items()}, "", heads=4)
    close(y2, g["attn.y_nobias"])
    close(p2, g["attn.probs_nobias"])


def test_continuous_position_bias():
    g = load_golden("blocks")
    st = sub(g, "cpb.")
    close(O.cpb_bias(3, 4, st, ""), g["cpb.bias"])
    # only (2h-1)(2w-1) distinct rows exist; the table form must reproduce the full matrix
    rel = O.cpb_relpos(3, 4).reshape(-1, 2)
    assert len(torch.unique(rel, dim=0)) == (2 * 3 - 1) * (2 * 4 - 1)


def test_transformer_spatial_and_temporal():
    g = load_golden("blocks")
    st = {k: v for k, v in sub(g, "tr.").items() if k.startswith(("layers", "norm_out"))}
    shape = tuple(int(v) for v in g["tr.shape"])
    close(O.transformer(g["tr.xs"], st, "", 2, 4, shape, g["tr.bias"]), g["tr.ys"])
    close(O.transformer(g["tr.xt"], st, "", 2, 4, shape, None), g["tr.yt"])


VIT_CFG = dict(dim=32, codebook_size=64, image_size=16, patch_size=4, temporal_patch_size=2,
               spatial_depth=1, temporal_depth=1, dim_head=8, heads=4)
CLIP_CFG = dict(VIT_CFG, text_layers=2, text_heads=4)


def test_ctvit_glue():
    g = load_golden("ctvit")
    st = sub(g, "sd.")
    vol = g["volume"]
    pt = O.patch_embed(vol, st, "to_patch_emb.", 4, 2)
    close(pt, g["patch_tokens"])
    close(O.ctvit_encode(pt, st, "", VIT_CFG), g["encoded"], rtol=1e-3, atol=1e-4)
    tokens, idx, _, _ = O.ctvit_forward(vol, st, "", VIT_CFG, training=False)
    assert torch.equal(idx, g["indices"])
    close(tokens, g["tokens"], rtol=1e-3, atol=1e-4)


def test_bert_cls_matches_transformers():
    g = load_golden("bert")
    st = sub(g, "sd.")
    hid = O.bert_cls(g["input_ids"], g["token_type_ids"], g["attention_mask"], st, "", 2, 4, return_all=True)
    close(hid, g["last_hidden_state"], rtol=1e-3, atol=1e-4)


def _clip_batches(g):
    out = []
    for s in range(2):
        txt = {k: g[f"step{s}.{k}"] for k in ("input_ids", "token_type_ids", "attention_mask")}
        out.append((txt, g[f"step{s}.volume"]))
    return out


def test_ctclip_eval_forward():
    g = load_golden("ctclip")
    st = sub(g, "sd.")
    (txt, vol), _ = _clip_batches(g)
    o = O.ctclip_forward(txt, vol, st, CLIP_CFG, training=False)
    close(o["sim"], g["eval.sim"], rtol=1e-3, atol=1e-4)
    close(o["image_latents"], g["eval.image_latents"], rtol=1e-3, atol=1e-4)
    close(o["text_latents"], g["eval.text_latents"], rtol=1e-3, atol=1e-4)
    close(o["temp"], g["eval.temp"])
    close(o["image_tokens"], g["eval.image_tokens"], rtol=1e-3, atol=1e-4)


def buffers_of(st):
    return [k for k in st if k.endswith(".beta") or "vq._codebook." in k or not st[k].is_floating_point()]


def test_two_training_steps_loss_gradnorm_weights():
    """CTClipTrainer.train_step order: fwd, symmetric CE, bwd, clip 0.5, Adam (CTClipTrainer.py:181-204)."""
    g = load_golden("ctclip")
    st = sub(g, "sd.")
    losses, norms, final = O.train_steps(st, _clip_batches(g), CLIP_CFG, lr=1.25e-5, max_grad_norm=0.5,
                                         frozen=buffers_of(st))
    for s in range(2):
        assert abs(losses[s] - float(g[f"step{s}.loss"])) <= 1e-4 * abs(float(g[f"step{s}.loss"]))
        assert abs(norms[s] - float(g[f"step{s}.grad_norm"])) <= 2e-3 * float(g[f"step{s}.grad_norm"])
    for k, v in sub(g, "final.").items():
        close(final[k], v, rtol=1e-4, atol=1e-6)


def test_first_step_gradients():
    g = load_golden("ctclip")
    st0 = sub(g, "sd.")
    frozen = set(buffers_of(st0))
    st = {k: (v.clone().requires_grad_(True) if (v.is_floating_point() and k not in frozen) else v)
          for k, v in st0.items()}
    (txt, vol), _ = _clip_batches(g)
    o = O.ctclip_forward(txt, vol, st, CLIP_CFG, training=True)
    O.symmetric_info_nce(o["sim"]).backward()
    ref = sub(g, "step0.grad.")
    assert len(ref) > 40
    # the reference's statically unused parameters receive no gradient
    for k, v in st.items():
        if v.requires_grad and k not in ref:
            assert v.grad is None or float(v.grad.abs().max()) == 0.0, k
    for k, gr in ref.items():
        if gr.numel() == 0:
            continue
        scale = float(gr.abs().max()) + 1e-12
        err = float((st[k].grad - gr).abs().max())
        assert err <= 2e-3 * scale + 1e-7, (k, err, scale)


def test_adam_matches_reference_factory():
    g = load_golden("optimizer")
    for tag, wd, dec in (("adam", 0.0, False), ("adamw", 1e-2, True)):
        w, b = g[f"{tag}.w0"].clone(), g[f"{tag}.b0"].clone()
        mw, vw, mb, vb = (torch.zeros_like(w), torch.zeros_like(w), torch.zeros_like(b), torch.zeros_like(b))
        for s in range(3):
            O.adam_step(w, g[f"{tag}.gw{s}"], mw, vw, s + 1, 1e-2, weight_decay=wd, decoupled=dec)
            # 1-D tensors sit in the no-decay group (optimizer.py:10-11,49-52)
            O.adam_step(b, g[f"{tag}.gb{s}"], mb, vb, s + 1, 1e-2, weight_decay=0.0)
            close(w, g[f"{tag}.w{s+1}"], rtol=1e-5, atol=1e-7)
            close(b, g[f"{tag}.b{s+1}"], rtol=1e-5, atol=1e-7)


# ---- volume ingest and attribution loops, pinned by the reference's own functions -------------
def test_preprocess_resize_crop_and_pipeline_match_reference():
    """tests/golden/preprocess.npz: outputs of the reference's unmodified `resize_array`, `crop_and_pad` and `process_file`
    (src/utils/preprocess.py:20-157, run behind a `nibabel` stand-in that serves synthetic scans)."""
    g = load_golden("preprocess")
    for i in range(3):
        y = O.preprocess_resize_array(g[f"resize{i}.x"], tuple(g[f"resize{i}.cur"].tolist()), tuple(g[f"resize{i}.tgt"].tolist()))
        assert tuple(y.shape) == tuple(g[f"resize{i}.y"].shape)
        close(y, g[f"resize{i}.y"], rtol=1e-6, atol=1e-6)
    for i in range(5):
        y = O.preprocess_crop_and_pad(g[f"crop{i}.x"], tuple(int(v) for v in g[f"crop{i}.tgt"]), pad_value=-1)
        assert torch.equal(y, g[f"crop{i}.y"])
    for i in range(2):
        slope, icpt, xy, z = (float(v) for v in g[f"file{i}.meta"])
        y = O.preprocess_volume(g[f"file{i}.raw"], slope, icpt, xy, z)            # reference targets: 480 x 480 x 240
        assert tuple(y.shape) == (1, 240, 480, 480)
        lo, hi = g[f"file{i}.lo"], g[f"file{i}.hi"]
        box = y[0, lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]]
        close(box, g[f"file{i}.box"], rtol=1e-6, atol=1e-6)
        y[0, lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]] = -1
        assert bool((y == -1).all())                                               # everything else is the pad value


ATTR_CFG = dict(dim=32, codebook_size=512, image_size=16, patch_size=4, temporal_patch_size=2, spatial_depth=1,
                temporal_depth=1, dim_head=8, heads=4, text_layers=2, text_heads=4)


def test_occlusion_and_integrated_gradients_match_reference_loops():
    """tests/golden/attribution.npz: the reference's own `Visualizations._compute_occlusion` (window list, importance
    accumulation, count normalisation, min-max, trilinear resize, threshold, rot90; src/utils/visualizations.py:335-424)
    and the numeric part of `visualize_integrated_gradients` (:851-898) on the reference CTCLIP."""
    import numpy as np
    g = load_golden("attribution")
    st = sub(g, "sd.")
    txt = {k: g[f"txt.{k}"] for k in ("input_ids", "token_type_ids", "attention_mask")}
    patch, stride = tuple(int(v) for v in g["occ.patch"]), tuple(int(v) for v in g["occ.stride"])
    _, count, final = O.occlusion_heatmap(txt, g["image"], st, ATTR_CFG, patch, stride, float(g["occ.threshold"]))
    ref = g["occ.heatmap"].numpy()
    assert final.shape == ref.shape and float(count.max()) == 2.0
    assert np.abs(final - ref).max() <= 2e-4, np.abs(final - ref).max()
    assert ((final == 0) == (ref == 0)).mean() >= 0.999                            # same thresholded support
    _, ig = O.integrated_gradients(txt, g["image"], st, ATTR_CFG, steps=int(g["ig.steps"]))
    ref = g["ig.map"].numpy()
    assert ig.shape == ref.shape
    assert ((ig > 0) == (ref > 0)).mean() >= 0.999                                 # the top-decile mask
    assert np.abs(ig - ref).max() <= 2e-3                                          # ** 0.05 amplifies f32 round-off near 0
