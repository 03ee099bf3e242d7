"""GPU parity at the PRODUCTION geometry (BASELINE.json configs[1], reference src/train_ctclip.py:19-37): CT-ViT base
(dim 512, 4+4 layers, 8x32 heads, patch 20 / tubelet 10, codebook 8192) + BERT-base-shaped text encoder + 294 912 -> 512
visual projection, two 480x480x240 volumes with 128-token reports, against the f32 CPU oracle on the same weights and
inputs.  Everything the toy-sized parity tests cannot reach runs here: `patch_ln_fwd_fast / bwd_fast` with p=20, pt=10,
W=480 (the multiply-high divisors for 20 / 200 / 30 / 500), the K=4000 patch GEMM, 576-token spatial attention with the
2209-row position table, the 8192-code `vq_topk3` sweep, the split-K visual projection over 294 912 features.

Bars (stated per check): contrastive loss within 1e-3 relative of the oracle with the VQ code decisions pinned to the
oracle's (the north-star bar; isolates the continuous arithmetic), stage outputs within 3e-2..5e-2 of peak, gradients by
`grad_parity` (per significant tensor 8e-2, global cosine > 0.999); free-running (codes searched by the HIP kernels on
bf16-noisy encoder outputs) the code agreement rate and the loss delta are MEASURED, printed and bounded.

The oracle forward+backward at this size is ~4.5 TFLOP of f32 CPU work and ~25 GB of autograd state: a few minutes on the
GPU box's 16-core share.
"""
import math
import os
import time

import pytest
import torch

from test_hip_model import check, grad_parity

pytestmark = pytest.mark.gpu
DEV = "cuda"

VIT = dict(dim=512, codebook_size=8192, image_size=480, patch_size=20, temporal_patch_size=10, spatial_depth=4,
           temporal_depth=4, dim_head=32, heads=8)
TEXT = dict(hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072, vocab_size=30522,
            max_position_embeddings=512, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
B, L, DEPTH, SIZE = 2, 128, 240, 480


def build_production_model():
    from transformers import BertConfig, BertModel
    from models.ctclip import CTCLIP
    from utils.ctvit import CTViT
    torch.manual_seed(0)
    grid = VIT["image_size"] // VIT["patch_size"]
    clip = CTCLIP(text_encoder=BertModel(BertConfig(**TEXT)), image_encoder=CTViT(**VIT), dim_text=TEXT["hidden_size"],
                  dim_image=grid * grid * VIT["dim"], dim_latent=512)
    # default inits leave every LayerNorm at (1, 0) and every bias at 0, which hides scale / shift bugs: perturb them
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for n, p in clip.named_parameters():
            if p.ndim == 1 and p.numel() > 0 and "null_kv" not in n:
                if n.endswith(("gamma", "q_scale", "k_scale")) or ("norm" in n.lower() and n.endswith("weight")) \
                        or n.endswith(("to_patch_emb.1.weight", "to_patch_emb.3.weight", ".3.0.weight")):
                    p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
                else:
                    p.copy_(0.1 * torch.randn(p.shape, generator=g))
    return clip


def production_batch():
    g = torch.Generator().manual_seed(1234)
    vol = (torch.randn(B, 1, DEPTH, SIZE, SIZE, generator=g) * 0.5).clamp_(-1, 1)
    vol = vol.to(torch.bfloat16).to(torch.float32)          # the benchmark feeds bf16 volumes: both sides see these values
    ids = torch.randint(0, TEXT["vocab_size"], (B, L), generator=g)
    lens = torch.tensor([L, 57])
    mask = (torch.arange(L)[None] < lens[:, None]).long()
    return vol, {"input_ids": ids, "token_type_ids": torch.zeros_like(ids), "attention_mask": mask}


def oracle_step(st0, vol, txt, threads=None):
    """f32 CPU oracle: forward (eval-mode VQ: frozen codebook) + backward of the symmetric InfoNCE loss.
    -> (loss, out dict (detached), {name: grad}, stage tensors)."""
    from oracle import ctclip_oracle as O
    if threads:
        torch.set_num_threads(threads)
    cfg = dict(VIT, text_layers=TEXT["num_hidden_layers"], text_heads=TEXT["num_attention_heads"])
    frozen = {k for k in st0 if k.endswith(".beta") or "vq._codebook." in k or not st0[k].is_floating_point()}
    st = {k: (v.clone().requires_grad_(True) if (v.is_floating_point() and k not in frozen and v.numel()) else v)
          for k, v in st0.items()}
    t0 = time.time()
    P = "visual_transformer."
    with torch.no_grad():
        patch_tokens = O.patch_embed(vol[:1], st0, P + "to_patch_emb.", VIT["patch_size"], VIT["temporal_patch_size"])
    out = O.ctclip_forward(txt, vol, st, cfg, training=False)
    loss = O.symmetric_info_nce(out["sim"])
    t1 = time.time()
    loss.backward()
    t2 = time.time()
    print(f"  oracle: forward {t1 - t0:.1f} s, backward {t2 - t1:.1f} s on {torch.get_num_threads()} threads")
    grads = {k: v.grad for k, v in st.items() if isinstance(v, torch.Tensor) and v.requires_grad and v.grad is not None}
    keep = {k: out[k].detach() for k in ("sim", "image_latents", "text_latents", "indices", "text_cls", "image_tokens")}
    return float(loss.detach()), keep, grads, {"patch_tokens0": patch_tokens}


def test_production_geometry_loss_gradients_and_code_flips():
    from ctclip_hip import ops
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    clip = build_production_model()
    st0 = {k: v.detach().clone() for k, v in clip.state_dict().items()}
    vol, txt = production_batch()
    loss_o, out_o, grads_o, stages = oracle_step(st0, vol, txt, threads=max(1, min(16, avail)))
    print(f"  oracle loss {loss_o:.6f}; sim {out_o['sim'].flatten().tolist()}")

    clip = clip.to(DEV)
    vit, vq = clip.visual_transformer, clip.visual_transformer.vq
    vold = vol.to(DEV).to(torch.bfloat16)
    txd = {k: v.to(DEV) for k, v in txt.items()}

    # ---- stage checks on the kernels only this geometry reaches -------------------------------------------------------
    clip.eval()
    with torch.no_grad():
        pt = vit.patch_embed(vold[:1])
        check("tubelet embedding p=20 pt=10 (LN 4000 -> 4000x512 GEMM -> LN 512)", pt, stages["patch_tokens0"], 3e-2)
        cls = clip.encode_text(txd)
        check("BERT-base CLS at L=128 with padding", cls, out_o["text_cls"], 3e-2)

    # ---- (2) free running: what a user gets ---------------------------------------------------------------------------
    with torch.no_grad():
        sim, il, tl, _, toks = clip(txd, vold)
        free = float(ops.InfoNCEFn.apply(sim))
    idx_h = vq.last_indices.reshape(-1).cpu()
    idx_o = out_o["indices"].reshape(-1)
    agree = float((idx_h == idx_o).float().mean())
    flips = int((idx_h != idx_o).sum())
    rel_free = abs(free - loss_o) / abs(loss_o)
    print(f"  FREE-RUNNING: loss {free:.6f} vs oracle {loss_o:.6f} (rel {rel_free:.3e}); code agreement {agree:.5f} "
          f"({flips} of {idx_o.numel()} tokens x 8192 codes flipped)")
    # each flipped code is an unrelated unit vector in the mean-over-depth features: measured 1.1e-1 of peak at 0.7 % flips
    check("image latents (free-running codes)", il, out_o["image_latents"], 2e-1)
    check("text latents", tl, out_o["text_latents"], 3e-2)
    assert agree >= 0.97, "more than 3 % of the nearest-code decisions differ from the f32 oracle"
    # ~190 of 27 648 nearest-code decisions are genuine near-ties that bf16 noise flips; WHICH ones depends on the rounding
    # realisation, and with two pairs in the batch the loss moves with them: the same inputs through the four combinations of the
    # round-5 kernel paths (tubelet embedding fused / unfused, head-norm in the GEMM epilogue / separate) read 6.2e-4, 2.4e-3,
    # 3.4e-3 and 5.2e-3 at 186-197 flips (profiles/r05_production_free_running.txt).  The 1e-3 bar is the pinned-code one below.
    assert rel_free <= 1.2e-2, "free-running loss further than 1.2e-2 from the oracle"

    # ---- (1) pinned codes: the 1e-3 bar, then gradients ---------------------------------------------------------------
    clip.train()
    vit.eval()                                   # oracle ran with a frozen codebook (training=False)
    vq.forced_indices = out_o["indices"].reshape(B, -1)
    for p in clip.parameters():
        p.grad = None
    sim, il, tl, _, toks = clip(txd, vold)
    loss = ops.InfoNCEFn.apply(sim)
    rel = abs(float(loss) - loss_o) / abs(loss_o)
    print(f"  PINNED CODES: loss {float(loss):.6f} vs oracle {loss_o:.6f} rel {rel:.3e} (bar 1e-3)")
    check("image tokens (pinned codes: gathered codebook rows)", toks, out_o["image_tokens"], 1e-5)
    check("image latents (split-K over 294 912)", il, out_o["image_latents"], 2e-2)
    check("similarity matrix", sim, out_o["sim"], 2e-2)
    assert rel <= 1e-3
    loss.backward()
    torch.cuda.synchronize()
    vq.forced_indices = None
    named = dict(clip.named_parameters())
    grad_parity(named, grads_o, 8e-2, "production-geometry gradients vs oracle (pinned codes)")
    # the statically unused parameters get no gradient on either side
    for k, p in named.items():
        if k not in grads_o and p.numel():
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k


@pytest.mark.skipif(bool(os.environ.get("CTCLIP_TEST_PRODUCT_GATES")), reason="this IS the product-gates child run")
def test_parity_suites_with_the_training_runs_gemm_gates():
    """The suite's default hook (CTCLIP_GEMM_V2_ALL, tests/conftest.py) widens the GEMM size gates, so in this process the
    production-geometry model sends FF1 + GEGLU (K = 512) and BERT's K = 768 / N = 3072 products partly to other kernels than a
    training run does.  The library reads the hook once per process, so the model / production parity tests are run once more
    in a fresh child process with the gates a training run has (CTCLIP_TEST_PRODUCT_GATES=1: gemm5 takes act == GEGLU and
    N >= 2048, gemm3 the rest; reference src/utils/attention.py:38-51).  One child at a time, started after this process's own
    GPU work is idle."""
    import subprocess
    import sys
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CTCLIP_TEST_PRODUCT_GATES="1")
    env.pop("CTCLIP_GEMM_V2_ALL", None)
    cmd = [sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_hip_production.py"),
           os.path.join(root, "tests", "test_hip_model.py"), "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"]
    r = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
    tail = "\n".join((r.stdout or "").splitlines()[-15:])
    print(tail)
    assert r.returncode == 0, f"product-gates child run failed:\n{tail}\n{(r.stderr or '')[-2000:]}"
    assert " passed" in tail and "failed" not in tail
