"""GPU parity on the EDGE geometries of the training step (reference src/models/ctclip.py:99-129, src/utils/ctvit.py:88-125):
a single pair (1 x 1 similarity matrix), a single time step (depth = one tubelet), a 2 x 2 patch grid, three frames, a
one-token report, an odd head count, a codebook smaller than one MFMA tile row block -- every size at which a tile, a strip
or a reduction is ragged or degenerate.  Against the f32 CPU oracle on the same weights and inputs, nearest-code decisions
pinned to the oracle's: similarity matrix and latents within 3e-2 of peak, loss within 1e-3 relative (absolute 1e-6 where the
loss itself is 0), gradients by `grad_parity` at 8e-2 (bf16 operands, tiny tensors)."""
import pytest
import torch

from test_hip_model import check, grad_parity

pytestmark = pytest.mark.gpu
DEV = "cuda"

BASE_VIT = dict(dim=64, codebook_size=256, image_size=64, patch_size=16, temporal_patch_size=16, spatial_depth=1,
                temporal_depth=1, dim_head=32, heads=2)
CASES = {
    # name: (vit overrides, batch, depth, report length)
    "single_pair": ({}, 1, 32, 16),
    "one_time_step": ({}, 3, 16, 16),
    "grid_2x2": (dict(image_size=32), 2, 32, 16),
    "three_frames": ({}, 2, 48, 16),
    "one_token_report": ({}, 2, 32, 1),
    "odd_head_count": (dict(dim=96, heads=3), 2, 32, 16),
    "small_codebook": (dict(codebook_size=40), 2, 32, 16),
    "narrow_heads": (dict(dim=48, dim_head=8, heads=3), 3, 32, 7),
}


def build(vit_over, batch, depth, L, seed):
    from transformers import BertConfig, BertModel
    from models.ctclip import CTCLIP
    from utils.ctvit import CTViT
    torch.manual_seed(seed)
    vit = dict(BASE_VIT, **vit_over)
    bcfg = dict(hidden_size=64, num_hidden_layers=1, num_attention_heads=2, intermediate_size=128, vocab_size=101,
                max_position_embeddings=32, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    grid = vit["image_size"] // vit["patch_size"]
    clip = CTCLIP(text_encoder=BertModel(BertConfig(**bcfg)), image_encoder=CTViT(**vit), dim_text=64,
                  dim_image=grid * grid * vit["dim"], dim_latent=32)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():                                     # default inits hide scale / shift bugs
        for n, p in clip.named_parameters():
            if p.ndim == 1 and p.numel() > 0 and "null_kv" not in n:
                p.add_(0.1 * torch.randn(p.shape, generator=g))
    vol = (torch.randn(batch, 1, depth, vit["image_size"], vit["image_size"], generator=g) * 0.5).clamp(-1, 1)
    ids = torch.randint(0, 101, (batch, L), generator=g)
    lens = torch.randint(1, L + 1, (batch,), generator=g)
    mask = (torch.arange(L)[None] < lens[:, None]).long()
    txt = {"input_ids": ids, "token_type_ids": torch.zeros_like(ids), "attention_mask": mask}
    return clip, txt, vol, dict(vit, text_layers=1, text_heads=2)


@pytest.mark.parametrize("name", list(CASES))
def test_edge_geometry_vs_oracle(name):
    from ctclip_hip import ops
    from oracle import ctclip_oracle as O
    over, batch, depth, L = CASES[name]
    clip, txt, vol, cfg = build(over, batch, depth, L, seed=sum(map(ord, name)))
    st0 = {k: v.clone() for k, v in clip.state_dict().items()}
    frozen = {k for k in st0 if k.endswith(".beta") or "vq._codebook." in k or not st0[k].is_floating_point()}
    sto = {k: (v.clone().requires_grad_(True) if (v.is_floating_point() and k not in frozen and v.numel()) else v)
           for k, v in st0.items()}
    out_o = O.ctclip_forward(txt, vol, sto, cfg, training=False)
    loss_o = O.symmetric_info_nce(out_o["sim"])
    loss_o.backward()

    clip = clip.to(DEV).train()
    clip.visual_transformer.eval()                             # frozen codebook, like the oracle's training=False
    vq = clip.visual_transformer.vq
    vq.forced_indices = out_o["indices"].reshape(batch, -1)
    txd = {k: v.to(DEV) for k, v in txt.items()}
    sim, il, tl, *_ = clip(txd, vol.to(DEV))
    check(f"{name}: similarity matrix", sim, out_o["sim"], 3e-2)
    check(f"{name}: image latents", il, out_o["image_latents"], 3e-2)
    check(f"{name}: text latents", tl, out_o["text_latents"], 3e-2)
    loss = ops.InfoNCEFn.apply(sim)
    lo = float(loss_o.detach())
    print(f"  {name}: loss {float(loss):.6f} vs oracle {lo:.6f}")
    assert abs(float(loss) - lo) <= max(1e-3 * abs(lo), 1e-6)
    loss.backward()
    grads_o = {k: v.grad for k, v in sto.items() if isinstance(v, torch.Tensor) and v.requires_grad}
    if batch == 1:
        # one pair: the loss is log(1) = 0 whatever the weights are, and so is every gradient -- on both sides
        worst = max(float(p.grad.abs().max()) for p in clip.parameters() if p.grad is not None)
        assert abs(lo) <= 1e-6 and worst <= 1e-6, (lo, worst)
    else:
        grad_parity(dict(clip.named_parameters()), grads_o, 8e-2, f"{name}: gradients vs oracle (pinned codes)")
    # free running on the same inputs: the HIP nearest-code search must agree with the oracle's on (nearly) every token
    vq.forced_indices = None
    with torch.no_grad():
        clip(txd, vol.to(DEV))
    agree = float((vq.last_indices.reshape(-1).cpu() == out_o["indices"].reshape(-1)).float().mean())
    print(f"  {name}: free-running code agreement {agree:.4f}")
    assert agree >= 0.9
