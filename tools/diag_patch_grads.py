#!/usr/bin/env python3
"""PatchEmbedFn with the fused tubelet kernels against the unfused chain: outputs and every parameter gradient, small and production
geometry.   python3 tools/diag_patch_grads.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ct-clip-ut_amd")):
    sys.path.insert(0, p)
import torch
from ctclip_hip import ops
from utils.ctvit import CTViT

for cfg, B, D in ((dict(dim=512, codebook_size=64, image_size=64, patch_size=16, temporal_patch_size=16, spatial_depth=1, temporal_depth=1,
                        dim_head=32, heads=8), 4, 64),
                  (dict(dim=512, codebook_size=64, image_size=480, patch_size=20, temporal_patch_size=10, spatial_depth=1, temporal_depth=1,
                        dim_head=32, heads=8), 2, 240)):
    torch.manual_seed(0)
    vit = CTViT(**cfg).cuda()
    S = cfg["image_size"]
    vol = (torch.randn(B, 1, D, S, S, device="cuda") * 0.5).clamp(-1, 1).to(torch.bfloat16)
    res = {}
    for fused in (False, True):
        ops.PATCH_FUSED = fused
        vit.zero_grad(set_to_none=True)
        y = vit.patch_embed(vol)
        g = torch.Generator(device="cuda").manual_seed(1)
        dy = torch.randn(y.shape, device="cuda", generator=g)
        y.backward(dy)
        res[fused] = (y.detach().clone(), {n: p.grad.clone() for n, p in vit.to_patch_emb.named_parameters()})
    ya, yb = res[False][0], res[True][0]
    print(f"geometry {S}^2 x {D}, {B} volumes: output rel diff fused vs chain {float((ya - yb).norm() / ya.norm()):.3e}")
    for n in res[False][1]:
        a, b = res[False][1][n], res[True][1][n]
        print(f"   d({n}): |chain| {float(a.norm()):.4e} |fused| {float(b.norm()):.4e} rel diff {float((a - b).norm() / a.norm()):.3e}")
