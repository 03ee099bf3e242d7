"""CT-ViT image encoder, MI355X-native.  Drop-in for the reference's `utils.ctvit.CTViT`
(src/utils/ctvit.py:9-125): same constructor, attributes and state-dict keys.

Forward = tubelet patch-embed kernel chain -> 4 spatial transformer layers (576-token sequences, relative
position bias) -> 4 temporal layers (24-token sequences) -> cosine-sim VQ; all HIP (see ctclip_hip.ops).
"""
from pathlib import Path

import torch
from torch import nn
from einops.layers.torch import Rearrange

from ctclip_hip import ops
from ctclip_hip.vq import VectorQuantize
from utils.attention import Attention, Transformer, ContinuousPositionBias, _hooked  # noqa: F401

F32 = torch.float32
BF16 = torch.bfloat16


class CTViT(nn.Module):
    def __init__(self, dim=512, codebook_size=8192, image_size=480, patch_size=20, temporal_patch_size=10,
                 spatial_depth=4, temporal_depth=4, dim_head=64, heads=8, channels=1, attn_dropout=0.0, ff_dropout=0.0,
                 model_type="ctclip"):
        super().__init__()
        self.model_type = model_type
        self.image_size = image_size
        self.patch_size = patch_size
        self.temporal_patch_size = temporal_patch_size
        self.patch_height = image_size // patch_size
        self.patch_width = image_size // patch_size
        self.channels = channels

        self.spatial_rel_pos_bias = ContinuousPositionBias(dim=dim, heads=heads)

        # kept for checkpoint-key compatibility (reference ctvit.py:37-42); only GenerateCT uses it
        self.to_patch_emb_first_frame = nn.Sequential(
            Rearrange("b c 1 (h p1) (w p2) -> b 1 h w (c p1 p2)", p1=patch_size, p2=patch_size),
            nn.LayerNorm(channels * patch_size ** 2),
            nn.Linear(channels * patch_size ** 2, dim),
            nn.LayerNorm(dim),
        )
        feat = channels * patch_size ** 2 * temporal_patch_size
        self.to_patch_emb = nn.Sequential(
            Rearrange("b c (t pt) (h p1) (w p2) -> b t h w (c pt p1 p2)", p1=patch_size, p2=patch_size,
                      pt=temporal_patch_size),
            nn.LayerNorm(feat),
            nn.Linear(feat, dim),
            nn.LayerNorm(dim),
        )
        kw = dict(dim=dim, dim_head=dim_head, heads=heads, attn_dropout=attn_dropout, ff_dropout=ff_dropout, peg=True,
                  peg_causal=True)
        self.enc_spatial_transformer = Transformer(depth=spatial_depth, **kw)
        self.enc_temporal_transformer = Transformer(depth=temporal_depth, **kw)
        self.vq = VectorQuantize(dim=dim, codebook_size=codebook_size, use_cosine_sim=True,
                                 freeze_codebook=not self.training)
        self._shadow = ops.ShadowCache()

    def load(self, path, strict=False):
        path = Path(path)
        if not path.exists():
            raise FileNotFoundError(f"Model state file not found at: {path}")
        try:
            self.load_state_dict(torch.load(str(path), map_location="cpu"), strict)
        except Exception as e:
            raise RuntimeError(f"Failed to load state dictionary from {path}: {e}")

    # -- patch embedding (reference ctvit.py:44-52,112) ------------------------------------------------
    def patch_embed(self, image):
        if not image.is_cuda:
            raise RuntimeError("CTViT: MI355X HIP path only (no CPU fallback); move the volume to cuda")
        _, ln1, lin, ln2 = self.to_patch_emb
        def build():
            # LayerNorm(F)'s gamma / beta folded into the projection (ctclip_patch_affine_fold): the GEMM operand is the plain
            # normalised tubelet row, weight W * gamma (bf16, K padded to 8), bias b + W beta
            N, F_ = lin.weight.shape
            ldw = ops.pad8(F_)
            wg = torch.empty(N, ldw, dtype=BF16, device=lin.weight.device)
            bfold = torch.empty(N, dtype=F32, device=lin.weight.device)
            ops.hip.patch_affine_fold(lin.weight.detach().contiguous(), lin.bias.detach(), ln1.weight.detach(), ln1.bias.detach(),
                                      wg, bfold, N, F_, ldw)
            wT = wg.t().contiguous()
            # wsum[n] = sum_f bf16(W gamma)[n][f]: the mean term of the fused tubelet embedding's epilogue (ctclip_patch_embed_fused)
            return {"w": wg, "wT": wT, "b": bfold, "wsum": ops.colsum(wT),
                    "ones": torch.ones(F_, dtype=F32, device=wg.device), "zeros": torch.zeros(F_, dtype=F32, device=wg.device)}

        sh = self._shadow.get("patch", (lin.weight, lin.bias, ln1.weight, ln1.bias), build)
        if image.dtype not in (F32, BF16):
            image = image.to(F32)
        return ops.PatchEmbedFn.apply(image, ln1.weight, ln1.bias, lin.weight, lin.bias, ln2.weight, ln2.bias, sh,
                                      (self.patch_size, self.temporal_patch_size))

    # -- spatial then temporal transformer (reference ctvit.py:88-103) ---------------------------------
    def encode(self, tokens):
        b, t, h, w, d = tokens.shape
        video_shape = (b, t, h, w)
        attn_bias = self.spatial_rel_pos_bias.lookup(h, w, device=tokens.device)
        x = tokens.reshape(b * t, h * w, d)
        sp, tp = self.enc_spatial_transformer, self.enc_temporal_transformer
        if _hooked(sp) or _hooked(tp):
            # someone watches a transformer's output: keep the reference's layouts and re-order in separate passes
            x = sp(x, attn_bias=attn_bias, video_shape=video_shape)
            x = ops.SwapMiddleFn.apply(x.reshape(b, t, h * w, d))                # (b t)(h w) -> (b h w) t
            x = tp(x.reshape(b * h * w, t, d), video_shape=video_shape)
            x = ops.SwapMiddleFn.apply(x.reshape(b, h * w, t, d))                # back to b t (h w)
            return x.reshape(b, t, h, w, d)
        # the two re-orderings (ctvit.py:96,101) are written by each transformer's final LayerNorm
        x = sp(x, attn_bias=attn_bias, video_shape=video_shape, out_swap=(t, h * w))          # -> [b, h*w, t, d]
        x = tp(x.reshape(b * h * w, t, d), video_shape=video_shape, out_swap=(h * w, t))      # -> [b, t, h*w, d]
        return x.reshape(b, t, h, w, d)

    def forward(self, image, return_only_codebook_ids=False):
        if self.model_type == "ctgenerate":
            raise NotImplementedError("model_type='ctgenerate' (first-frame embedding) is outside the CT-CLIP path")
        tokens = self.patch_embed(image)
        tokens = self.encode(tokens)
        b, t, h, w, d = tokens.shape
        self.vq.train()                                                           # reference ctvit.py:117
        quant, indices, _ = self.vq(tokens.reshape(b, t * h * w, d), freeze_codebook=not self.training)
        if return_only_codebook_ids:
            return indices.reshape(b, t, h, w)
        return quant.reshape(b, t, h, w, d)
