"""CTCLIP contrastive model, MI355X-native.  Drop-in for the reference's `models.ctclip`
(src/models/ctclip.py:10-129): same constructor, attributes, forward 5-tuple and state-dict keys."""
from pathlib import Path

import os

import torch
import torch.distributed as dist
from torch import nn

from ctclip_hip import ops
from ctclip_hip.text import bert_last_hidden_state, is_hf_bert

F32 = torch.float32
BF16 = torch.bfloat16


class GatherWithGrad(torch.autograd.Function):
    """all_gather whose backward keeps only the local slice of the incoming gradient, with NO reduction
    (reference ctclip.py:10-41).  Every rank computes the full global loss, so after gradient averaging the
    encoders see (1/W) * d(global loss) while `temperature` sees the un-scaled gradient."""

    @staticmethod
    def forward(ctx, tensor):
        world = dist.get_world_size()
        t = tensor.contiguous()
        out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        if t.is_cuda and dist.get_backend() == "nccl":
            dist.all_gather_into_tensor(out, t)                       # one RCCL call
        else:                                                          # gloo (CPU tests)
            parts = list(out.chunk(world, 0))
            dist.all_gather(parts, t)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        world = dist.get_world_size()
        return grad_output.chunk(world, dim=0)[dist.get_rank()].contiguous()


class CTCLIP(nn.Module):
    def __init__(self, *, text_encoder, image_encoder, dim_text, dim_image, dim_latent, temperature_init=1.0):
        super().__init__()
        self.text_transformer = text_encoder
        self.visual_transformer = image_encoder
        self.to_text_latent = nn.Linear(dim_text, dim_latent, bias=False)
        self.to_visual_latent = nn.Linear(dim_image, dim_latent, bias=False)
        self.temperature = nn.Parameter(torch.tensor(temperature_init))
        # True: node-global contrastive batch whenever torch.distributed is initialised (the reference's behaviour,
        # ctclip.py:94-97); False: local negatives only (BASELINE config 3).
        self.gather_negatives = True
        self._shadow = ops.ShadowCache()

    def load(self, path, strict=False):
        path = Path(path)
        if not path.exists():
            raise FileNotFoundError(f"Model state file not found at: {path}")
        try:
            state_dict = torch.load(str(path), map_location=torch.device("cuda" if torch.cuda.is_available() else "cpu"))
            self.load_state_dict(state_dict, strict)
        except Exception as e:
            raise RuntimeError(f"Failed to load state dictionary from {path}: {e}")

    def gather_features(self, features):
        if self.gather_negatives and dist.is_available() and dist.is_initialized():
            return GatherWithGrad.apply(features)
        return features

    def encode_text(self, text_inputs):
        tt = self.text_transformer
        if is_hf_bert(tt):
            return bert_last_hidden_state(tt, **text_inputs)[:, 0, :]
        return tt(**text_inputs).last_hidden_state[:, 0, :]

    def forward(self, text_inputs, image_inputs, text_embeds=None):
        fork = ops.fork_text_stream(image_inputs.device) if text_inputs else None
        if fork is not None:
            # the text tower on its own stream next to the image tower (ops.fork_text_stream); its latent follows on that stream
            main, side = fork
            with torch.cuda.stream(side):
                text_output = self.encode_text(text_inputs)
                text_latents = ops.LinearF32Fn.apply(text_output.to(F32), self.to_text_latent.weight, None, False)  # :115
                text_latents = ops.RowNormFn.apply(text_latents)                                  # :119
                if text_latents.requires_grad:
                    text_latents = ops.JoinAfterBackwardFn.apply(text_latents)   # backward() ends with the caller's stream joined
        else:
            text_output = self.encode_text(text_inputs) if text_inputs else text_embeds      # ctclip.py:107
        image_tokens = self.visual_transformer(image_inputs)                                  # :110
        if fork is not None:
            main.wait_stream(side)
            text_latents.record_stream(main)
        if not image_tokens.is_cuda:
            raise RuntimeError("CTCLIP: MI355X HIP path only (no CPU fallback)")
        wv = self.to_visual_latent.weight
        if wv.shape[1] % 8 or wv.shape[0] % 8:
            raise ValueError("dim_image and dim_latent must be multiples of 8 for the bf16 MFMA path")
        def make():                                   # bf16 shadow of the 294 912 -> 512 projection, refilled by ctclip_shadow_multi
            S = ops.ShadowSet(wv.device)
            w16 = S.zeros(*wv.shape)
            S.add(wv, w16)
            S.out = w16
            return S
        wv16 = self._shadow.get_set("wv", (wv,), make)
        image_latents = ops.VisualLatentFn.apply(image_tokens.to(F32), wv, wv16)              # :111-112,116
        if fork is None:
            text_latents = ops.LinearF32Fn.apply(text_output.to(F32), self.to_text_latent.weight, None, False)  # :115
            text_latents = ops.RowNormFn.apply(text_latents)                                  # :119
        image_latents = ops.RowNormFn.apply(image_latents)                                    # :120
        if self.gather_negatives and dist.is_available() and dist.is_initialized():
            # :123-124 as ONE collective: the reference gathers text then image latents with two all_gathers; both are
            # [B, dim_latent] and latency-bound, so they travel side by side in one [B, 2*dim_latent] buffer.  Row r of
            # the gathered buffer is still rank-major pair r, and GatherWithGrad's backward (local slice, no reduce)
            # acts on rows, so values and gradients are those of the two separate calls.
            L = text_latents.shape[1]
            both = GatherWithGrad.apply(torch.cat((text_latents, image_latents), dim=1))
            text_latents, image_latents = both[:, :L].contiguous(), both[:, L:].contiguous()
        sim = ops.SimMatrixFn.apply(image_latents, text_latents, self.temperature)            # :127
        return sim, image_latents, text_latents, self.temperature.exp(), image_tokens
