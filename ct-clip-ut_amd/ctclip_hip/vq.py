"""Cosine-similarity VectorQuantize with the buffer names CT-CLIP checkpoints carry
(`vq._codebook.{initted,cluster_size,embed}`).

Stands in for `vector_quantize_pytorch.VectorQuantize(dim, codebook_size, use_cosine_sim=True)` as the
reference constructs it (src/utils/ctvit.py:66) and calls it (ctvit.py:117-118).  The library itself is a
third-party dependency that is not part of the reference tree (version unpinned): the arithmetic follows its
published cosine-sim codebook and is PARITY UNPINNED (see oracle/ctclip_oracle.py:vq_cosine).
The 116 GFLOP/volume nearest-code search runs as an MFMA GEMM with a running arg-max epilogue; the
[tokens, codebook] score matrix is never materialised.
"""
import torch
from torch import nn
import torch.distributed as dist

from . import ops

F32 = torch.float32
BF16 = torch.bfloat16


class CosineSimCodebook(nn.Module):
    def __init__(self, dim, codebook_size, decay=0.8):
        super().__init__()
        self.decay = decay
        embed = torch.empty(1, codebook_size, dim)
        nn.init.kaiming_uniform_(embed)
        embed = embed / embed.norm(dim=-1, keepdim=True).clamp_min(1e-12)
        self.register_buffer("initted", torch.Tensor([True]))
        self.register_buffer("cluster_size", torch.zeros(1, codebook_size))
        self.register_buffer("embed", embed)


class VectorQuantize(nn.Module):
    def __init__(self, dim, codebook_size, use_cosine_sim=True, freeze_codebook=False, decay=0.8, **unused):
        super().__init__()
        if not use_cosine_sim:
            raise NotImplementedError("CT-ViT uses the cosine-similarity codebook (ctvit.py:66)")
        self.dim, self.codebook_size = dim, codebook_size
        self._codebook = CosineSimCodebook(dim, codebook_size, decay)
        self._shadow = ops.ShadowCache()
        # test hook: when set to a LongTensor [b, n] the nearest-code decision is taken from it instead of being
        # searched, so parity tests can separate continuous arithmetic from discrete near-tie flips.
        self.forced_indices = None
        self.last_indices = None
        self._pending_ema = None     # (work handle | None, bins, esum) of a codebook update whose all-reduce is in flight

    def _embed16(self):
        e = self._codebook.embed
        def make():
            S = ops.ShadowSet(e.device)
            e16 = S.zeros(*e.shape[1:])
            S.add(e[0], e16)
            S.out = e16
            return S
        return self._shadow.get_set("e16", (e,), make)

    def forward(self, x, freeze_codebook=False):
        """x [b, n, d] -> (quantised [b,n,d] (straight-through), indices [b,n], commitment loss 0)."""
        if not x.is_cuda:
            raise RuntimeError("VectorQuantize: MI355X HIP path only (no CPU fallback)")
        if self.dim % 8:
            raise ValueError("VectorQuantize: dim must be a multiple of 8")
        self.flush_ema()
        cb = self._codebook
        embed = cb.embed[0]
        quant, idx = ops.VQFn.apply(x.to(F32), embed, self._embed16(), self.forced_indices)
        self.last_indices = idx
        if self.training and not freeze_codebook:
            fn = quant.grad_fn
            x2, inv = (fn.aux if fn is not None and hasattr(fn, "aux") else _renorm(x))
            with torch.no_grad():
                bins, esum, flat = ops.vq_ema_accum(x2, inv, idx, self.codebook_size, self.dim)
                handle = None
                if dist.is_available() and dist.is_initialized():
                    # the library all-reduces bins and embed_sum.  Here: ONE collective over the flat buffer
                    # that holds both, asynchronous -- nothing reads the new codebook before the next forward, so the
                    # 16.8 MB exchange runs under the rest of this step (flush_ema() joins it)
                    handle = dist.all_reduce(flat, async_op=True)
                self._pending_ema = (handle, bins, esum)
                if handle is None:
                    self.flush_ema()
        return quant, idx, torch.zeros((), device=x.device)

    def flush_ema(self):
        """Apply a codebook update whose statistics were (all-)reduced in the background.  Called at the start of the
        next forward and by the trainer at the end of a step, so buffers read in between are always up to date."""
        pend = self._pending_ema
        if pend is None:
            return
        self._pending_ema = None
        handle, bins, esum = pend
        if handle is not None:
            handle.wait()                                          # the current stream now orders after RCCL's
        cb = self._codebook
        with torch.no_grad():
            ops.vq_ema_apply(cb.embed, cb.cluster_size, bins, esum, cb.decay)
            cb.embed.add_(0)                                       # bump the version: bf16 shadow is stale

    def state_dict(self, *args, **kwargs):
        self.flush_ema()
        return super().state_dict(*args, **kwargs)


def _renorm(x):
    from .lib import hip
    x2 = x.detach().to(F32).reshape(-1, x.shape[-1]).contiguous()
    inv = torch.empty(x2.shape[0], dtype=F32, device=x.device)
    hip.rownorm_fwd(x2, None, None, inv, x2.shape[0], x2.shape[1], 1e-12)
    return x2, inv
