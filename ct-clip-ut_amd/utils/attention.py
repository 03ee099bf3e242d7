"""Transformer building blocks of CT-ViT, MI355X-native.

Drop-in surface for the reference's `utils.attention` (src/utils/attention.py): same class names,
constructor signatures, sub-module / parameter names (so reference checkpoints load key-for-key) and
call conventions.  The arithmetic runs in hand-written gfx950 kernels through `ctclip_hip`; the modules
only own parameters and sequence the launches.

Implemented: the branches CT-CLIP uses (self-attention, no null-kv, no mask, non-causal, PEG causal in t).
Branches the reference carries for GenerateCT only (cross-attention context, causal ALiBi, null key/values,
attention masks, dropout > 0) raise NotImplementedError rather than silently computing something else.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import nn

from ctclip_hip import ops
from ctclip_hip.lib import hip

F32 = torch.float32
BF16 = torch.bfloat16


def exists(v):
    return v is not None


def _need_cuda(x: torch.Tensor, who: str) -> None:
    if not x.is_cuda:
        raise RuntimeError(f"{who}: the CT-CLIP hot path runs on MI355X HIP kernels only; got a {x.device} tensor "
                           "(there is no CPU fallback -- the CPU restatement under oracle/ is test infrastructure)")


def _hooked(m: nn.Module) -> bool:
    return bool(m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or m._backward_pre_hooks)


class LayerNorm(nn.Module):
    """Bias-less LayerNorm (reference attention.py:27-34): learnable `gamma`, zero `beta` buffer."""

    def __init__(self, dim):
        super().__init__()
        self.gamma = nn.Parameter(torch.ones(dim))
        self.register_buffer("beta", torch.zeros(dim))

    def forward(self, x):
        _need_cuda(x, "LayerNorm")
        return ops.LayerNormFn.apply(x.to(F32), self.gamma, None, 1e-5)

    def forward_swapped(self, x, a_ext, c_ext):
        """The same normalisation of rows laid out [B][a_ext][c_ext], returned as [B, c_ext, a_ext, d]."""
        _need_cuda(x, "LayerNorm")
        return ops.LayerNormFn.apply(x.to(F32), self.gamma, None, 1e-5, (int(a_ext), int(c_ext)))


class GEGLU(nn.Module):
    """Placeholder that keeps the reference's Sequential indices (attention.py:38-41); the gate is fused
    into FeedForward's kernels and this module is never called on the hot path."""

    def forward(self, x):
        raise RuntimeError("GEGLU is fused into FeedForward.forward; call the FeedForward module instead")


class FeedForward(nn.Sequential):
    """LN -> Linear(dim, 2*inner, no bias) -> GEGLU -> Dropout -> Linear(inner, dim, no bias).

    Same children / indices as the reference factory (attention.py:43-51): keys 0.weight, 0.bias, 1.weight,
    4.weight.  inner = int(mult * 2/3 * dim)."""

    def __init__(self, dim, mult=4, dropout=0.0):
        inner = int(mult * (2 / 3) * dim)
        super().__init__(nn.LayerNorm(dim), nn.Linear(dim, inner * 2, bias=False), GEGLU(), nn.Dropout(dropout),
                         nn.Linear(inner, dim, bias=False))
        self.dim, self.inner, self.dropout = dim, inner, dropout
        self._shadow = ops.ShadowCache()

    def _shadows(self):
        w1, w2 = self[1].weight, self[4].weight
        I, Ip = self.inner, ops.pad64(self.inner)      # 1365 -> 1408: every K of the MLP is a whole number of 64-wide k-tiles

        def make():
            # value rows 0..I-1 and gate rows I..2I-1 of the reference weight, each zero-padded to Ip and then interleaved
            # in 32-row blocks [val 32 | gate 32 | val 32 | ...] (ops.GEGLU_BLOCK): each wave's 64-column slab of the GEMM
            # tile then holds a value block and its gate block and the GEGLU is applied in the GEMM epilogue, straight from
            # the accumulator registers (ctclip_gemm_bf16_geglu).  Filled by ctclip_shadow_multi (pad rows / columns stay zero).
            blk = ops.GEGLU_BLOCK
            S = ops.ShadowSet(w1.device)
            w1p, w1T = S.zeros(2 * Ip, self.dim), S.zeros(self.dim, 2 * Ip)
            w2p, w2T = S.zeros(self.dim, Ip), S.zeros(Ip, self.dim)
            S.add(w1[:I], w1p, blk=blk)
            S.add(w1[I:], w1p[blk:], blk=blk)
            S.add(w1[:I], w1T, transpose=True, blk=blk)
            S.add(w1[I:], w1T[:, blk:], transpose=True, blk=blk)
            S.add(w2, w2p)
            S.add(w2, w2T, transpose=True)
            S.out = {"w1": w1p, "w2": w2p, "w1T": w1T, "w2T": w2T, "inner": I, "inner_p": Ip}
            return S

        return self._shadow.get_set("ff", (w1, w2), make)

    def forward(self, x, residual: bool = False):
        _need_cuda(x, "FeedForward")
        if self.dropout > 0 and self.training:
            raise NotImplementedError("ff_dropout > 0 is outside the CT-CLIP configuration (train_ctclip.py:19-29)")
        if self.dim % 8:
            raise ValueError("FeedForward: dim must be a multiple of 8 for the bf16 MFMA path")
        return ops.FeedForwardFn.apply(x.to(F32), self[0].weight, self[0].bias, self[1].weight, self[4].weight,
                                       self._shadows(), residual)


class PEG(nn.Module):
    """Position-generating depthwise conv (reference attention.py:55-83), on the memory order of x."""

    def __init__(self, dim, causal=False):
        super().__init__()
        self.causal = causal
        self.dsconv = nn.Conv3d(dim, dim, 3, groups=dim)

    def forward(self, x, shape: Tuple[int, int, int, int] = None, residual: bool = False):
        _need_cuda(x, "PEG")
        if not self.causal:
            raise NotImplementedError("non-causal PEG is not used by CT-ViT (ctvit.py:54-62 sets peg_causal=True)")
        if x.ndim == 3 and shape is None:
            raise ValueError("PEG needs `shape` for [b, n, d] input")
        if x.ndim == 5:
            shape = x.shape[:4]
        # the Function works on flat memory and returns the input's shape: no view nodes in the autograd graph, so the
        # bf16 gradient shadow tagged by its backward reaches the previous layer's backward intact
        y, y16 = ops.PegFn.apply(x.to(F32), self.dsconv.weight, self.dsconv.bias, tuple(int(s) for s in shape), residual)
        self._last_bf16 = y16                      # bf16 copy of the output, written by the same kernel (fused path only)
        return y


class PositionBias:
    """bias[h,i,j] = table[relidx[i,j], h]: the de-duplicated form ContinuousPositionBias.lookup() returns.

    `table` [R, heads] carries the autograd graph of the position MLP, `dense` is its [heads,n,n] expansion
    (what the reference materialises, attention.py:277); attention backward reduces d(bias) on chip into
    d(table)."""

    def __init__(self, table, relidx, dense, grid=None):
        self.table, self.relidx, self.dense, self.grid = table, relidx, dense, grid
        self.rows = table.shape[0]


class Attention(nn.Module):
    """Cosine-sim attention (reference attention.py:87-182)."""

    def __init__(self, dim, dim_context=None, dim_head=64, heads=8, causal=False, num_null_kv=0, norm_context=True,
                 dropout=0.0, scale=8):
        super().__init__()
        self.heads, self.causal, self.scale = heads, causal, scale
        self.dim, self.dim_head, self.dropout = dim, dim_head, dropout
        inner = dim_head * heads
        dim_context = dim if dim_context is None else dim_context
        if causal:
            raise NotImplementedError("causal attention (ALiBi) belongs to GenerateCT, not to the CT-CLIP path")
        self.attn_dropout = nn.Dropout(dropout)
        self.norm = LayerNorm(dim)
        self.context_norm = LayerNorm(dim_context) if norm_context else nn.Identity()
        self.num_null_kv = num_null_kv
        self.null_kv = nn.Parameter(torch.randn(heads, 2 * num_null_kv, dim_head))
        self.to_q = nn.Linear(dim, inner, bias=False)
        self.to_kv = nn.Linear(dim_context, inner * 2, bias=False)
        self.q_scale = nn.Parameter(torch.ones(dim_head))
        self.k_scale = nn.Parameter(torch.ones(dim_head))
        self.to_out = nn.Linear(inner, dim, bias=False)
        self.return_attn = False       # set True to materialise the [b,h,n,n] probabilities (attention.py:182)
        self._shadow = ops.ShadowCache()

    def _shadows(self):
        H, dh = self.heads, self.dim_head
        dp = ops.head_pad(dh)
        wq, wkv, wo, qs, ks = self.to_q.weight, self.to_kv.weight, self.to_out.weight, self.q_scale, self.k_scale
        gamma = self.norm.gamma

        def build():                                  # head sizes that need zero rows per head (dh != dp): framework ops
            inner = H * dh
            k_w, v_w = wkv[:inner], wkv[inner:]
            pad = lambda w: ops.pad_head_rows(w, H, dh, dp)
            scale_p = lambda s: torch.cat((s.detach().to(F32), s.new_zeros(dp - dh)))
            d = {
                "wq": pad(wq).to(BF16).contiguous(),
                "wkv": torch.cat((pad(k_w), pad(v_w)), 0).to(BF16).contiguous(),
                "wout": pad(wo.t()).t().to(BF16).contiguous(),
                "q_scale": scale_p(qs), "k_scale": scale_p(ks),
            }
            # transposed copies of the (small) weights: dgrad then has the forward's k-major x k-major layout
            d.update(wqT=d["wq"].t().contiguous(), wkvT=d["wkv"].t().contiguous(), woutT=d["wout"].t().contiguous())
            return d

        def make():                                   # production head size: one descriptor table, no framework ops
            inner, dim = H * dh, self.dim
            S = ops.ShadowSet(wq.device)
            o = {"wq": S.zeros(inner, dim), "wkv": S.zeros(2 * inner, dim), "wout": S.zeros(dim, inner),
                 "wqT": S.zeros(dim, inner), "wkvT": S.zeros(dim, 2 * inner), "woutT": S.zeros(inner, dim),
                 "q_scale": S.zeros(dp, dtype=F32), "k_scale": S.zeros(dp, dtype=F32),
                 # the LayerNorm's gamma folded into the q projection (q = xhat (Wq gamma)^T, attention.py:140,142): the GEMM
                 # operand is the plain normalised row, which is then all the LayerNorm backward needs (ops.AttentionFn);
                 # wcat = [Wqg^T | Wkv^T] is the one k-major operand of the block's input gradient
                 # dx = [rstd dq | dkv] [Wqg ; Wkv] (ctclip_gemm_bf16_lnbwd), wbar the row sums of Wqg its row constants need
                 "wqg": S.zeros(inner, dim), "wqgT": S.zeros(dim, inner), "wcat": S.zeros(dim, 3 * inner),
                 "wbar": S.zeros(inner, dtype=F32)}
            S.add(wq, o["wq"]); S.add(wq, o["wqT"], transpose=True)
            S.add(wkv, o["wkv"]); S.add(wkv, o["wkvT"], transpose=True)
            S.add(wo, o["wout"]); S.add(wo, o["woutT"], transpose=True)
            S.add(qs, o["q_scale"]); S.add(ks, o["k_scale"])
            S.add(wq, o["wqg"], scale=gamma); S.add(wq, o["wqgT"], transpose=True, scale=gamma)
            S.add(wq, o["wcat"], transpose=True, scale=gamma); S.add(wkv, o["wcat"][:, inner:], transpose=True)
            S.add(wq, o["wbar"], rowsum=True, scale=gamma)
            S.out = o
            return S

        if dp == dh:
            return self._shadow.get_set("attn", (wq, wkv, wo, qs, ks, gamma), make)
        return self._shadow.get("attn", (wq, wkv, wo, qs, ks, gamma), build)

    def forward(self, x, mask=None, context=None, attn_bias=None, residual: bool = False, x16=None):
        _need_cuda(x, "Attention")
        if exists(context) or exists(mask) or self.num_null_kv > 0:
            raise NotImplementedError("cross-attention context / masks / null key-values are GenerateCT-only branches")
        if self.dropout > 0 and self.training:
            raise NotImplementedError("attn_dropout > 0 is outside the CT-CLIP configuration")
        if self.dim % 8:
            raise ValueError("Attention: dim must be a multiple of 8 for the bf16 MFMA path")
        dh = self.dim_head
        cfg = (self.heads, dh, ops.head_pad(dh), float(self.scale), residual, bool(self.return_attn))
        if attn_bias is None:
            bias_t, aux = None, {"kind": None}
        elif isinstance(attn_bias, PositionBias):
            bias_t = attn_bias.table
            aux = {"kind": "table", "dense": attn_bias.dense, "relidx": attn_bias.relidx, "rows": attn_bias.rows,
                   "grid": attn_bias.grid or (0, 0)}
        else:
            bias_t, aux = attn_bias, {"kind": "dense"}
        aux["x16"] = x16                            # optional bf16 copy of x from the producing kernel (saves a cast pass)
        y, probs = ops.AttentionFn.apply(x.to(F32), self.norm.gamma, self.to_q.weight, self.to_kv.weight, self.q_scale,
                                         self.k_scale, self.to_out.weight, bias_t, self._shadows(), cfg, aux)
        return y, (probs if self.return_attn else None)


class ContinuousPositionBias(nn.Module):
    """Log-distance relative position bias MLP (reference attention.py:230-277), f32.

    The reference evaluates the MLP on all (h*w)^2 pairs every forward (177 GFLOP at 24x24).  Only
    (2h-1)(2w-1) distinct inputs exist, so this module evaluates the MLP on those rows and gathers;
    every output value is produced by the same row-wise arithmetic."""

    def __init__(self, *, dim, heads, num_dims=2, layers=2, log_dist=True, cache_rel_pos=False):
        super().__init__()
        self.num_dims, self.log_dist, self.heads = num_dims, log_dist, heads
        self.net = nn.ModuleList([])
        self.net.append(nn.Sequential(nn.Linear(num_dims, dim), nn.LeakyReLU(0.1)))
        for _ in range(layers - 1):
            self.net.append(nn.Sequential(nn.Linear(dim, dim), nn.LeakyReLU(0.1)))
        self.net.append(nn.Linear(dim, heads))
        self.cache_rel_pos = cache_rel_pos
        self.register_buffer("rel_pos", None, persistent=False)
        self._geom = {}

    def _tables(self, dims, device):
        key = (tuple(dims), str(device))
        if key not in self._geom:
            if len(dims) != 2:
                raise NotImplementedError("only the 2-D (h, w) bias of CT-ViT is implemented")
            h, w = dims
            ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
            grid = torch.stack((ys, xs)).reshape(2, -1).t()                       # attention.py:262-264
            rel = grid[:, None, :] - grid[None, :, :]                             # :265  (i - j)
            idx = (rel[..., 0] + (h - 1)) * (2 * w - 1) + (rel[..., 1] + (w - 1))  # row id in the unique table
            uy, ux = torch.meshgrid(torch.arange(-(h - 1), h), torch.arange(-(w - 1), w), indexing="ij")
            uniq = torch.stack((uy, ux), -1).reshape(-1, 2).to(F32)
            if self.log_dist:
                uniq = torch.sign(uniq) * torch.log(uniq.abs() + 1)               # :267-268
            if uniq.shape[0] > 65535:
                raise ValueError("relative-position table too large for uint16 indices")
            self._geom[key] = (uniq.to(device), idx.to(torch.uint16).contiguous().to(device))
        return self._geom[key]

    def lookup(self, *dimensions, device) -> PositionBias:
        rows, relidx = self._tables(dimensions, device)
        x = rows
        for layer in self.net[:-1]:
            lin = layer[0]
            x = ops.LinearF32Fn.apply(x, lin.weight, lin.bias, True)
        table = ops.LinearF32Fn.apply(x, self.net[-1].weight, self.net[-1].bias, False)   # [R, heads]
        n = relidx.shape[0]
        dense = torch.empty(self.heads, n, n, dtype=F32, device=device)
        hip.bias_expand(table.detach(), relidx, dense, self.heads, n)
        return PositionBias(table, relidx, dense, grid=tuple(int(d) for d in dimensions))

    def forward(self, *dimensions, device=torch.device("cpu")):
        """Reference signature: returns the dense [heads, n, n] bias.  (The reference ignores `device` and
        hard-codes 'cuda', attention.py:261; here the parameters' device decides.)"""
        dev = self.net[-1].weight.device
        _need_cuda(self.net[-1].weight, "ContinuousPositionBias")
        pb = self.lookup(*dimensions, device=dev)
        return _ExpandBias.apply(pb.table, pb.relidx, self.heads)


class _ExpandBias(torch.autograd.Function):
    """table [R, heads] -> dense [heads, n, n] with gradient (scatter-add) for callers of the reference API."""

    @staticmethod
    def forward(ctx, table, relidx, heads):
        n = relidx.shape[0]
        dense = torch.empty(heads, n, n, dtype=F32, device=table.device)
        hip.bias_expand(table, relidx, dense, heads, n)
        ctx.save_for_backward(relidx)
        ctx.rows = table.shape[0]
        return dense

    @staticmethod
    def backward(ctx, d):
        (relidx,) = ctx.saved_tensors
        heads = d.shape[0]
        flat = relidx.reshape(-1).to(torch.int64)
        out = torch.zeros(ctx.rows, heads, dtype=F32, device=d.device)
        out.index_add_(0, flat, d.reshape(heads, -1).t().contiguous())
        return out, None, None


class Transformer(nn.Module):
    """depth x { PEG + res, Attention + res, FeedForward + res }, final bias-less LayerNorm
    (reference attention.py:281-336).  layers[i] = ModuleList([PEG | None, Attention, None, FeedForward])."""

    def __init__(self, dim, *, depth, dim_context=None, causal=False, dim_head=64, heads=8, ff_mult=4, peg=False,
                 peg_causal=False, attn_num_null_kv=2, has_cross_attn=False, attn_dropout=0.0, ff_dropout=0.0):
        super().__init__()
        if has_cross_attn:
            raise NotImplementedError("cross-attention layers belong to GenerateCT's MaskGit, not the CT-CLIP path")
        self.layers = nn.ModuleList([])
        for _ in range(depth):
            self.layers.append(nn.ModuleList([
                PEG(dim=dim, causal=peg_causal) if peg else None,
                Attention(dim=dim, dim_head=dim_head, heads=heads, causal=causal, dropout=attn_dropout),
                None,
                FeedForward(dim=dim, mult=ff_mult, dropout=ff_dropout),
            ]))
        self.norm_out = LayerNorm(dim)

    def forward(self, x, video_shape: Tuple[int, int, int, int] = None, attn_bias=None, context=None,
                self_attn_mask=None, cross_attn_context_mask=None, out_swap=None):
        """Reference signature (attention.py:313-336) plus `out_swap=(A, C)`: the rows of the result, [B][A][C] as this
        transformer sees them, are returned as [B, C, A, d] -- the re-ordering CTViT.encode applies next (ctvit.py:96,
        101), written by the final LayerNorm itself."""
        _need_cuda(x, "Transformer")
        if exists(context) or exists(self_attn_mask) or exists(cross_attn_context_mask):
            raise NotImplementedError("context / masks are GenerateCT-only branches")
        x = x.to(F32)
        for peg, self_attn, _cross, ff in self.layers:
            # Residual adds are fused into the kernels' epilogues unless someone hooked the sub-module
            # (attribution code hooks layers[i][1], visualizations.py:242-263): then the module is called the
            # reference way so hooks see the branch output.
            x16 = None
            if exists(peg):
                if _hooked(peg):
                    x = peg(x, shape=video_shape) + x
                else:
                    x = peg.forward(x, shape=video_shape, residual=True)
                    x16 = peg._last_bf16
                    peg._last_bf16 = None
            if _hooked(self_attn):
                out, _w = self_attn(x, attn_bias=attn_bias)
                x = out + x
            else:
                x, _w = self_attn.forward(x, attn_bias=attn_bias, residual=True, x16=x16)
            x = ff(x) + x if _hooked(ff) else ff.forward(x, residual=True)
        if out_swap is None:
            return self.norm_out(x)
        if _hooked(self.norm_out):                   # a hook wants the reference's output: normalise, then re-order
            y = self.norm_out(x)
            return ops.SwapMiddleFn.apply(y.reshape(-1, int(out_swap[0]), int(out_swap[1]), y.shape[-1]))
        return self.norm_out.forward_swapped(x, out_swap[0], out_swap[1])
