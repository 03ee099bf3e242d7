"""HIP execution of a `transformers.BertModel` text encoder (reference call site src/models/ctclip.py:107).

The reference hands CTCLIP a HuggingFace BertModel (src/train_ctclip.py:17).  CTCLIP keeps that module
object -- its parameters, names and state-dict keys are untouched -- and, when it recognises the BERT layout,
runs the encoder through the same gfx950 kernels as CT-ViT: fused QKV GEMM, LDS-staged softmax attention
with the padding mask (d_head 64), post-LN residual blocks, erf-GELU MLP.

Absolute position embeddings, post-LayerNorm, erf GELU (BertConfig defaults, and CXR-BERT's).  In train mode the
config's dropouts apply where transformers applies them: after the embedding LayerNorm, on the attention probabilities
(keep flags drawn here, applied inside the attention kernels: ctclip_attn_fwd_dropout / _bwd_dropout) and on the two dense
outputs before their residual adds (ctclip_dropout_add / _bwd: flags re-evaluated from a counter-based generator, nothing
stored).  The random stream is the library's, seeded per forward from torch's generator, not a replay of the reference's:
parity with dropout on is statistical, the p = 0 path is what the golden vectors pin.
"""
from __future__ import annotations

import math

import torch
from torch.autograd import Function

from . import ops
from .lib import hip

F32 = torch.float32
BF16 = torch.bfloat16


def is_hf_bert(m) -> bool:
    return all(hasattr(m, a) for a in ("embeddings", "encoder", "config")) and hasattr(m.encoder, "layer") \
        and hasattr(m.embeddings, "word_embeddings") and hasattr(m.embeddings, "token_type_embeddings")


class BertEmbedFn(Function):
    @staticmethod
    def forward(ctx, ids, tt, word, pos, type_, lnw, lnb, eps):
        B, L = ids.shape
        Hd = word.shape[1]
        rows = B * L
        ids_c, tt_c = ids.contiguous(), (tt.contiguous() if tt is not None else None)
        e = torch.empty(rows, Hd, dtype=F32, device=word.device)
        hip.bert_embed_fwd(ids_c, tt_c, word.detach(), pos.detach(), type_.detach(), e, rows, L, Hd)
        _, y, mean, rstd = ops.layernorm(e, lnw.detach(), lnb.detach(), eps, want16=False, want32=True)
        ctx.save_for_backward(ids_c, tt_c if tt_c is not None else ids_c.new_empty(0), e, mean, rstd, lnw)
        ctx.dims = (B, L, Hd, word.shape[0], pos.shape[0], type_.shape[0])
        ctx.params = (word, pos, type_, lnw, lnb)
        return y

    @staticmethod
    @ops.announces
    def backward(ctx, dy):
        ids, tt, e, mean, rstd, lnw = ctx.saved_tensors
        B, L, Hd, nw, npos, nt = ctx.dims
        rows = B * L
        dev = dy.device
        p_word, p_pos, p_type, p_lw, p_lb = ctx.params
        de = torch.empty(rows, Hd, dtype=F32, device=dev)
        dlw, k1 = ops.grad_slot(p_lw)
        dlb, k2 = ops.grad_slot(p_lb)
        hip.layernorm_bwd(ops._c(dy), e, lnw, mean, rstd, None, de, None, dlw, dlb, rows, Hd)
        dword, k3 = ops.grad_slot(p_word)                 # 30522 x 768: scatter-add straight into the arena
        dpos, k4 = ops.grad_slot(p_pos)
        dtype_, k5 = ops.grad_slot(p_type)
        hip.bert_embed_bwd(ids, tt if tt.numel() else None, de, dword, dpos, dtype_, rows, L, Hd, p_type.shape[0], nw)
        r = ops._ret
        return None, None, r(dword, k3), r(dpos, k4), r(dtype_, k5), r(dlw, k1), r(dlb, k2), None


_SITE_SHIFT = 40       # element counters of one dropout site stay below 2^40; (layer, site) selects the offset above it


def _site_offset(layer_idx, site):
    """offset of dropout site `site` (0 attention probabilities, 1 attention-output dense, 2 FFN-output dense) of encoder
    layer `layer_idx` in the counter space of ctclip_dropout_*"""
    return (layer_idx * 4 + site) << _SITE_SHIFT


def _dropped(g32, g16, drop, site):
    """Gradient of the dense branch behind a hidden-state dropout: (f32, bf16) of g * keep / (1 - p), the keep flags
    re-evaluated from (seed, offset); the inputs pass through when there was no dropout."""
    p_hid, seed, layer_idx = drop[1], drop[3], drop[4]
    if p_hid <= 0:
        return g32, g16
    d = torch.empty_like(g32)
    d16 = torch.empty_like(g16)
    hip.dropout_bwd(g32, d, d16, g32.numel(), p_hid, seed, _site_offset(layer_idx, site))
    return d, d16


def _attn_bwd(ctx_drop, *args):
    """hip.attn_bwd, or its dropout form with the forward's keep flags spliced in after `mask`."""
    keep_a, p_att = ctx_drop[0], ctx_drop[2]
    if keep_a is None:
        return hip.attn_bwd(*args)
    return hip.attn_bwd_dropout(*args[:12], keep_a, 1.0 / (1.0 - p_att), *args[12:])


class BertLayerFn(Function):
    """One post-LN encoder layer on x [B*L, H] f32.  p = (qw,qb,kw,kb,vw,vb,aow,aob,l1w,l1b,iw,ib,ow,ob,l2w,l2b)."""

    @staticmethod
    def forward(ctx, x, mask_add, sh, cfg, *p):
        B, L, heads, dh, dp, eps = cfg[:6]
        p_hid, p_att = (cfg[6], cfg[7]) if len(cfg) > 6 else (0.0, 0.0)      # training-mode dropout (0 = none)
        seed, layer_idx = (cfg[8], cfg[9]) if len(cfg) > 8 else (0, 0)       # the step's seed, this layer's counter range
        M, Hd = x.shape
        inner = heads * dp
        I = sh["inter"]
        dev = x.device
        xb = ops.cast16(x)
        qkv = ops.gemm(xb, sh["wqkv"], M, 3 * inner, Hd, bias=sh["bqkv"])
        o = torch.empty(M, inner, dtype=BF16, device=dev)
        lse = torch.empty(B, heads, L, dtype=F32, device=dev)
        scale = 1.0 / math.sqrt(dh)
        # BertSelfAttention drops attention probabilities, BertSelfOutput / BertOutput drop the dense output before the
        # residual add (transformers modeling_bert.py).  Keep flags are a function of (seed, layer, site, element)
        # evaluated inside the kernels (ctclip_dropout_*); only the attention flags are materialised, for the attention
        # kernels.  The p = 0 path below is untouched by any of it.
        keep_a = None
        if p_att > 0:
            keep_a = torch.empty(B, heads, L, L, dtype=torch.uint8, device=dev)
            hip.dropout_keep(keep_a, keep_a.numel(), p_att, seed, _site_offset(layer_idx, 0))
            hip.attn_fwd_dropout(qkv, qkv[:, inner:], qkv[:, 2 * inner:], o, lse, None, mask_add, keep_a, 1.0 / (1.0 - p_att),
                                 B, L, heads, dp, 3 * inner, 3 * inner, 3 * inner, inner, scale)
        else:
            hip.attn_fwd(qkv, qkv[:, inner:], qkv[:, 2 * inner:], o, lse, None, mask_add, B, L, heads, dp,
                         3 * inner, 3 * inner, 3 * inner, inner, scale)
        if p_hid > 0:
            ad = ops.gemm(o, sh["wao"], M, Hd, inner, out_dtype=F32, bias=p[7].detach())
            a = torch.empty_like(ad)
            hip.dropout_add(ops._c(x), ad, a, a.numel(), p_hid, seed, _site_offset(layer_idx, 1))
        else:
            a = ops.gemm(o, sh["wao"], M, Hd, inner, out_dtype=F32, bias=p[7].detach(), resid=x)
        x1_16, x1, mean1, rstd1 = ops.layernorm(a, p[8].detach(), p[9].detach(), eps, want16=True, want32=True)
        hpre = ops.gemm(x1_16, sh["wi"], M, I, Hd, bias=p[11].detach())
        m = torch.empty_like(hpre)
        hip.gelu_fwd(hpre, m, hpre.numel())
        if p_hid > 0:
            od = ops.gemm(m, sh["wo"], M, Hd, I, out_dtype=F32, bias=p[13].detach())
            o2 = torch.empty_like(od)
            hip.dropout_add(x1, od, o2, o2.numel(), p_hid, seed, _site_offset(layer_idx, 2))
        else:
            o2 = ops.gemm(m, sh["wo"], M, Hd, I, out_dtype=F32, bias=p[13].detach(), resid=x1)
        _, x2, mean2, rstd2 = ops.layernorm(o2, p[14].detach(), p[15].detach(), eps, want16=False, want32=True)
        ctx.save_for_backward(xb, qkv, o, lse, a, mean1, rstd1, x1_16, hpre, m, o2, mean2, rstd2, p[8], p[14],
                              mask_add if mask_add is not None else x.new_empty(0))
        ctx.sh, ctx.cfg, ctx.params = sh, cfg, p
        ctx.drop = (keep_a, p_hid, p_att, seed, layer_idx)
        return x2

    @staticmethod
    @ops.announces
    def backward(ctx, dy):
        xb, qkv, o, lse, a, mean1, rstd1, x1_16, hpre, m, o2, mean2, rstd2, l1w, l2w, mask_add = ctx.saved_tensors
        sh = ctx.sh
        B, L, heads, dh, dp, eps = ctx.cfg[:6]
        keep_a, p_hid, p_att = ctx.drop[:3]
        M, Hd = a.shape
        inner, I = heads * dp, sh["inter"]
        dev = dy.device
        if mask_add.numel() == 0:
            mask_add = None
        p = ctx.params
        if dp != dh:
            return BertLayerFn._backward_padded(ctx, dy)
        G = [ops.grad_slot(t) for t in p]                  # param-shaped accumulate buffers (.grad when pre-bound)
        g = [b for b, _ in G]
        # output LayerNorm
        do2 = torch.empty(M, Hd, dtype=F32, device=dev)
        do2b = torch.empty(M, Hd, dtype=BF16, device=dev)
        hip.layernorm_bwd(ops._c(dy), o2, l2w, mean2, rstd2, None, do2, do2b, g[14], g[15], M, Hd)
        dod, do2b = _dropped(do2, do2b, ctx.drop, 2)          # into the dense branch; the residual keeps do2 itself
        ops.colsum(dod, out=g[13])
        dm = ops.dgrad(do2b, sh["wo"], M, Hd, I, out_dtype=BF16, wT16=sh.get("woT"))
        ops.wgrad(do2b, m, Hd, I, M, out=g[12])
        dh_ = torch.empty_like(hpre)
        hip.gelu_bwd(dm, hpre, dh_, hpre.numel())
        ops.colsum(dh_, out=g[11])
        dx1 = ops.dgrad(dh_, sh["wi"], M, I, Hd, resid=do2, wT16=sh.get("wiT"))
        ops.wgrad(dh_, x1_16, I, Hd, M, out=g[10])
        # attention-output LayerNorm
        da = torch.empty(M, Hd, dtype=F32, device=dev)
        dab = torch.empty(M, Hd, dtype=BF16, device=dev)
        hip.layernorm_bwd(dx1, a, l1w, mean1, rstd1, None, da, dab, g[8], g[9], M, Hd)
        dad, dab = _dropped(da, dab, ctx.drop, 1)
        ops.colsum(dad, out=g[7])
        do = ops.dgrad(dab, sh["wao"], M, Hd, inner, out_dtype=BF16, wT16=sh.get("waoT"))
        ops.wgrad(dab, o, Hd, inner, M, out=g[6])
        dqkv = torch.empty(M, 3 * inner, dtype=BF16, device=dev)
        delta = torch.empty(B, heads, L, dtype=F32, device=dev)
        _attn_bwd(ctx.drop, qkv, qkv[:, inner:], qkv[:, 2 * inner:], o, do, lse, delta, dqkv, dqkv[:, inner:],
                  dqkv[:, 2 * inner:], None, mask_add, None, None, None, 0, 0, 0, B, L, heads, dp,
                  3 * inner, 3 * inner, 3 * inner, inner, inner, 3 * inner, 3 * inner, 3 * inner,
                  1.0 / math.sqrt(dh))
        for j in range(3):                                  # query / key / value: weight and bias
            part = dqkv[:, j * inner:(j + 1) * inner]
            ops.colsum(part, out=g[2 * j + 1])
            ops.wgrad(part, xb, inner, Hd, M, out=g[2 * j])
        dx = ops.dgrad(dqkv, sh["wqkv"], M, 3 * inner, Hd, resid=da, wT16=sh.get("wqkvT"))
        return (dx, None, None, None) + tuple(ops._ret(b, d) for b, d in G)

    @staticmethod
    def _backward_padded(ctx, dy):
        """Head dims below 32 (toy configs): kernels run on zero-padded heads, gradients are un-padded afterwards."""
        xb, qkv, o, lse, a, mean1, rstd1, x1_16, hpre, m, o2, mean2, rstd2, l1w, l2w, mask_add = ctx.saved_tensors
        sh = ctx.sh
        B, L, heads, dh, dp, eps = ctx.cfg[:6]
        keep_a, p_hid, p_att = ctx.drop[:3]
        M, Hd = a.shape
        inner, I = heads * dp, sh["inter"]
        dev = dy.device
        if mask_add.numel() == 0:
            mask_add = None
        z = lambda n: torch.zeros(n, dtype=F32, device=dev)
        # output LayerNorm
        do2 = torch.empty(M, Hd, dtype=F32, device=dev)
        do2b = torch.empty(M, Hd, dtype=BF16, device=dev)
        dl2w, dl2b = z(Hd), z(Hd)
        hip.layernorm_bwd(ops._c(dy), o2, l2w, mean2, rstd2, None, do2, do2b, dl2w, dl2b, M, Hd)
        dod, do2b = _dropped(do2, do2b, ctx.drop, 2)
        dbo = ops.colsum(dod)
        dm = ops.dgrad(do2b, sh["wo"], M, Hd, I, out_dtype=BF16, wT16=sh.get("woT"))
        dwo = ops.wgrad(do2b, m, Hd, I, M)
        dh_ = torch.empty_like(hpre)
        hip.gelu_bwd(dm, hpre, dh_, hpre.numel())
        dbi = ops.colsum(dh_)
        dx1 = ops.dgrad(dh_, sh["wi"], M, I, Hd, resid=do2, wT16=sh.get("wiT"))
        dwi = ops.wgrad(dh_, x1_16, I, Hd, M)
        # attention-output LayerNorm
        da = torch.empty(M, Hd, dtype=F32, device=dev)
        dab = torch.empty(M, Hd, dtype=BF16, device=dev)
        dl1w, dl1b = z(Hd), z(Hd)
        hip.layernorm_bwd(dx1, a, l1w, mean1, rstd1, None, da, dab, dl1w, dl1b, M, Hd)
        dad, dab = _dropped(da, dab, ctx.drop, 1)
        dbao = ops.colsum(dad)
        do = ops.dgrad(dab, sh["wao"], M, Hd, inner, out_dtype=BF16, wT16=sh.get("waoT"))
        dwao = ops.wgrad(dab, o, Hd, inner, M)
        dqkv = torch.empty(M, 3 * inner, dtype=BF16, device=dev)
        delta = torch.empty(B, heads, L, dtype=F32, device=dev)
        _attn_bwd(ctx.drop, qkv, qkv[:, inner:], qkv[:, 2 * inner:], o, do, lse, delta, dqkv, dqkv[:, inner:],
                  dqkv[:, 2 * inner:], None, mask_add, None, None, None, 0, 0, 0, B, L, heads, dp,
                  3 * inner, 3 * inner, 3 * inner, inner, inner, 3 * inner, 3 * inner, 3 * inner,
                  1.0 / math.sqrt(dh))
        dbqkv = ops.colsum(dqkv)
        dx = ops.dgrad(dqkv, sh["wqkv"], M, 3 * inner, Hd, resid=da, wT16=sh.get("wqkvT"))
        dwqkv = ops.wgrad(dqkv, xb, 3 * inner, Hd, M)
        un = lambda w: ops.unpad_head_rows(w, heads, dh, dp)
        unb = lambda b_: ops.unpad_head_rows(b_[:, None], heads, dh, dp)[:, 0]
        grads = (un(dwqkv[:inner]), unb(dbqkv[:inner]), un(dwqkv[inner:2 * inner]), unb(dbqkv[inner:2 * inner]),
                 un(dwqkv[2 * inner:]), unb(dbqkv[2 * inner:]), un(dwao.t()).t(), dbao, dl1w, dl1b, dwi, dbi, dwo, dbo,
                 dl2w, dl2b)
        return (dx, None, None, None) + grads


def _layer_params(layer):
    at, out = layer.attention, layer.output
    s = at.self
    return (s.query.weight, s.query.bias, s.key.weight, s.key.bias, s.value.weight, s.value.bias,
            at.output.dense.weight, at.output.dense.bias, at.output.LayerNorm.weight, at.output.LayerNorm.bias,
            layer.intermediate.dense.weight, layer.intermediate.dense.bias, out.dense.weight, out.dense.bias,
            out.LayerNorm.weight, out.LayerNorm.bias)


def _layer_shadows(layer, heads, dh, dp):
    cache = layer.__dict__.setdefault("_ctclip_shadow", ops.ShadowCache())
    p = _layer_params(layer)

    def build():                                      # head sizes that need zero rows per head: framework ops
        pad = lambda w: ops.pad_head_rows(w, heads, dh, dp)
        padb = lambda b: ops.pad_head_rows(b[:, None], heads, dh, dp)[:, 0]
        d = {
            "wqkv": torch.cat((pad(p[0]), pad(p[2]), pad(p[4])), 0).to(BF16).contiguous(),
            "bqkv": torch.cat((padb(p[1]), padb(p[3]), padb(p[5]))).to(F32).contiguous(),
            "wao": pad(p[6].t()).t().to(BF16).contiguous(),
            "wi": p[10].to(BF16).contiguous(), "wo": p[12].to(BF16).contiguous(), "inter": p[10].shape[0],
        }
        d.update(wqkvT=d["wqkv"].t().contiguous(), waoT=d["wao"].t().contiguous(), wiT=d["wi"].t().contiguous(),
                 woT=d["wo"].t().contiguous())
        return d

    def make():                                       # dh == dp (BERT-base: 64): one descriptor table (ctclip_shadow_multi)
        inner, Hd, I = heads * dh, p[0].shape[1], p[10].shape[0]
        S = ops.ShadowSet(p[0].device)
        o = {"wqkv": S.zeros(3 * inner, Hd), "wqkvT": S.zeros(Hd, 3 * inner), "bqkv": S.zeros(3 * inner, dtype=F32),
             "wao": S.zeros(Hd, inner), "waoT": S.zeros(inner, Hd), "wi": S.zeros(I, Hd), "wiT": S.zeros(Hd, I),
             "wo": S.zeros(Hd, I), "woT": S.zeros(I, Hd), "inter": I}
        for j, (w, b) in enumerate(((p[0], p[1]), (p[2], p[3]), (p[4], p[5]))):
            S.add(w, o["wqkv"][j * inner:]); S.add(w, o["wqkvT"][:, j * inner:], transpose=True)
            S.add(b, o["bqkv"][j * inner:])
        S.add(p[6], o["wao"]); S.add(p[6], o["waoT"], transpose=True)
        S.add(p[10], o["wi"]); S.add(p[10], o["wiT"], transpose=True)
        S.add(p[12], o["wo"]); S.add(p[12], o["woT"], transpose=True)
        S.out = o
        return S

    if dh == dp:
        return cache.get_set("bert", p, make)
    return cache.get("bert", p, build)


def bert_last_hidden_state(model, input_ids, token_type_ids=None, attention_mask=None, **unused):
    """HIP forward of transformers.BertModel -> last_hidden_state [B, L, H] (f32, autograd-enabled)."""
    cfg = model.config
    if not input_ids.is_cuda:
        raise RuntimeError("text encoder: MI355X HIP path only (no CPU fallback); move the token ids to cuda")
    if getattr(cfg, "position_embedding_type", "absolute") not in (None, "absolute"):
        raise NotImplementedError("only absolute position embeddings are implemented")
    if cfg.hidden_act not in ("gelu",):
        raise NotImplementedError(f"hidden_act={cfg.hidden_act!r}: only erf-GELU is implemented")
    p_hid = float(cfg.hidden_dropout_prob) if model.training else 0.0
    p_att = float(cfg.attention_probs_dropout_prob) if model.training else 0.0
    if not (0.0 <= p_hid < 1.0 and 0.0 <= p_att < 1.0):
        raise ValueError("dropout probabilities must be in [0, 1)")
    B, L = input_ids.shape
    Hd, heads = cfg.hidden_size, cfg.num_attention_heads
    if Hd % 8 or cfg.intermediate_size % 8:
        raise ValueError("hidden/intermediate sizes must be multiples of 8 for the bf16 MFMA path")
    dh = Hd // heads
    dp = ops.head_pad(dh)
    emb = model.embeddings
    x = BertEmbedFn.apply(input_ids, token_type_ids, emb.word_embeddings.weight, emb.position_embeddings.weight,
                          emb.token_type_embeddings.weight, emb.LayerNorm.weight, emb.LayerNorm.bias,
                          float(cfg.layer_norm_eps))
    if p_hid > 0:
        x = torch.nn.functional.dropout(x, p_hid, True)          # BertEmbeddings.dropout, after its LayerNorm
    mask_add = None
    if attention_mask is not None:
        mask_add = ((1.0 - attention_mask.to(F32)) * torch.finfo(F32).min).contiguous()
    # one seed per forward from torch's (CPU) default generator -- torch.manual_seed() makes a run reproducible, no device
    # sync -- and a counter range per (layer, site): see ctclip_dropout_* in include/ctclip_hip.h
    seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if (p_hid > 0 or p_att > 0) else 0
    for li, layer in enumerate(model.encoder.layer):
        lcfg = (B, L, heads, dh, dp, float(cfg.layer_norm_eps), p_hid, p_att, seed, li)
        x = BertLayerFn.apply(x, mask_add, _layer_shadows(layer, heads, dh, dp), lcfg, *_layer_params(layer))
    return x.reshape(B, L, Hd)
