"""Autograd functions over libctclip_hip.so.

Every forward/backward here is a sequence of C-ABI kernel launches on torch's current HIP stream; torch
supplies device memory, the autograd graph and (for one-off weight re-layouts) tensor plumbing.  There
is no eager/CPU fallback: CPU tensors raise inside `hip.*`.

Numerics contract: MFMA operands are bf16, accumulation / residual stream / LayerNorm /
softmax statistics / losses / optimiser are f32.
"""
from __future__ import annotations

import math
import os
import weakref

import torch
from torch.autograd import Function

from .lib import hip

BF16 = torch.bfloat16
F32 = torch.float32

_weight_epoch = 0


def bump_weight_epoch() -> None:
    """Call after parameters were updated through raw pointers (the fused optimiser does)."""
    global _weight_epoch
    _weight_epoch += 1


class ShadowCache:
    """Per-module cache of kernel-layout copies of parameters (bf16, padded, re-ordered).

    Rebuilt lazily when any source parameter's version counter, storage or the global weight epoch changes."""

    def __init__(self):
        self._store = {}

    def get(self, key, params, builder):
        ver = (_weight_epoch,) + tuple((p._version, p.data_ptr()) for p in params)
        ent = self._store.get(key)
        if ent is None or ent[0] != ver:
            with torch.no_grad():
                ent = (ver, builder())
            self._store[key] = ent
        return ent[1]

    def get_set(self, key, params, make):
        """The same for descriptor-driven shadows: `make()` -> ShadowSet is called once (and again if a parameter's storage
        moved); afterwards a changed version only re-runs the copy kernel (PLAN.refresh) -- one launch for the whole model per
        optimiser step -- and no tensor is re-created."""
        ver = (_weight_epoch,) + tuple((p._version, p.data_ptr()) for p in params)
        ent = self._store.get(key)
        ptrs = tuple(p.data_ptr() for p in params)
        if ent is None or ent[2] != ptrs:
            with torch.no_grad():
                s_ = make()
            if ent is not None:
                PLAN.unregister(ent[1])
            PLAN.register(s_)
            ent = [None, s_, ptrs]
            self._store[key] = ent
        if ent[0] != ver:
            s_ = ent[1]
            # already refilled by another module's refresh at this epoch, and none of OUR parameters was written since we last looked
            covered = s_.fresh_epoch == ver[0] and ent[0] is not None and ent[0][1:] == ver[1:]
            if not covered:
                with torch.no_grad():
                    PLAN.refresh(s_, ver[0])
            ent[0] = ver
        return ent[1].out


class ShadowSet:
    """The descriptor-driven shadows of one module (ctclip_shadow_multi, csrc/elementwise.hip): destination tensors that live as
    long as the module and the table rows that (re)fill them from the f32 parameters.  `out` is what the module's kernels read."""

    def __init__(self, device):
        self.device = device
        self.rows = []                       # [src, dst, rows, cols, src_ld, dst_ld, flags, scale]
        self.keep = []                       # the tensors behind the raw pointers
        self.out = {}
        self.fresh_epoch = None              # weight epoch the destinations were last filled at
        self._table = None

    def zeros(self, *shape, dtype=torch.bfloat16):
        t = torch.zeros(*shape, dtype=dtype, device=self.device)
        self.keep.append(t)
        return t

    def add(self, src, dst, *, transpose=False, blk=0, scale=None, rowsum=False):
        """dst <- src ([rows, cols] view of an f32 parameter or a bf16 tensor; 1-D = one row) in the layout the flags select.
        `dst` is a view whose pointer already includes any row / column offset; its row stride is the leading dimension."""
        src = src.detach()
        s2 = src if src.dim() == 2 else src.reshape(1, -1)
        assert s2.stride(1) == 1 and (dst.dim() == 1 or dst.stride(-1) == 1)
        rows, cols = s2.shape
        flags = (1 if transpose else 0) | (2 if s2.dtype == torch.bfloat16 else 0) | (4 if dst.dtype == torch.float32 else 0) \
            | (8 if rowsum else 0) | (int(blk) << 8)
        dst_ld = dst.stride(0) if dst.dim() == 2 else (cols if not transpose else 1)
        sc = scale.detach() if scale is not None else None
        self.rows.append([s2.data_ptr(), dst.data_ptr(), rows, cols, s2.stride(0), dst_ld, flags, 0 if sc is None else sc.data_ptr()])
        self.keep += [src, dst] + ([sc] if sc is not None else [])
        self._table = None

    def tiles(self):
        return [((r[2] + 31) // 32) * (1 if r[6] & 8 else (r[3] + 63) // 64) for r in self.rows]


class ShadowPlan:
    """Every LIVE ShadowSet of the process.  refresh(set) brings a set's destinations up to date: after an optimiser step
    (the weight epoch moved) the first caller refills ALL sets with one launch; a set whose own parameters were written in
    place in between is refilled alone.

    The sets are held WEAKLY: a set belongs to its module's ShadowCache and pins that module's f32 parameters and every bf16 /
    transposed shadow tensor (`keep`), so a strong reference here would keep every model ever built in the process resident on
    the GPU and refill dead models' shadows after each optimiser step.  When a module (or the whole model) is dropped its sets go
    with it, leave the plan, and the cached all-sets table is rebuilt."""

    def __init__(self):
        self._refs = []                      # weakref.ref(ShadowSet), in registration order
        self.epoch_done = None
        self._all = None

    @property
    def sets(self):
        return [s_ for s_ in (r() for r in self._refs) if s_ is not None]

    def _dropped(self, ref):
        self._refs = [r for r in self._refs if r is not ref]
        self._all = None

    def unregister(self, s_):
        self._refs = [r for r in self._refs if r() is not None and r() is not s_]
        self._all = None

    @staticmethod
    def _device_table(sets):
        rows = [r for s_ in sets for r in s_.rows]
        tiles = [t for s_ in sets for t in s_.tiles()]
        starts, acc = [], 0
        for t in tiles:
            starts.append(acc)
            acc += t
        dev = sets[0].device
        return (torch.tensor(rows, dtype=torch.int64, device=dev), torch.tensor(starts, dtype=torch.int32, device=dev), len(rows), acc)

    @staticmethod
    def _launch(tab):
        table, starts, n, total = tab
        hip.shadow_multi(table, starts, n, total)

    def register(self, s_):
        self._refs.append(weakref.ref(s_, self._dropped))
        self._all = None

    def prime(self, epoch=None):
        """Bring EVERY registered set up to date on the current stream if the weight epoch moved (what the first refresh() of a
        step does).  Called before work is forked onto a second stream, so that no module refills another stream's shadows."""
        epoch = _weight_epoch if epoch is None else epoch
        live = self.sets
        if self.epoch_done != epoch and len(live) > 1:
            self.refresh(live[0], epoch)
        elif self.epoch_done != epoch:
            self.epoch_done = epoch

    def refresh(self, s_, epoch):
        """Fill s_'s destinations now.  The first call after the weight epoch moved fills EVERY registered set with one launch
        (and marks them: `fresh_epoch`); later calls in the same epoch fill the one set."""
        live = self.sets
        if self.epoch_done != epoch and len(live) > 1:
            if self._all is None:
                by_dev = {}
                for x in live:
                    by_dev.setdefault(x.device, []).append(x)
                self._all = {d: self._device_table(v) for d, v in by_dev.items()}
            for d, tab in self._all.items():
                with torch.cuda.device(d):
                    self._launch(tab)
            self.epoch_done = epoch
            for x in live:
                x.fresh_epoch = epoch
            return
        if s_._table is None:
            s_._table = self._device_table([s_])
        self._launch(s_._table)
        s_.fresh_epoch = epoch


PLAN = ShadowPlan()

# ---------------------------------------------------------------------------------------------------
# The text tower's stream.  BERT-base on batch x 128 tokens is ~150 kernels per pass that each fill a fraction of the chip (its
# GEMMs have 48 row tiles); run next to the image tower instead of in front of it they disappear into the image tower's tails:
# 413 -> 405 ms per 96-pair step.  models/ctclip.py forks it in forward; autograd runs the backward of those nodes on the same
# stream by its own rule; everything that reads what the other stream wrote joins first (join_side_streams: the trainer after
# backward, HipAdam.zero_grad / step, GradSync before a collective).  CTCLIP_TEXT_STREAM=0 keeps one stream.
# ---------------------------------------------------------------------------------------------------
_text_stream = {"on": os.environ.get("CTCLIP_TEXT_STREAM", "1") != "0", "side": {}, "main": {}}


def fork_text_stream(device):
    """-> (main, side) streams for a forward that is about to run the text tower on `side`, or None when switched off.  Every
    weight shadow is refilled on the main stream first, so that no module refills another stream's operands."""
    if not _text_stream["on"] or device.type != "cuda":
        return None
    idx = device.index if device.index is not None else torch.cuda.current_device()
    side = _text_stream["side"].get(idx)
    if side is None:
        side = _text_stream["side"][idx] = torch.cuda.Stream(device=device)
    main = torch.cuda.current_stream(device)
    if main == side:
        return None
    _text_stream["main"][idx] = main
    PLAN.prime()
    side.wait_stream(main)
    return main, side


def join_side_streams():
    """The current stream waits for everything issued so far on the text stream AND on the stream it was forked from (their
    backward kernels accumulate into one gradient arena in place)."""
    if not _text_stream["side"]:
        return
    cur = torch.cuda.current_stream()
    for table in (_text_stream["side"], _text_stream["main"]):
        st = table.get(cur.device.index)
        if st is not None and st != cur:
            cur.wait_stream(st)


class JoinAfterBackwardFn(Function):
    """Identity on the text tower's output, applied on the text stream.  Its backward is the FIRST node autograd runs on that
    stream; it queues an engine callback that makes the stream `.backward()` was called on wait for the text stream once the
    whole backward has been issued -- so every `.grad` the text tower accumulated in place is complete on the caller's stream
    when `backward()` returns, whatever the caller does next (the reference clips gradient norms between backward() and step(),
    src/utils/CTClipTrainer.py:199-202).  The explicit joins (trainer, HipAdam, GradSync) stay as belt and braces."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, dy):
        torch.autograd.Variable._execution_engine.queue_callback(join_side_streams)
        return dy


def pad8(n: int) -> int:
    return (n + 7) // 8 * 8


def head_pad(dh: int) -> int:
    if dh <= 32:
        return 32
    if dh <= 64:
        return 64
    raise ValueError(f"dim_head {dh} > 64 is not supported by the gfx950 attention kernel")


def pad_head_rows(w: torch.Tensor, heads: int, dh: int, dp: int) -> torch.Tensor:
    """[heads*dh, K] -> [heads*dp, K] with zero rows inserted per head."""
    if dh == dp:
        return w
    out = w.new_zeros(heads, dp, w.shape[1])
    out[:, :dh] = w.reshape(heads, dh, -1)
    return out.reshape(heads * dp, -1)


def unpad_head_rows(w: torch.Tensor, heads: int, dh: int, dp: int) -> torch.Tensor:
    if dh == dp:
        return w
    return w.reshape(heads, dp, -1)[:, :dh].reshape(heads * dh, -1)


# ---------------------------------------------------------------------------------------------------
# thin kernel wrappers
# ---------------------------------------------------------------------------------------------------

def gemm(A, B, M, N, K, *, a_kmajor=True, b_kmajor=True, out=None, out_dtype=BF16, bias=None, resid=None,
         split_k=1, accumulate=False, alpha=1.0, act=0):
    """C[M,N] = opA(A) opB(B) on MFMA; A,B are 2-D bf16 (row stride taken from the tensors)."""
    if out is None:
        out = (torch.zeros if accumulate else torch.empty)(M, N, dtype=out_dtype, device=A.device)
    hip.gemm_bf16(A, B, out, bias, resid, M, N, K, A.stride(0), B.stride(0), out.stride(0),
                  0 if resid is None else resid.stride(0), int(a_kmajor), int(b_kmajor),
                  int(out.dtype == F32), split_k, int(accumulate), float(alpha), act)
    return out


def _splits_for(m_out: int, n_out: int, k: int) -> int:
    """split-K factor of a weight-gradient GEMM (small m_out x n_out, contraction over all tokens): ONE workgroup of the
    256x256x32 kernel (gemm4.hip, one resident per CU) for each of the 256 CUs, and at least 8 k-tiles per split.  Every split
    writes its partial tile to the split-K workspace and a second kernel sums them in order, so the partial volume grows
    with the split count: measured at 64 pairs (tools/bench_gemm.py, WG = tiles x splits): out / q projections 256 us at 256
    workgroups, 298 at 512, 344 at 768; kv 459 / 487 / 550; FF1, FF2 and the tubelet projection do not care."""
    if k % 32 == 0:
        nk = k // 32
        tiles = ((m_out + 255) // 256) * ((n_out + 255) // 256)
        return max(1, min(nk // 8 if nk >= 8 else 1, max(1, 256 // tiles)))
    nk = (k + 63) // 64
    tiles = ((m_out + 127) // 128) * ((n_out + 127) // 128)
    return max(1, min(nk, (1024 + tiles - 1) // tiles))


# (Rounds 1-3 could issue the weight-gradient GEMMs on a second stream, CTCLIP_WGRAD_STREAM=1: measured equal in round 3 and 2 %
# slower in round 4 -- 406 vs 397 ms, full-chip kernels only stretch each other -- and removed.  The text tower, whose kernels do
# NOT fill the chip, is what runs on a second stream: fork_text_stream above.)
def _wgrad_now(dy16, x16, n_feat, k_feat, tokens, out):
    hip.gemm_bf16(dy16, x16, out, None, None, n_feat, k_feat, tokens, dy16.stride(0), x16.stride(0), out.stride(0), 0,
                  0, 0, 1, _splits_for(n_feat, k_feat, tokens), 1, 1.0, 0)
    return out


def wgrad(dy16, x16, n_feat, k_feat, tokens, out=None):
    """dW[n_feat,k_feat] (+)= dy^T x over `tokens` rows; f32 atomics (out must be zero or a running sum).

    `out` given: a `.grad` slot that is accumulated into."""
    if out is None:
        out = torch.zeros(n_feat, k_feat, dtype=F32, device=dy16.device)
    return _wgrad_now(dy16, x16, n_feat, k_feat, tokens, out)

def colsum(x2d, out=None):
    """out[c] (+)= sum_r x[r, c]  (bias gradients); x f32 or bf16.  `out` given: a `.grad` slot that is accumulated into."""
    rows, cols = x2d.shape
    if out is None:
        out = torch.zeros(cols, dtype=F32, device=x2d.device)
    hip.colsum_accum(x2d, int(x2d.dtype == BF16), rows, cols, x2d.stride(0), out)
    return out


def dgrad(dy16, w16, tokens, n_feat, k_feat, *, out_dtype=F32, resid=None, wT16=None):
    """dx[tokens,k_feat] = dy[tokens,n_feat] W[n_feat,k_feat].

    With wT16 = W^T stored [k_feat, n_feat] (a transposed bf16 shadow of the small weight) the product is k-major x
    k-major, i.e. exactly the forward layout, and runs on the pipelined LDS-DMA kernel."""
    if wT16 is not None:
        return gemm(dy16, wT16, tokens, k_feat, n_feat, a_kmajor=True, b_kmajor=True, out_dtype=out_dtype, resid=resid)
    return gemm(dy16, w16, tokens, k_feat, n_feat, a_kmajor=True, b_kmajor=False, out_dtype=out_dtype, resid=resid)


# FF1's weight rows (and h's columns) are stored as [value 32 | gate 32 | value 32 | ...] blocks: a wave's 64-column slab of
# the 256 x 256 GEMM tile is then one value block and its gate block, and the kernel's register layout (gemm3.hip:nfrag_row)
# gives every lane a value and its gate, so the GEGLU forward / backward run in the GEMM epilogue without any exchange
GEGLU_BLOCK = 32


def pad64(n: int) -> int:
    return (n + 63) // 64 * 64


def layernorm(x2d, gamma, beta, eps, want16=True, want32=False):
    rows, dim = x2d.shape
    y16 = torch.empty(rows, dim, dtype=BF16, device=x2d.device) if want16 else None
    y32 = torch.empty(rows, dim, dtype=F32, device=x2d.device) if want32 else None
    mean = torch.empty(rows, dtype=F32, device=x2d.device)
    rstd = torch.empty(rows, dtype=F32, device=x2d.device)
    hip.layernorm_fwd(x2d, gamma, beta, y16, y32, mean, rstd, rows, dim, float(eps))
    return y16, y32, mean, rstd


def cast16(x):
    y = torch.empty(x.shape, dtype=BF16, device=x.device)
    hip.cast_f32_bf16(x, y, x.numel())
    return y


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def grad_slot(p, note=True):
    """-> (buffer, direct).  Every gradient kernel ACCUMULATES, so when the parameter already owns a contiguous f32 .grad
    (the flat arena of ctclip_hip.optim.HipAdam pre-binds one) the kernels add straight into it and the Function returns
    None for that input: no zero-fill, no extra `grad += new` pass per tensor.

    A parameter whose optimiser averages gradients across ranks carries its listener (`_ctclip_sync`, a GradSync): it is told
    BEFORE the slot is handed out (a second backward in one step must not write into a slice RCCL is still reducing) and,
    with `note`, again once the running backward has issued all of its kernels (announce_grads)."""
    g = p.grad
    if g is not None and g.dtype == F32 and g.is_contiguous() and g.shape == p.shape and g.device == p.device:
        sync = getattr(p, "_ctclip_sync", None)
        if sync is not None:
            sync.before_write(p)
            if note:
                _issued.append(p)
        return g, True
    return torch.zeros(p.shape, dtype=F32, device=p.device), False


def _ret(buf, direct):
    return None if direct else buf


# Gradient-ready reports.  GradSync (optim.py) tags every parameter of ITS optimiser with `_ctclip_sync = self` -- a
# per-optimiser registration, so several trainers can live in one process -- to start that parameter's bucket all-reduce as
# soon as its gradient has been fully ISSUED into the arena (overlap with the rest of backward).  Parameters whose gradient
# autograd delivers (a Function that returns it) are reported by a post-accumulate hook instead; the Functions below
# accumulate in place and return None, so they have to say themselves when they are done: grad_slot() notes the parameter,
# announce_grads() at the end of the backward reports every noted one.
_issued = []


def announce_grads():
    if _issued:
        ps = list(_issued)
        _issued.clear()
        for p in ps:
            sync = getattr(p, "_ctclip_sync", None)
            if sync is not None:
                sync.param_ready(p)


def announces(backward):
    """Decorator for a Function.backward that accumulates parameter gradients in place: report them once it has issued
    all of its kernels."""
    def wrapped(ctx, *grads):
        out = backward(ctx, *grads)
        announce_grads()
        return out
    wrapped.__name__ = backward.__name__
    return wrapped


def _grad_ready(p, direct):
    sync = getattr(p, "_ctclip_sync", None)
    if direct and sync is not None:
        sync.param_ready(p)


def _tag16(t32, t16):
    """Attach the bf16 copy a kernel produced for free to the f32 gradient it mirrors.  The next Function.backward picks
    it up with _get16 instead of launching a cast kernel; the pointer check makes a stale or re-wrapped tensor harmless."""
    t32._ctclip_bf16 = (t32.data_ptr(), t32.shape, t16)
    return t32


def _get16(t32):
    tag = getattr(t32, "_ctclip_bf16", None)
    if tag is not None and tag[0] == t32.data_ptr() and tag[1] == t32.shape and t32.is_contiguous():
        return tag[2].reshape(-1, t32.shape[-1])
    return cast16(_c(t32).reshape(-1, t32.shape[-1]))


# ---------------------------------------------------------------------------------------------------
# LayerNorm as a module-level op (Transformer.norm_out, patch-embed tail)
# ---------------------------------------------------------------------------------------------------
class LayerNormFn(Function):
    """LayerNorm over the last dimension.  swap = (A, C): the rows of x are [B][A][C] and the result comes back as
    [B, C, A, d] -- the CT-ViT's token re-ordering between its spatial and temporal transformers (ctvit.py:96,99,101)
    done by the kernel's row addressing instead of a separate permutation pass over the stream."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, swap=None):
        shape = x.shape
        x2 = _c(x).reshape(-1, shape[-1])
        if swap is None:
            _, y, mean, rstd = layernorm(x2, gamma, beta, eps, want16=False, want32=True)
            y = y.reshape(shape)
        else:
            A, C = int(swap[0]), int(swap[1])
            rows, dim = x2.shape
            if rows % (A * C):
                raise ValueError(f"LayerNormFn: {rows} rows are not a whole number of [{A}][{C}] blocks")
            y = torch.empty(rows // (A * C), C, A, dim, dtype=F32, device=x2.device)
            mean = torch.empty(rows, dtype=F32, device=x2.device)
            rstd = torch.empty(rows, dtype=F32, device=x2.device)
            hip.layernorm_swap_fwd(x2, gamma.detach(), None if beta is None else beta.detach(), y, mean, rstd, rows, dim,
                                   float(eps), A, C)
        ctx.save_for_backward(x2, mean, rstd)
        ctx.params = (gamma, beta)
        ctx.swap, ctx.xshape = swap, shape
        return y

    @staticmethod
    @announces
    def backward(ctx, dy):
        x2, mean, rstd = ctx.saved_tensors
        gamma, beta = ctx.params
        dx = torch.empty_like(x2)
        dx16 = torch.empty(x2.shape, dtype=BF16, device=x2.device)
        dg, dg_d = grad_slot(gamma)
        db, db_d = grad_slot(beta) if beta is not None else (None, True)
        if ctx.swap is None:
            hip.layernorm_bwd(_c(dy).reshape(x2.shape), x2, gamma.detach(), mean, rstd, None, dx, dx16, dg, db, x2.shape[0],
                              x2.shape[1])
        else:
            hip.layernorm_swap_bwd(_c(dy), x2, gamma.detach(), mean, rstd, dx, dx16, dg, db, x2.shape[0], x2.shape[1],
                                   int(ctx.swap[0]), int(ctx.swap[1]))
        return _tag16(dx.reshape(ctx.xshape), dx16), _ret(dg, dg_d), _ret(db, db_d), None, None


# ---------------------------------------------------------------------------------------------------
# PEG
# ---------------------------------------------------------------------------------------------------
def peg_fused_ok(h, w, d):
    """Grids ctclip_peg_bwd_fused takes (csrc/peg.hip: 12 positions per thread, 8 channel pairs per workgroup, three dy planes in LDS)."""
    if d % 16:
        return False
    strips = (w + 11) // 12
    threads = (h * strips * 8 + 63) // 64 * 64
    pitch = strips * 13 + 3
    while pitch % 4 != 2:
        pitch += 1
    lds = (3 * (h + 2) * pitch * 8 + 27 * 8) * 8
    return threads <= 512 and lds <= 160 * 1024


# CTCLIP_PEG_FUSED=0 keeps the two backward kernels (A/B runs)
PEG_FUSED = os.environ.get("CTCLIP_PEG_FUSED", "1") != "0"


class PegFn(Function):
    """attention.py:55-83 (+ the residual of :325 when `residual`)."""

    @staticmethod
    def forward(ctx, x, weight, bias, shape, residual):
        b, t, h, w = (int(s) for s in shape)
        d = x.shape[-1]
        xc = _c(x)
        w27 = weight.detach().reshape(d, 27).t().contiguous()
        y = torch.empty_like(xc)
        y16 = torch.empty(xc.shape, dtype=BF16, device=xc.device)      # the next op (attention) wants its input in bf16 too
        hip.peg_fwd(xc, w27, bias.detach(), y, y16, b, t, h, w, d, int(residual))
        ctx.save_for_backward(xc, w27)
        ctx.geom = (b, t, h, w, d, int(residual))
        ctx.params = (weight, bias)
        ctx.mark_non_differentiable(y16)
        ctx.set_materialize_grads(False)      # or autograd zero-fills a [tokens, d] bf16 "gradient" of y16 for every backward
        return y, y16

    @staticmethod
    @announces
    def backward(ctx, dy, _dy16):
        if dy is None:
            return None, None, None, None, None
        xc, w27 = ctx.saved_tensors
        b, t, h, w, d, residual = ctx.geom
        dyc = _c(dy)
        dx = torch.empty_like(dyc)
        dx16 = torch.empty(dyc.shape, dtype=BF16, device=dyc.device)
        fused = PEG_FUSED and peg_fused_ok(h, w, d)                # data + weight gradient in one pass over dy
        if not fused:
            hip.peg_bwd_data(dyc, w27, dx, dx16, b, t, h, w, d, residual)
        _tag16(dx, dx16)
        p_w, p_b = ctx.params
        gw, dw_direct = grad_slot(p_w, note=False)
        gb, db_direct = grad_slot(p_b, note=False)
        dw27 = torch.zeros(27, d, dtype=F32, device=dy.device)
        if dw_direct and db_direct:
            _issued.extend(q for q in (p_w, p_b) if getattr(q, "_ctclip_sync", None) is not None)
            # the tap-major result is folded into the [d,1,3,3,3] gradient in place
            if fused:
                hip.peg_bwd_fused(dyc, xc, w27, dx, dx16, dw27, gb, b, t, h, w, d, residual)
            else:
                hip.peg_bwd_weight(dyc, xc, dw27, gb, b, t, h, w, d)
            gw.view(d, 27).add_(dw27.t())
            return dx, None, None, None, None
        db = torch.zeros(d, dtype=F32, device=dy.device)
        if fused:
            hip.peg_bwd_fused(dyc, xc, w27, dx, dx16, dw27, db, b, t, h, w, d, residual)
        else:
            hip.peg_bwd_weight(dyc, xc, dw27, db, b, t, h, w, d)
        return dx, dw27.t().reshape(d, 1, 3, 3, 3), db, None, None


# ---------------------------------------------------------------------------------------------------
# Attention block: LN -> q / kv projections -> cosine-sim attention -> out projection (+ residual)
# ---------------------------------------------------------------------------------------------------
LOG2E = 1.4426950408889634
LN2 = 0.6931471805599453
# q / k cosine normalisation inside the projections' GEMM epilogue (ctclip_gemm_bf16_headnorm); CTCLIP_HEADNORM_IN_GEMM=0 keeps
# the separate head-norm pass over the raw projections (A/B runs, and the path of head sizes other than 32)
HEADNORM_IN_GEMM = os.environ.get("CTCLIP_HEADNORM_IN_GEMM", "1") != "0"


def deterministic() -> bool:
    """True when bit-reproducible results are asked for -- torch.use_deterministic_algorithms(True), the switch the
    reference's attribution code sets at import (src/utils/visualizations.py:29-39).  Almost everything on this path is
    reproducible unconditionally (include/ctclip_hip.h, "reproducibility"); the flag only selects the ordered form of
    the one sum whose fast form is order-dependent: the relative-position d(bias) of the spatial attention."""
    return torch.are_deterministic_algorithms_enabled()


def attn_head_major_ok(n, dh, dp, dim, heads, want_probs):
    """Shapes the head-major spatial-attention kernels (csrc/attention_hm.hip) take: d_head 32, whole 32-token tiles, at most
    20 of them (the d(bias) tiles of a query block and the wave images fill the LDS), an even number of heads (the GEMM
    epilogue that writes the layout works on 64-column slabs), and nobody asking for the probabilities."""
    return (dh == dp == 32 and n % 32 == 0 and 1 <= n // 32 <= 20 and dim % 32 == 0 and heads % 2 == 0
            and not want_probs)


class AttentionFn(Function):
    """attention.py:126-182 for self-attention without null-kv / mask / causal (the CT-ViT configuration).

    bias_t is the autograd-visible bias input: None, a dense [heads,n,n] tensor, or (table mode) the
    [R, heads] position table whose dense expansion / index map come in `aux`.

    Logits live in the log2 domain on every path: head-norm multiplies q by scale * log2(e) and the attention kernels are
    told scale = ln 2 (natural logit = ln2 * q.k + bias), so no kernel multiplies a score tile by a constant.  Shapes
    attn_head_major_ok() accepts run on head-major q / k / v / dO ([sequence][head][token][32], written that way by
    head-norm and by the kv / out-projection-gradient GEMM epilogues) with the static softmax shift of ctclip_attn_shift."""

    @staticmethod
    def forward(ctx, x, gamma, wq, wkv, q_scale, k_scale, wout, bias_t, sh, cfg, aux):
        heads, dh, dp, scale, residual, want_probs = cfg
        nseq, n, dim = x.shape
        M = nseq * n
        inner = heads * dp
        dev = x.device
        hm = attn_head_major_ok(n, dh, dp, dim, heads, want_probs)
        qmult = float(scale) * LOG2E
        x2 = _c(x).reshape(M, dim)
        # production head size: gamma is folded into the q projection (shadow "wqg"), the saved GEMM operand n1 is the PLAIN
        # normalised row and the backward needs neither x nor the mean (ctclip_layernorm_bwd_xhat)
        fold = "wqg" in sh
        n1, _, mean, rstd = layernorm(x2, None if fold else gamma.detach(), None, 1e-5)
        x16 = aux.get("x16")
        xb = x16.reshape(M, dim) if (x16 is not None and x16.numel() == M * dim and x16.is_contiguous()) else cast16(x2)
        wq16 = sh["wqg" if fold else "wq"]
        qinv = torch.empty(M, heads, dtype=F32, device=dev)
        kinv = torch.empty(M, heads, dtype=F32, device=dev)
        o = torch.empty(M, inner, dtype=BF16, device=dev)
        lse = torch.empty(nseq, heads, n, dtype=F32, device=dev)
        kind = aux["kind"]                                        # None | "dense" | "table"
        bias_dense = None
        if kind == "dense":
            bias_dense = _c(bias_t.detach().to(F32))
        elif kind == "table":
            bias_dense = aux["dense"]
        # head size 32, an even number of heads: the cosine normalisation of q and k (attention.py:146-153) runs in the REGISTER
        # EPILOGUE of their projections (ctclip_gemm_bf16_headnorm) -- the raw q / k are never written, read back or kept; the
        # backward works from the normalised rows and 1 / norm
        fusedhn = HEADNORM_IN_GEMM and dp == 32 and inner % 64 == 0 and dim % 32 == 0
        q = None
        if hm:
            kv = torch.empty(2, nseq, heads, n, dp, dtype=BF16, device=dev)          # QUIRK attention.py:138: kv from un-normalised x
            qh = torch.empty(nseq, heads, n, dp, dtype=BF16, device=dev)
            if fusedhn:
                hip.gemm_bf16_headnorm(xb, sh["wkv"], kv, kinv, sh["k_scale"], M, 2 * inner, dim, xb.stride(0),
                                       sh["wkv"].stride(0), 0, n, heads, inner, 1.0)
                hip.gemm_bf16_headnorm(n1, wq16, qh, qinv, sh["q_scale"], M, inner, dim, n1.stride(0), wq16.stride(0), 0, n,
                                       heads, inner, qmult)
                kh = kv[0]
            else:
                q = gemm(n1, wq16, M, inner, dim)
                hip.gemm_bf16_headmajor(xb, sh["wkv"], kv, M, 2 * inner, dim, xb.stride(0), sh["wkv"].stride(0), n, heads)
                kh = torch.empty(nseq, heads, n, dp, dtype=BF16, device=dev)
                hip.headnorm_fwd(q, sh["q_scale"], qh, qinv, M, heads, dp, inner, 0, qmult, 0, n)
                hip.headnorm_fwd(kv[0], sh["k_scale"], kh, kinv, M, heads, dp, 0, 0, 1.0, n, n)
            shift = torch.empty(heads + 1, dtype=F32, device=dev)
            if kind == "table":
                tb = bias_t.detach()
                hip.attn_shift(sh["q_scale"], sh["k_scale"], dp, qmult, tb, tb.shape[0], 1, tb.stride(0), heads, shift)
            elif kind == "dense":
                hip.attn_shift(sh["q_scale"], sh["k_scale"], dp, qmult, bias_dense, n * n, n * n, 1, heads, shift)
            else:
                hip.attn_shift(sh["q_scale"], sh["k_scale"], dp, qmult, None, 0, 0, 0, heads, shift)
            hip.attn_hm_fwd(qh, kh, kv[1], o, lse, bias_dense, shift, nseq, n, heads, inner)
        elif fusedhn:
            kv = torch.empty(M, 2 * inner, dtype=BF16, device=dev)   # [normalised k | v]; QUIRK attention.py:138: from un-normalised x
            qh = torch.empty(M, inner, dtype=BF16, device=dev)
            hip.gemm_bf16_headnorm(xb, sh["wkv"], kv, kinv, sh["k_scale"], M, 2 * inner, dim, xb.stride(0), sh["wkv"].stride(0),
                                   2 * inner, 0, heads, inner, 1.0)
            hip.gemm_bf16_headnorm(n1, wq16, qh, qinv, sh["q_scale"], M, inner, dim, n1.stride(0), wq16.stride(0), inner, 0, heads,
                                   inner, qmult)
            kh = kv[:, :inner]
            hip.attn_fwd(qh, kh, kv[:, inner:], o, lse, bias_dense, None, nseq, n, heads, dp, inner, 2 * inner, 2 * inner,
                         inner, LN2)
        else:
            q = gemm(n1, wq16, M, inner, dim)
            kv = gemm(xb, sh["wkv"], M, 2 * inner, dim)           # QUIRK attention.py:138: kv from un-normalised x
            qh = torch.empty_like(q)
            kh = torch.empty(M, inner, dtype=BF16, device=dev)
            hip.headnorm_fwd(q, sh["q_scale"], qh, qinv, M, heads, dp, inner, inner, qmult, 0, 0)
            hip.headnorm_fwd(kv, sh["k_scale"], kh, kinv, M, heads, dp, 2 * inner, inner, 1.0, 0, 0)
            hip.attn_fwd(qh, kh, kv[:, inner:], o, lse, bias_dense, None, nseq, n, heads, dp, inner, inner, 2 * inner,
                         inner, LN2)
        y = gemm(o, sh["wout"], M, dim, inner, out_dtype=F32, resid=x2 if residual else None)
        if want_probs:
            probs = torch.empty(nseq, heads, n, n, dtype=F32, device=dev)
            hip.attn_probs(qh, kh, lse, bias_dense, None, probs, nseq, n, heads, dp, inner, kh.stride(0), LN2)
        else:
            probs = x.new_empty(0)
        keep_x = x2.new_empty(0) if fold else x2                  # folded: the f32 input row is not kept for the backward
        # fused head-norm: the raw q does not exist and kh is a view of kv -- neither is saved on its own
        ctx.save_for_backward(keep_x, gamma, mean, rstd, n1, xb, q if q is not None else x2.new_empty(0), kv, qh,
                              kh if not fusedhn else x2.new_empty(0), qinv, kinv, o, lse,
                              bias_dense if bias_dense is not None else x2.new_empty(0))
        ctx.sh, ctx.cfg, ctx.aux, ctx.shape, ctx.hm, ctx.fusedhn = sh, cfg, aux, (nseq, n, dim), hm, fusedhn
        ctx.params = (gamma, wq, wkv, q_scale, k_scale, wout)
        ctx.mark_non_differentiable(probs)
        return y.reshape(nseq, n, dim), probs

    @staticmethod
    @announces
    def backward(ctx, dy, _dprobs):
        x2, gamma, mean, rstd, n1, xb, q, kv, qh, kh, qinv, kinv, o, lse, bias_dense = ctx.saved_tensors
        sh, aux, hm, fusedhn = ctx.sh, ctx.aux, ctx.hm, ctx.fusedhn
        heads, dh, dp, scale, residual, _ = ctx.cfg
        nseq, n, dim = ctx.shape
        M, inner = nseq * n, heads * dp
        if fusedhn:                                    # the normalised k is the first half of kv; the raw q / k do not exist
            kh = kv[0] if hm else kv[:, :inner]
        ldk = inner if (hm or not fusedhn) else 2 * inner
        dev = dy.device
        kind = aux["kind"]
        qmult = float(scale) * LOG2E
        if bias_dense.numel() == 0:
            bias_dense = None
        dy2 = _c(dy).reshape(M, dim)
        dyb = _get16(dy)
        dqh = torch.empty(M, inner, dtype=BF16, device=dev)
        dkh = torch.empty(M, inner, dtype=BF16, device=dev)
        # fused form (production head size, 8 heads): [rstd dq | dk | dv] is ONE bf16 operand of the block's input-gradient GEMM, whose
        # epilogue applies the LayerNorm backward (ctclip_gemm_bf16_lnbwd); dkv is its last 2 inner columns
        fused = "wcat" in sh and dp == dh == 32 and heads == 8 and dim % 8 == 0
        dcat = torch.empty(M, 3 * inner, dtype=BF16, device=dev) if fused else None
        dkv = dcat[:, inner:] if fused else torch.empty(M, 2 * inner, dtype=BF16, device=dev)
        lkv = dkv.stride(0)
        delta = torch.empty(nseq, heads, n, dtype=F32, device=dev)
        dbias_dense = dtable = rel = None
        tsize = gh = gw = 0
        if kind == "dense" and ctx.needs_input_grad[7]:
            dbias_dense = torch.zeros(heads, n, n, dtype=F32, device=dev)
        elif kind == "table":
            tsize = aux["rows"]
            dtable = torch.zeros(heads, tsize, dtype=F32, device=dev)
            gh, gw = aux.get("grid", (0, 0))
            rel = None if gw else aux["relidx"]
        # deterministic algorithms requested: the gradient passes run WITHOUT the bias gradient (its fast form adds the
        # sequences' dS tiles under LDS locks, in arrival order) and ctclip_attn_dbias_ordered sums them in sequence order
        ordered = deterministic() and (dbias_dense is not None or dtable is not None)
        if ordered and dtable is not None and not gw:
            raise NotImplementedError("deterministic d(bias) of an index table that is not a 2-D relative-position grid")
        fast_dense, fast_table = (None, None) if ordered else (dbias_dense, dtable)
        f_rel, f_ts, f_gh, f_gw = (None, 0, 0, 0) if ordered else (rel, tsize, gh, gw)
        if hm:
            # d(o) = dy Wout as a k-major x k-major product with the transposed weight shadow, written head-major
            do = torch.empty(nseq, heads, n, dp, dtype=BF16, device=dev)
            hip.gemm_bf16_headmajor(dyb, sh["woutT"], do, M, inner, dim, dyb.stride(0), sh["woutT"].stride(0), n, heads)
            hip.attn_hm_bwd(qh, kh, kv[1], o, do, lse, delta, dqh, dkh, dkv[:, inner:], bias_dense, fast_dense, f_rel, fast_table,
                            f_ts, f_gh, f_gw, nseq, n, heads, inner, inner, inner, lkv)
        else:
            do = dgrad(dyb, sh["wout"], M, dim, inner, out_dtype=BF16, wT16=sh.get("woutT"))
            hip.attn_bwd(qh, kh, kv[:, inner:], o, do, lse, delta, dqh, dkh, dkv[:, inner:], bias_dense, None,
                         fast_dense, f_rel, fast_table, f_ts, f_gh, f_gw, nseq, n, heads, dp,
                         inner, ldk, 2 * inner, inner, inner, inner, inner, lkv, LN2)
        if ordered:
            dense = dbias_dense if dbias_dense is not None else torch.zeros(heads, n, n, dtype=F32, device=dev)
            if hm:
                hip.attn_dbias_ordered(qh, kh, kv[1], do, lse, delta, bias_dense, dense, nseq, n, heads, 1, 0, 0, 0, 0, LN2)
            else:
                hip.attn_dbias_ordered(qh, kh, kv[:, inner:], do, lse, delta, bias_dense, dense, nseq, n, heads, 0,
                                       inner, ldk, 2 * inner, inner, LN2)
            if dtable is not None:
                hip.attn_dbias_table(dense, dtable, heads, gh, gw)
        # what the head-norm backward reads as "x": the raw projections, or (fusedhn) the normalised rows the forward kept
        k_raw, k_ld, k_hm = (kv[0], 0, n) if hm else (kv, 2 * inner, 0)
        q_x, q_ld, q_hm = (qh, (0 if hm else inner), (n if hm else 0)) if fusedhn else (q, inner, 0)
        xn = int(fusedhn)
        p_gamma, p_wq, p_wkv, p_qs, p_ks, p_wout = ctx.params
        dq = torch.empty(M, inner, dtype=BF16, device=dev)
        dbias = None
        if kind == "dense":
            dbias = dbias_dense
        elif kind == "table":
            dbias = dtable.t().contiguous()          # [R, heads] like the MLP output
        dx = torch.empty(M, dim, dtype=F32, device=dev)
        if dp == dh:                                   # production layout: gradients land in param-shaped buffers
            gqs, d1 = grad_slot(p_qs)
            gks, d2 = grad_slot(p_ks)
            gwq, d3 = grad_slot(p_wq)
            gwkv, d4 = grad_slot(p_wkv)
            gwo, d5 = grad_slot(p_wout)
            gg, d6 = grad_slot(p_gamma)
            wgrad(dyb, o, dim, inner, M, out=gwo)
            c1 = c2 = None
            if fused:
                c1 = torch.empty(M, dtype=F32, device=dev)
                c2 = torch.empty(M, dtype=F32, device=dev)
                hip.headnorm_bwd_ln(dqh, q_x, qinv, sh["q_scale"], dq, gqs, M, heads, dp, inner, q_ld, inner, qmult,
                                    rstd, sh["wbar"], dim, dcat, dcat.stride(0), c1, c2, q_hm, xn)
            else:
                hip.headnorm_bwd(dqh, q_x, qinv, sh["q_scale"], dq, gqs, M, heads, dp, inner, q_ld, inner, qmult, q_hm, xn)
            hip.headnorm_bwd(dkh, k_raw, kinv, sh["k_scale"], dkv, gks, M, heads, dp, inner, k_ld, lkv, 1.0, k_hm, xn)
            # both data gradients leave their (store-bound, K = 256 / 512) GEMMs in bf16; the f32 residual-path gradient
            # dy2 is added inside the LayerNorm backward, so the residual stream itself never passes through bf16
            fold = "wqg" in sh
            if fused:
                G = wgrad(dq, n1, inner, dim, M)
                hip.patch_affine_bwd(G, None, p_wq.detach(), gamma.detach(), None, gwq, gg, None, inner, dim, dim, 0)
                wgrad(dkv, xb, 2 * inner, dim, M, out=gwkv)
                dx16 = torch.empty(M, dim, dtype=BF16, device=dev) if aux.get("x16") is None else None
                hip.gemm_bf16_lnbwd(dcat, sh["wcat"], dx, dx16, M, dim, 3 * inner, dcat.stride(0), sh["wcat"].stride(0), n1, c1, c2,
                                    dy2 if residual else None)
                dxr = dx.reshape(nseq, n, dim)
                return (_tag16(dxr, dx16) if dx16 is not None else dxr, _ret(gg, d6), _ret(gwq, d3), _ret(gwkv, d4), _ret(gqs, d1),
                        _ret(gks, d2), _ret(gwo, d5), dbias, None, None, None)
            dn1 = dgrad(dq, sh["wqg" if fold else "wq"], M, inner, dim, out_dtype=BF16, wT16=sh.get("wqgT" if fold else "wqT"))
            if fold:
                # G = dq^T xhat is the one weight-gradient product; d(Wq) = G gamma, d(gamma) = sum_n Wq G (exact, no division)
                G = wgrad(dq, n1, inner, dim, M)
                hip.patch_affine_bwd(G, None, p_wq.detach(), gamma.detach(), None, gwq, gg, None, inner, dim, dim, 0)
            else:
                wgrad(dq, n1, inner, dim, M, out=gwq)
            dxkv = dgrad(dkv, sh["wkv"], M, 2 * inner, dim, out_dtype=BF16, wT16=sh.get("wkvT"))
            wgrad(dkv, xb, 2 * inner, dim, M, out=gwkv)
            # x came from a PEG (it handed us its bf16 copy): this gradient goes back into the PEG backward, which reads
            # f32 only, so a bf16 mirror of it would be a gigabyte written for nobody
            dx16 = torch.empty(M, dim, dtype=BF16, device=dev) if aux.get("x16") is None else None
            if fold:
                hip.layernorm_bwd_xhat(dn1, n1, rstd, dy2 if residual else None, dxkv, dx, dx16, M, dim)
            else:
                hip.layernorm_bwd_bf16(dn1, x2, gamma, mean, rstd, dy2 if residual else None, dxkv, dx, dx16, gg, None, M, dim)
            dxr = dx.reshape(nseq, n, dim)
            return (_tag16(dxr, dx16) if dx16 is not None else dxr, _ret(gg, d6), _ret(gwq, d3), _ret(gwkv, d4), _ret(gqs, d1), _ret(gks, d2),
                    _ret(gwo, d5), dbias, None, None, None)
        dwout = wgrad(dyb, o, dim, inner, M)
        dqs = torch.zeros(dp, dtype=F32, device=dev)
        dks = torch.zeros(dp, dtype=F32, device=dev)
        hip.headnorm_bwd(dqh, q_x, qinv, sh["q_scale"], dq, dqs, M, heads, dp, inner, q_ld, inner, qmult, q_hm, xn)
        hip.headnorm_bwd(dkh, kv, kinv, sh["k_scale"], dkv, dks, M, heads, dp, inner, 2 * inner, 2 * inner, 1.0, 0, xn)
        dn1 = dgrad(dq, sh["wq"], M, inner, dim, wT16=sh.get("wqT"))
        dwq = wgrad(dq, n1, inner, dim, M)
        dxkv = dgrad(dkv, sh["wkv"], M, 2 * inner, dim, resid=dy2 if residual else None, wT16=sh.get("wkvT"))
        dwkv = wgrad(dkv, xb, 2 * inner, dim, M)
        dgamma = torch.zeros_like(gamma)
        hip.layernorm_bwd(dn1, x2, gamma, mean, rstd, dxkv, dx, None, dgamma, None, M, dim)
        dwq = unpad_head_rows(dwq, heads, dh, dp)
        dwk = unpad_head_rows(dwkv[:inner], heads, dh, dp)
        dwv = unpad_head_rows(dwkv[inner:], heads, dh, dp)
        dwout = unpad_head_rows(dwout.t(), heads, dh, dp).t()
        return (dx.reshape(nseq, n, dim), dgamma, dwq, torch.cat((dwk, dwv), 0), dqs[:dh], dks[:dh], dwout, dbias,
                None, None, None)


# ---------------------------------------------------------------------------------------------------
# GEGLU feed-forward: LN -> Linear(dim, 2I) -> GEGLU -> Linear(I, dim) (+ residual)   attention.py:38-51
# ---------------------------------------------------------------------------------------------------
class FeedForwardFn(Function):
    @staticmethod
    def forward(ctx, x, ln_w, ln_b, w1, w2, sh, residual):
        shape = x.shape
        dim = shape[-1]
        x2 = _c(x).reshape(-1, dim)
        M = x2.shape[0]
        I, Ip = sh["inner"], sh["inner_p"]
        n2, _, mean, rstd = layernorm(x2, ln_w.detach(), ln_b.detach(), 1e-5)
        h = torch.empty(M, 2 * Ip, dtype=BF16, device=x.device)    # [val 32 | gate 32 | ...] blocks (GEGLU_BLOCK), kept for the backward
        g = torch.empty(M, Ip, dtype=BF16, device=x.device)
        hip.gemm_bf16_geglu(n2, sh["w1"], h, g, M, Ip, dim, n2.stride(0), sh["w1"].stride(0), 2 * Ip, Ip)
        y = gemm(g, sh["w2"], M, dim, Ip, out_dtype=F32, resid=x2 if residual else None)
        ctx.save_for_backward(x2, ln_w, mean, rstd, n2, h, g)
        ctx.sh, ctx.residual = sh, residual
        ctx.params = (ln_w, ln_b, w1, w2)
        return y.reshape(shape)

    @staticmethod
    @announces
    def backward(ctx, dy):
        x2, ln_w, mean, rstd, n2, h, g = ctx.saved_tensors
        if getattr(ctx, "h_consumed", False):
            raise RuntimeError("FeedForwardFn.backward ran twice on one graph (retain_graph=True): the GEGLU pre-activations "
                               "are overwritten by their gradients in place during the first pass")
        ctx.h_consumed = True
        sh = ctx.sh
        M, dim = x2.shape
        I, Ip = sh["inner"], sh["inner_p"]
        dy2 = _c(dy).reshape(M, dim)
        dyb = _get16(dy)
        p_lw, p_lb, p_w1, p_w2 = ctx.params
        gw1, d1 = grad_slot(p_w1)
        gw2, d2 = grad_slot(p_w2)
        glw, d3 = grad_slot(p_lw)
        glb, d4 = grad_slot(p_lb)
        wgrad(dyb, g, dim, I, M, out=gw2)                          # g's zero pad columns I..Ip-1 are simply not produced
        # dg = dy W2 and the GEGLU backward in one pass: dg stays on chip, h is overwritten with d(h) in place (h is dead
        # after this; a second backward through the same graph is not supported)
        w2T = sh["w2T"]
        small = ((M + 255) // 256) * ((Ip + 255) // 256) < 192
        scratch = torch.empty(M, Ip, dtype=BF16, device=dy.device) if small else None
        hip.gemm_bf16_geglu_bwd(dyb, w2T, h, scratch, M, Ip, dim, dyb.stride(0), w2T.stride(0), 2 * Ip, Ip)
        dh = h
        dn2 = dgrad(dh, sh["w1"], M, 2 * Ip, dim, out_dtype=BF16, wT16=sh.get("w1T"))
        # one product over all 2*Ip columns of dh (the pad columns are zero): n2 is streamed once and 2*Ip = 2816 is a
        # whole number of 256-row tiles, where two I = 1365-row products each round up to six
        def ff1_wgrad():
            gw1p = wgrad(dh, n2, 2 * Ip, dim, M)                   # rows in the interleaved [val | gate] block order
            hip.geglu_wgrad_unblock(gw1p, gw1, I, GEGLU_BLOCK, dim)    # += value rows 0..I-1, gate rows I..2I-1 of the reference weight
        ff1_wgrad()
        dx = torch.empty(M, dim, dtype=F32, device=dy.device)
        dx16 = torch.empty(M, dim, dtype=BF16, device=dy.device)
        hip.layernorm_bwd_bf16(dn2, x2, ln_w, mean, rstd, dy2 if ctx.residual else None, None, dx, dx16, glw, glb, M, dim)
        return _tag16(dx.reshape(dy.shape), dx16), _ret(glw, d3), _ret(glb, d4), _ret(gw1, d1), _ret(gw2, d2), None, None


# ---------------------------------------------------------------------------------------------------
# f32 linear (tiny matrices where bf16 would cost parity: position MLP, text latent projection)
# ---------------------------------------------------------------------------------------------------
def sgemm(A, B, M, N, K, *, a_kmajor=True, b_kmajor=True, bias=None, act=0, aux=None, alpha=1.0, alpha_dev=None,
          alpha_exp=False, out=None, accumulate=False, slope=0.1):
    if out is None:
        out = torch.empty(M, N, dtype=F32, device=A.device)
    hip.gemm_f32(A, B, out, bias, aux, M, N, K, A.stride(0), B.stride(0), out.stride(0),
                 0 if aux is None else aux.stride(0), int(a_kmajor), int(b_kmajor), float(alpha), alpha_dev,
                 int(alpha_exp), act, float(slope), int(accumulate))
    return out


class LinearF32Fn(Function):
    """y = act(x W^T + b) in f32; act in {None, 'leaky'} (slope 0.1)."""

    @staticmethod
    def forward(ctx, x, w, b, leaky):
        x2 = _c(x).reshape(-1, x.shape[-1])
        M, K = x2.shape
        N = w.shape[0]
        y = sgemm(x2, _c(w.detach()), M, N, K, bias=None if b is None else b.detach(), act=2 if leaky else 0)
        ctx.save_for_backward(x2, w, y)
        ctx.leaky, ctx.has_b, ctx.xshape = leaky, b is not None, x.shape
        return y.reshape(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, w, y = ctx.saved_tensors
        M, K = x2.shape
        N = w.shape[0]
        dy2 = _c(dy).reshape(M, N)
        if ctx.leaky:                                    # dy * leaky'(pre-activation); sign(pre) == sign(y)
            t = torch.empty_like(dy2)
            hip.leaky_bwd(dy2, y, t, dy2.numel(), 0.1)
            dy2 = t
        dx = sgemm(dy2, _c(w.detach()), M, K, N, a_kmajor=True, b_kmajor=False)
        dw = sgemm(dy2, x2, N, K, M, a_kmajor=False, b_kmajor=False)
        db = colsum(dy2) if ctx.has_b else None
        return dx.reshape(ctx.xshape), dw, db, None


# ---------------------------------------------------------------------------------------------------
# patch embedding: gather + LN(F) + Linear(F, dim) + bias + LN(dim)          ctvit.py:44-52
# ---------------------------------------------------------------------------------------------------
# CTCLIP_PATCH_FUSED=0 keeps the unfused chain (gather + LayerNorm kernel -> [tokens, F] bf16 operand -> GEMM) for A/B runs
PATCH_FUSED = os.environ.get("CTCLIP_PATCH_FUSED", "1") != "0"


def patch_embed_fusable(vol, C, Dz, Hy, Wx, pt, p, dim, tokens, sh):
    """The geometries ctclip_patch_embed_fused / ctclip_patch_wgrad_fused take (include/ctclip_hip.h); everything else -- f32
    volumes, odd tubelets, an embedding width other than 512 -- stays on the unfused chain."""
    F_ = C * pt * p * p
    return (PATCH_FUSED and vol.dtype == BF16 and dim == 512 and p % 4 == 0 and Wx % 4 == 0 and F_ % 32 == 0 and 128 <= F_ <= 4096
            and tokens % 32 == 0 and C * Dz * Hy * Wx < 2 ** 31 and "wsum" in sh and vol.data_ptr() % 16 == 0)


class PatchEmbedFn(Function):
    """The affine part of LayerNorm(F) is folded into the projection (include/ctclip_hip.h: ctclip_patch_affine_fold):
    z = xhat (W gamma)^T + (b + W beta).  The GEMM operand `A` is then the plain normalised tubelet row, and the backward
    needs ONE product over the tokens, G = dz^T xhat: d(W), d(gamma), d(beta) all follow from G and colsum(dz)
    (ctclip_patch_affine_bwd) -- the [tokens, F] data gradient dz W and the pass over it are only formed when the volume
    itself is differentiated (integrated gradients)."""

    @staticmethod
    def forward(ctx, volume, ln1w, ln1b, w, b, ln2w, ln2b, sh, geom):
        patch, tpatch = geom
        B, C, Dz, Hy, Wx = volume.shape
        t, h, wt = Dz // tpatch, Hy // patch, Wx // patch
        F_ = C * tpatch * patch * patch
        dim = w.shape[0]
        M = B * t * h * wt
        vol = _c(volume)
        is16 = vol.dtype == BF16
        if not is16 and vol.dtype != F32:
            raise TypeError("volume must be float32 or bfloat16")
        ldA = pad8(F_)
        fused = patch_embed_fusable(vol, C, Dz, Hy, Wx, tpatch, patch, dim, M, sh)
        if fused:
            # ONE pass over the volume (csrc/patch_gemm.hip): the [tokens, F] normalised operand is never written; the per-token
            # (c, mean - c, rstd, mean) is what the backward needs to rebuild it from the volume
            tstat = torch.empty(M, 4, dtype=F32, device=vol.device)
            z = torch.empty(M, dim, dtype=F32, device=vol.device)
            hip.patch_embed_fused(vol, sh["w"], sh["w"].stride(0), sh["wsum"], sh["b"], z, dim, tstat, B, C, Dz, Hy, Wx, tpatch,
                                  patch, dim, 1e-5)
            A = mean1 = rstd1 = vol.new_empty(0)
        else:
            tstat = vol.new_empty(0)
            A = torch.empty(M, ldA, dtype=BF16, device=vol.device)
            mean1 = torch.empty(M, dtype=F32, device=vol.device)
            rstd1 = torch.empty(M, dtype=F32, device=vol.device)
            hip.patch_ln_fwd(vol, int(is16), None, None, A, mean1, rstd1, B, C, Dz, Hy, Wx, tpatch, patch, ldA, 1e-5)
            z = gemm(A, sh["w"], M, dim, ldA, out_dtype=F32, bias=sh["b"])
        _, y, mean2, rstd2 = layernorm(z, ln2w.detach(), ln2b.detach(), 1e-5, want16=False, want32=True)
        ctx.save_for_backward(vol, A, mean1, rstd1, z, mean2, rstd2, ln2w, tstat)
        ctx.sh, ctx.geom, ctx.dims, ctx.fused = sh, geom, (B, C, Dz, Hy, Wx, F_, ldA, dim, M), fused
        ctx.params = (ln1w, ln1b, w, b, ln2w, ln2b)
        return y.reshape(B, t, h, wt, dim)

    @staticmethod
    @announces
    def backward(ctx, dy):
        vol, A, mean1, rstd1, z, mean2, rstd2, ln2w, tstat = ctx.saved_tensors
        sh = ctx.sh
        patch, tpatch = ctx.geom
        B, C, Dz, Hy, Wx, F_, ldA, dim, M = ctx.dims
        dev = dy.device
        dy2 = _c(dy).reshape(M, dim)
        p_l1w, p_l1b, p_w, p_b, p_l2w, p_l2b = ctx.params
        dz = torch.empty(M, dim, dtype=F32, device=dev)
        dzb = torch.empty(M, dim, dtype=BF16, device=dev)
        d2w, k1 = grad_slot(p_l2w)
        d2b, k2 = grad_slot(p_l2b)
        hip.layernorm_bwd(dy2, z, ln2w, mean2, rstd2, None, dz, dzb, d2w, d2b, M, dim)
        db_now = colsum(dz)                                          # this call's sum: the folded terms below need it alone
        db, k3 = grad_slot(p_b)
        db += db_now
        dw, k4 = grad_slot(p_w)
        d1w, k5 = grad_slot(p_l1w)
        d1b, k6 = grad_slot(p_l1b)
        if ctx.fused:
            # dz^T xhat with xhat rebuilt from the volume; two extra columns carry the (mean - c) rstd term (ctclip_patch_wgrad_fused)
            G = torch.zeros(dim, F_ + 2, dtype=F32, device=dev)
            hip.patch_wgrad_fused(vol, dzb, dzb.stride(0), tstat, G, F_ + 2, B, C, Dz, Hy, Wx, tpatch, patch, dim)
            hip.patch_affine_bwd(G, db_now, _c(p_w.detach()), p_l1w.detach(), p_l1b.detach(), dw, d1w, d1b, dim, F_, F_ + 2, 2)
            if ctx.needs_input_grad[0]:
                mean1, rstd1 = tstat[:, 3].contiguous(), tstat[:, 2].contiguous()
        else:
            G = wgrad(dzb, A, dim, F_, M)                            # dz^T xhat, [dim, F] f32
            hip.patch_affine_bwd(G, db_now, _c(p_w.detach()), p_l1w.detach(), p_l1b.detach(), dw, d1w, d1b, dim, F_, F_, 0)
        dvol = None
        if ctx.needs_input_grad[0]:                    # input attribution only (integrated gradients)
            dA = dgrad(dzb, sh["w"], M, dim, ldA, out_dtype=BF16, wT16=sh.get("wT"))      # d(xhat) = dz (W gamma)
            dvol = torch.empty(vol.shape, dtype=F32, device=dev)
            hip.patch_ln_bwd_dx(vol, int(vol.dtype == BF16), dA, ldA, sh["ones"], mean1, rstd1, dvol, B, C, Dz, Hy, Wx,
                                tpatch, patch)
            dvol = dvol.to(vol.dtype)
        return dvol, _ret(d1w, k5), _ret(d1b, k6), _ret(dw, k4), _ret(db, k3), _ret(d2w, k1), _ret(d2b, k2), None, None


# ---------------------------------------------------------------------------------------------------
# token re-ordering between the spatial and temporal transformers (ctvit.py:96,99,101)
# ---------------------------------------------------------------------------------------------------
class SwapMiddleFn(Function):
    """[B, A, C, D] -> [B, C, A, D] (contiguous copy on device)."""

    @staticmethod
    def forward(ctx, x):
        B, A, C, D = x.shape
        out = torch.empty(B, C, A, D, dtype=F32, device=x.device)
        hip.swap_middle_f32(_c(x), out, B, A, C, D)
        return out

    @staticmethod
    def backward(ctx, dy):
        B, C, A, D = dy.shape
        out = torch.empty(B, A, C, D, dtype=F32, device=dy.device)
        hip.swap_middle_f32(_c(dy), out, B, C, A, D)
        return out


# Code groups of the VQ sweep (csrc/gemm3.hip, "CODE GROUPS"): measured at 64 / 88 pairs (profiles/r03_vq_code_groups.txt) the
# split lowers the fabric traffic by only 30 % (33.4 -> 23.5 GB per launch) and costs 2 - 14 % of time: one group stays the default.
VQ_CODE_GROUPS = int(os.environ.get("CTCLIP_VQ_CODE_GROUPS", "1"))

# ---------------------------------------------------------------------------------------------------
# VQ (cosine-sim codebook, straight-through)            ctvit.py:117-118
# ---------------------------------------------------------------------------------------------------
class VQFn(Function):
    """x [b, n, d] f32 -> (quantised [b,n,d] f32 with straight-through gradient to l2norm(x), indices [b,n])."""

    @staticmethod
    def forward(ctx, x, embed, embed16, forced_idx=None):
        b, n, d = x.shape
        M = b * n
        x2 = _c(x).reshape(M, d)
        xn16 = torch.empty(M, d, dtype=BF16, device=x.device)
        inv = torch.empty(M, dtype=F32, device=x.device)
        hip.rownorm_fwd(x2, xn16, None, inv, M, d, 1e-12)
        if forced_idx is not None:                                  # parity-test hook (VectorQuantize.forced_indices)
            idx = forced_idx.reshape(M).to(x.device, torch.long)
            quant = embed.index_select(0, idx)
            ctx.save_for_backward(x2, inv)
            ctx.mark_non_differentiable(idx)
            ctx.aux = (x2, inv)
            return quant.reshape(b, n, d), idx.reshape(b, n)
        ncodes = embed16.shape[0]
        # 16 candidates per code group = top-4 of each of 4 disjoint parts of the group
        groups = VQ_CODE_GROUPS if (ncodes % (256 * VQ_CODE_GROUPS) == 0 and d % 32 == 0) else 1
        ncand = 16 * groups
        pv = torch.empty(M, ncand, dtype=F32, device=x.device)
        pi = torch.empty(M, ncand, dtype=torch.int32, device=x.device)
        hip.vq_topk_grouped(embed16, xn16, pv, pi, ncodes, M, d, embed16.stride(0), xn16.stride(0), groups)
        idx = torch.empty(M, dtype=torch.long, device=x.device)
        quant = torch.empty(M, d, dtype=F32, device=x.device)
        # 2^-7: twice the worst-case bf16 rounding error of a unit-vector dot product -> exact f32 arg-max
        hip.vq_select(pv, pi, ncand, x2, inv, _c(embed), idx, quant, M, d, 2.0 ** -7)
        ctx.save_for_backward(x2, inv)
        ctx.mark_non_differentiable(idx)
        ctx.aux = (x2, inv)                                   # for the EMA update by the caller
        return quant.reshape(b, n, d), idx.reshape(b, n)

    @staticmethod
    def backward(ctx, dq, _didx):
        x2, inv = ctx.saved_tensors
        M, d = x2.shape
        dx = torch.empty_like(x2)
        hip.rownorm_bwd(_c(dq).reshape(M, d), x2, inv, dx, M, d)
        return dx.reshape(dq.shape), None, None, None


def vq_ema_accum(x2, inv, idx, ncodes, d):
    """Per-code statistics of the library's cosine-sim codebook update: -> (bins [ncodes], esum [ncodes, d], flat), the
    first two being views of the one flat buffer (so a data-parallel run reduces both with one collective).

    No atomics: the tokens are sorted by code (stable, so equal codes keep token order), the counts are the segment
    lengths, and ctclip_vq_ema_accum_sorted sums each code's rows in that order -- bit-reproducible, and faster than the
    ~450 M float atomics of the scatter-add form at the production shape."""
    M = x2.shape[0]
    flat = torch.zeros(ncodes * (d + 1), dtype=F32, device=x2.device)
    bins, esum = flat[:ncodes], flat[ncodes:].view(ncodes, d)
    code_sorted, order = torch.sort(idx.reshape(-1), stable=True)
    seg = torch.searchsorted(code_sorted, torch.arange(ncodes + 1, device=x2.device, dtype=code_sorted.dtype))
    bins.copy_((seg[1:] - seg[:-1]).to(F32))
    nchunks = (M + 255) // 256
    edge = torch.empty(2 * nchunks * d, dtype=F32, device=x2.device)
    edge_code = torch.empty(3 * nchunks, dtype=torch.long, device=x2.device)
    hip.vq_ema_accum_sorted(x2, inv, order, code_sorted, esum, edge, edge_code, M, d)
    return bins, esum, flat


def vq_ema_apply(embed, cluster_size, bins, esum, decay):
    """EMA codebook update from (reduced) statistics, in place on the buffers."""
    ncodes, d = embed.shape[-2], embed.shape[-1]
    hip.vq_ema_update(embed.reshape(ncodes, d), cluster_size.reshape(ncodes), bins, esum, ncodes, d, float(decay))


# ---------------------------------------------------------------------------------------------------
# CTCLIP tail: mean over depth + visual projection (bf16 MFMA, split-K over 294 912), latent norms, logits, loss
# ---------------------------------------------------------------------------------------------------
class VisualLatentFn(Function):
    """image_tokens [B,T,H,W,D] -> mean over T -> [B, H*W*D] @ W^T   (ctclip.py:111-112,116)."""

    @staticmethod
    def forward(ctx, tokens, w, w16):
        B, T = tokens.shape[0], tokens.shape[1]
        Fdim = tokens[0, 0].numel()
        a16 = torch.empty(B, Fdim, dtype=BF16, device=tokens.device)
        hip.mean_mid_fwd(_c(tokens), a16, None, B, T, Fdim)
        L = w16.shape[0]
        nk = (Fdim + 63) // 64
        split = max(1, min(nk, 1024 // max(1, (L + 127) // 128)))
        out = gemm(a16, w16, B, L, Fdim, out_dtype=F32, split_k=split, accumulate=True)
        ctx.save_for_backward(a16, w16)
        ctx.w, ctx.shape = w, tokens.shape
        return out

    @staticmethod
    def backward(ctx, dy):
        a16, w16 = ctx.saved_tensors
        B, Fdim = a16.shape
        L = w16.shape[0]
        T = ctx.shape[1]
        dyb = cast16(_c(dy))
        da = dgrad(dyb, w16, B, L, Fdim)                                   # [B, Fdim] f32
        dtok = torch.empty(ctx.shape, dtype=F32, device=dy.device)
        hip.mean_mid_bwd(da, dtok, B, T, Fdim)
        w = ctx.w
        if w.grad is not None and w.grad.is_contiguous() and w.grad.dtype == F32:
            # accumulate the (dim_latent x dim_image) weight gradient straight into the arena-backed .grad
            if getattr(w, "_ctclip_sync", None) is not None:
                w._ctclip_sync.before_write(w)
            gemm(dyb, a16, L, Fdim, B, a_kmajor=False, b_kmajor=False, out=w.grad, accumulate=True)
            dw = None
            _grad_ready(w, True)            # 53 % of all gradient bytes, ready at the very start of backward
        else:
            dw = wgrad(dyb, a16, L, Fdim, B)
        return dtok, dw, None


class RowNormFn(Function):
    """x / |x| per row (ctclip.py:119-120), f32."""

    @staticmethod
    def forward(ctx, x):
        x2 = _c(x)
        y = torch.empty_like(x2)
        inv = torch.empty(x2.shape[0], dtype=F32, device=x.device)
        hip.rownorm_fwd(x2, None, y, inv, x2.shape[0], x2.shape[1], 0.0)
        ctx.save_for_backward(x2, inv)
        return y

    @staticmethod
    def backward(ctx, dy):
        x2, inv = ctx.saved_tensors
        dx = torch.empty_like(x2)
        hip.rownorm_bwd(_c(dy), x2, inv, dx, x2.shape[0], x2.shape[1])
        return dx


class SimMatrixFn(Function):
    """sim = img @ txt^T * exp(temperature)   (ctclip.py:127), f32."""

    @staticmethod
    def forward(ctx, img, txt, temperature):
        img, txt = _c(img), _c(txt)
        Gi, L = img.shape
        Gt = txt.shape[0]
        sim = sgemm(img, txt, Gi, Gt, L, alpha_dev=temperature.detach(), alpha_exp=True)
        ctx.save_for_backward(img, txt, temperature, sim)
        return sim

    @staticmethod
    def backward(ctx, ds):
        img, txt, temperature, sim = ctx.saved_tensors
        ds = _c(ds)
        Gi, L = img.shape
        Gt = txt.shape[0]
        t = temperature.detach()
        dimg = sgemm(ds, txt, Gi, L, Gt, a_kmajor=True, b_kmajor=False, alpha_dev=t, alpha_exp=True)
        dtxt = sgemm(ds, img, Gt, L, Gi, a_kmajor=False, b_kmajor=False, alpha_dev=t, alpha_exp=True)
        dtemp = torch.zeros((), dtype=F32, device=ds.device)
        hip.dot_accum(ds, sim, dtemp, ds.numel())                      # d/dT (s0 e^T) = sim
        return dimg, dtxt, dtemp


class InfoNCEFn(Function):
    """0.5 * (CE(sim, arange) + CE(sim^T, arange))   (CTClipTrainer.py:164-175)."""

    @staticmethod
    def forward(ctx, sim):
        sim = _c(sim)
        G = sim.shape[0]
        if sim.shape[1] != G:
            raise ValueError("symmetric InfoNCE needs a square similarity matrix")
        loss = torch.empty((), dtype=F32, device=sim.device)
        dsim = torch.empty_like(sim)
        ws = torch.empty(2 * G, dtype=F32, device=sim.device)
        hip.infonce(sim, loss, dsim, G, ws)
        ctx.save_for_backward(dsim)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        (dsim,) = ctx.saved_tensors
        out = torch.empty_like(dsim)
        hip.scale_by_dev(dsim, _c(dloss.to(F32)), out, dsim.numel())
        return out
