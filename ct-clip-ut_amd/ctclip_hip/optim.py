"""Flat-arena parameters + fused global-norm clip + Adam/AdamW (reference CTClipTrainer.py:199-202,
src/utils/optimizer.py:42-54) and bucketed RCCL gradient all-reduce.

Design for MI355X: all trainable parameters of a group live in ONE contiguous f32 buffer (parameters become
views of it), and so do their gradients and the two Adam moments.  The optimiser step is then two kernel
launches over flat memory (sum of squares, then clip+Adam with the clip coefficient computed on device -- no
host sync), and data-parallel gradient averaging is a handful of large RCCL all-reduces over slices of the flat
gradient buffer instead of one NCCL call per tensor bucket copy.
"""
from __future__ import annotations

import math
from typing import Iterable, Optional

import torch
import torch.distributed as dist

from . import ops
from .lib import hip

F32 = torch.float32


class HipAdam(torch.optim.Optimizer):
    """Adam (decoupled=False) / AdamW (decoupled=True) over flat arenas.  Same hyper-parameter names and defaults
    handling as torch.optim.Adam/AdamW; `step(max_grad_norm=...)` additionally fuses clip_grad_norm_."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, decoupled=False):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, decoupled=decoupled)
        super().__init__(params, defaults)
        self._built = False
        self._step = 0
        self._arenas = []       # per group: dict(p=, g=, m=, v=, params=[...])
        self._gnorm_sq = None
        self.last_grad_norm_sq = None

    # -- arena construction (lazy: parameters may be moved to the GPU after the optimiser was created) ------
    def _build(self):
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.requires_grad]
            if not ps:
                self._arenas.append(None)
                continue
            dev = ps[0].device
            if any((not p.is_cuda) or p.dtype != F32 or p.device != dev for p in ps):
                raise RuntimeError("HipAdam needs float32 parameters on one cuda device: the optimiser step runs as "
                                   "HIP kernels and has no CPU fallback")
            offs, total = [], 0
            for p in ps:
                offs.append(total)
                total += (p.numel() + 3) // 4 * 4
            flat_p = torch.zeros(total, dtype=F32, device=dev)
            flat_g = torch.zeros(total, dtype=F32, device=dev)
            with torch.no_grad():
                for p, o in zip(ps, offs):
                    view = flat_p[o:o + p.numel()].view(p.shape)
                    view.copy_(p.data)
                    p.data = view
                    p.grad = flat_g[o:o + p.numel()].view(p.shape)
            self._arenas.append(dict(p=flat_p, g=flat_g, m=torch.zeros_like(flat_p), v=torch.zeros_like(flat_p),
                                     params=ps, offs=offs))
        dev = next(a for a in self._arenas if a is not None)["p"].device
        self._gnorm_sq = torch.zeros((), dtype=F32, device=dev)
        self._built = True
        ops.bump_weight_epoch()

    def flat_grads(self):
        if not self._built:
            self._build()
        return [a["g"] for a in self._arenas if a is not None]

    def zero_grad(self, set_to_none: bool = False):
        if not self._built:
            self._build()
        for a in self._arenas:
            if a is None:
                continue
            a["g"].zero_()
            for p, o in zip(a["params"], a["offs"]):
                if p.grad is None or p.grad.data_ptr() != a["g"].data_ptr() + 4 * o:
                    p.grad = a["g"][o:o + p.numel()].view(p.shape)

    @torch.no_grad()
    def step(self, closure=None, max_grad_norm: Optional[float] = None):
        if closure is not None:
            raise NotImplementedError("closures are not supported")
        if not self._built:
            self._build()
        self._step += 1
        gn = None
        if max_grad_norm:
            self._gnorm_sq.zero_()
            for a in self._arenas:
                if a is not None:
                    hip.sumsq_accum(a["g"], a["g"].numel(), self._gnorm_sq)
            gn = self._gnorm_sq
            self.last_grad_norm_sq = gn
        for group, a in zip(self.param_groups, self._arenas):
            if a is None:
                continue
            b1, b2 = group["betas"]
            hip.adam_step(a["p"], a["g"], a["m"], a["v"], None, a["p"].numel(), float(group["lr"]), float(b1), float(b2),
                          float(group["eps"]), float(group["weight_decay"]), int(bool(group["decoupled"])),
                          1.0 - b1 ** self._step, 1.0 - b2 ** self._step, gn, float(max_grad_norm or 0.0))
        ops.bump_weight_epoch()

    def grad_norm(self) -> float:
        """Global gradient norm seen by the last clipped step (host sync; for logging/tests)."""
        return float(self.last_grad_norm_sq.sqrt()) if self.last_grad_norm_sq is not None else float("nan")

    def state_dict(self):
        sd = super().state_dict()
        sd["hip"] = {"step": self._step,
                     "m": [None if a is None else a["m"].clone() for a in self._arenas],
                     "v": [None if a is None else a["v"].clone() for a in self._arenas]}
        return sd

    def load_state_dict(self, sd):
        extra = sd.pop("hip", None)
        super().load_state_dict(sd)
        if extra is not None:
            if not self._built:
                self._build()
            self._step = extra["step"]
            for a, m, v in zip(self._arenas, extra["m"], extra["v"]):
                if a is not None:
                    a["m"].copy_(m)
                    a["v"].copy_(v)


class GradSync:
    """Data-parallel gradient averaging over RCCL (`backend="nccl"` is RCCL on ROCm): all-reduce(avg) of slices of the
    flat gradient arenas.  Replaces DistributedDataParallel's per-bucket copies (reference CTClipTrainer.py:62-69,109-115).

    With 7 xGMI peers a few large collectives keep every link busy; bucket_mb sizes the slices so the first ones
    (issued as soon as backward ends) overlap with the rest of the host-side step."""

    def __init__(self, optimizer: HipAdam, bucket_mb: int = 128, group=None, overlap: bool = True):
        self.opt, self.group = optimizer, group
        self.bucket_elems = bucket_mb * (1 << 20) // 4
        self._early = []            # (arena index, start, stop, work handle or (handle, chunk)) launched during backward
        if overlap:
            from . import ops
            ops.grad_ready_hook = self.early_reduce

    def _launch(self, chunk):
        w = self.world()
        if chunk.is_cuda and dist.get_backend(self.group) == "nccl":
            return dist.all_reduce(chunk, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
        return (dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True), chunk)

    def early_reduce(self, param):
        """Called from a backward once `param`'s gradient kernels have been issued: start its all-reduce now, on RCCL's
        stream (which first waits for the work already queued on the compute stream), and remember the slice."""
        if self.world() == 1 or not self.opt._built:
            return
        for ai, a in enumerate(self.opt._arenas):
            if a is None:
                continue
            for p, o in zip(a["params"], a["offs"]):
                if p is param:
                    start, stop = o, o + p.numel()
                    self._early.append((ai, start, stop, self._launch(a["g"][start:stop])))
                    return

    def world(self) -> int:
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def all_reduce_grads(self):
        w = self.world()
        if w == 1:
            return
        handles = [e[3] for e in self._early]
        arenas = [a for a in getattr(self.opt, "_arenas", []) if a is not None] if hasattr(self.opt, "_arenas") else None
        grads = self.opt.flat_grads()
        for gi, g in enumerate(grads):
            n = g.numel()
            # slices not already in flight: the gaps around the early-reduced ranges of this arena
            done = sorted((s0, s1) for (ai, s0, s1, _) in self._early
                          if arenas is not None and arenas[gi] is self.opt._arenas[ai])
            pos = 0
            gaps = []
            for s0, s1 in done:
                if s0 > pos:
                    gaps.append((pos, s0))
                pos = max(pos, s1)
            if pos < n:
                gaps.append((pos, n))
            for g0, g1 in gaps:
                for s in range(g0, g1, self.bucket_elems):
                    handles.append(self._launch(g[s:min(g1, s + self.bucket_elems)]))
        self._early = []
        for h in handles:
            if isinstance(h, tuple):
                h[0].wait()
                h[1].div_(w)
            else:
                h.wait()
