"""Flat-arena parameters + fused global-norm clip + Adam/AdamW (reference CTClipTrainer.py:199-202,
src/utils/optimizer.py:42-54) and bucketed RCCL gradient all-reduce.

Design for MI355X: all trainable parameters of a group live in ONE contiguous f32 buffer (parameters become
views of it), and so do their gradients and the two Adam moments.  The optimiser step is then two kernel
launches over flat memory (sum of squares, then clip+Adam with the clip coefficient computed on device -- no
host sync), and data-parallel gradient averaging is a handful of large RCCL all-reduces over slices of the flat
gradient buffer instead of one NCCL call per tensor bucket copy.

Checkpoint format: `state_dict()` / `load_state_dict()` speak torch.optim.Adam's format (per-parameter `step`,
`exp_avg`, `exp_avg_sq`, index-packed `param_groups` with Adam's group keys), so the "optim" entry of a reference
checkpoint (CTClipTrainer.py:136-154) loads here and one written here loads into the reference's Adam/AdamW.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist

from . import ops
from .lib import hip

F32 = torch.float32

# group keys torch.optim.Adam / AdamW carry (torch 2.x); kept in every group so an exported state dict loads into them
_TORCH_ADAM_KEYS = dict(amsgrad=False, maximize=False, foreach=None, capturable=False, differentiable=False, fused=None)


def mark_unused(p: torch.Tensor) -> None:
    """Tag a parameter that never receives a gradient on the CT-CLIP path.  HipAdam keeps it in
    `param_groups` (so parameter indices equal the reference's `get_optimizer(model.parameters())`) but gives it no
    arena slot -- torch's Adam likewise never creates state for a parameter whose grad stays None."""
    p._ctclip_unused = True


class HipAdam(torch.optim.Optimizer):
    """Adam (decoupled_weight_decay=False) / AdamW (True) over flat arenas.  Same hyper-parameter names and group keys
    as torch.optim.Adam/AdamW; `step(max_grad_norm=...)` additionally fuses clip_grad_norm_."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, decoupled=False,
                 amsgrad=False, maximize=False, **torch_adam_flags):
        if amsgrad or maximize:
            raise NotImplementedError("amsgrad / maximize are not implemented by the fused HIP step")
        # torch.optim.Adam's implementation switches: `fused` / `foreach` choose between equivalent implementations (this one
        # is always fused); the two that change semantics cannot be honoured, and anything else is a typo
        unknown = set(torch_adam_flags) - {"fused", "foreach", "capturable", "differentiable"}
        if unknown:
            raise TypeError(f"HipAdam got unexpected keyword arguments {sorted(unknown)}")
        for k in ("capturable", "differentiable"):
            if torch_adam_flags.get(k):
                raise NotImplementedError(f"{k}=True is not supported by the fused HIP step")
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, **_TORCH_ADAM_KEYS,
                        decoupled_weight_decay=bool(decoupled))
        super().__init__(params, defaults)
        self._built = False
        self._step = 0
        self._arenas = []       # per group: dict(p=, g=, m=, v=, params=[...], offs=[...]) or None
        self._gnorm_sq = None
        self.last_grad_norm_sq = None
        self._grad_listeners = []   # GradSync objects averaging this optimiser's arenas (joined / re-armed by zero_grad)

    # -- arena construction (lazy: parameters may be moved to the GPU after the optimiser was created) ------
    def _build(self):
        self._arenas = []
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.requires_grad and not getattr(p, "_ctclip_unused", False)]
            if not ps:
                self._arenas.append(None)
                continue
            dev = ps[0].device
            if any(p.dtype != F32 or p.device != dev for p in ps):
                raise RuntimeError("HipAdam needs float32 parameters on one device")
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
                    slot = flat_g[o:o + p.numel()].view(p.shape)
                    if p.grad is not None:                   # a gradient that was there before the arena existed
                        slot.copy_(p.grad)
                    p.grad = slot
            self._arenas.append(dict(p=flat_p, g=flat_g, m=torch.zeros_like(flat_p), v=torch.zeros_like(flat_p),
                                     params=ps, offs=offs))
        live = [a for a in self._arenas if a is not None]
        if not live:
            raise RuntimeError("HipAdam: no trainable parameters")
        self._gnorm_sq = torch.zeros((), dtype=F32, device=live[0]["p"].device)
        self._built = True
        ops.bump_weight_epoch()

    def flat_grads(self):
        if not self._built:
            self._build()
        return [a["g"] for a in self._arenas if a is not None]

    def _bind_grads(self, adopt: bool):
        """Make every parameter's .grad the arena view again.  adopt: a gradient somebody assigned as a new tensor
        (`p.grad = t`, the torch.optim contract) is copied into its slot first; a grad of None counts as zero.

        Difference from torch.optim.Adam, by design of the flat arena: a parameter that HAS a slot is stepped every step with
        the one global step count -- a zero gradient still decays its moments (and applies weight decay) -- where torch
        skips a parameter whose grad is None and keeps a per-parameter step.  Parameters that never receive a gradient on
        this path are the ones tagged mark_unused(): they have no slot, no state and are never touched, like torch's."""
        for a in self._arenas:
            if a is None:
                continue
            base = a["g"].data_ptr()
            for p, o in zip(a["params"], a["offs"]):
                g = p.grad
                if g is not None and g.data_ptr() == base + 4 * o and g.dtype == F32:
                    continue
                slot = a["g"][o:o + p.numel()].view(p.shape)
                if adopt:
                    if g is None:
                        slot.zero_()
                    else:
                        slot.copy_(g)
                p.grad = slot

    def zero_grad(self, set_to_none: bool = False):
        if not self._built:
            self._build()
        for sync in self._grad_listeners:
            sync.reset()
        ops.join_side_streams()                              # a backward may still be accumulating on the text stream
        for a in self._arenas:
            if a is not None:
                a["g"].zero_()
        self._bind_grads(adopt=False)

    @torch.no_grad()
    def step(self, closure=None, max_grad_norm: Optional[float] = None):
        if closure is not None:
            raise NotImplementedError("closures are not supported")
        if not self._built:
            self._build()
        self._bind_grads(adopt=True)
        ops.join_side_streams()                              # gradients of the text tower were accumulated on its own stream
        if not self._arenas or not next(a for a in self._arenas if a is not None)["p"].is_cuda:
            raise RuntimeError("HipAdam.step: parameters are not on a HIP device; the optimiser step runs as HIP kernels "
                               "and has no CPU fallback")
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
                          float(group["eps"]), float(group["weight_decay"]), int(bool(group["decoupled_weight_decay"])),
                          1.0 - b1 ** self._step, 1.0 - b2 ** self._step, gn, float(max_grad_norm or 0.0))
        ops.bump_weight_epoch()

    def grad_norm(self) -> float:
        """Global gradient norm seen by the last clipped step (host sync; for logging/tests)."""
        return float(self.last_grad_norm_sq.sqrt()) if self.last_grad_norm_sq is not None else float("nan")

    # -- torch.optim.Adam-format checkpoints -------------------------------------------------------------------
    def _index_of(self):
        idx, out = 0, {}
        for group in self.param_groups:
            for p in group["params"]:
                out[id(p)] = idx
                idx += 1
        return out

    def state_dict(self):
        if not self._built:
            self._build()
        index = self._index_of()
        groups, start = [], 0
        for group in self.param_groups:
            packed = {k: v for k, v in group.items() if k != "params"}
            packed["params"] = list(range(start, start + len(group["params"])))
            start += len(group["params"])
            groups.append(packed)
        state = {}
        if self._step > 0:
            for a in self._arenas:
                if a is None:
                    continue
                for p, o in zip(a["params"], a["offs"]):
                    n = p.numel()
                    state[index[id(p)]] = {"step": torch.tensor(float(self._step)),
                                           "exp_avg": a["m"][o:o + n].view(p.shape).clone(),
                                           "exp_avg_sq": a["v"][o:o + n].view(p.shape).clone()}
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        """Accepts a torch.optim.Adam/AdamW state dict (what the reference's save_model writes) or one of ours (same
        format).  The caller's dict is not modified."""
        if not self._built:
            self._build()
        groups = sd["param_groups"]
        if len(groups) != len(self.param_groups):
            raise ValueError("loaded state dict has a different number of parameter groups")
        for mine, theirs in zip(self.param_groups, groups):
            if len(mine["params"]) != len(theirs["params"]):
                raise ValueError("loaded state dict contains a parameter group that doesn't match the size of "
                                 "optimizer's group")
            if theirs.get("amsgrad") or theirs.get("maximize"):
                raise NotImplementedError("amsgrad / maximize state cannot be continued by the fused HIP step")
            for k in ("lr", "betas", "eps", "weight_decay", "decoupled_weight_decay"):
                if k in theirs:
                    mine[k] = theirs[k]
        # saved index -> my parameter, group by group in order (torch's own mapping rule)
        by_saved = {}
        for mine, theirs in zip(self.param_groups, groups):
            for p, i in zip(mine["params"], theirs["params"]):
                by_saved[i] = p
        slot = {}
        for a in self._arenas:
            if a is not None:
                for p, o in zip(a["params"], a["offs"]):
                    slot[id(p)] = (a, o)
        steps = set()
        for a in self._arenas:
            if a is not None:
                a["m"].zero_()
                a["v"].zero_()
        for i, st in sd["state"].items():
            p = by_saved.get(int(i))
            if p is None:
                raise ValueError(f"state for unknown parameter index {i}")
            if id(p) not in slot:
                continue                                   # state of a parameter this path never updates
            a, o = slot[id(p)]
            n = p.numel()
            if tuple(st["exp_avg"].shape) != tuple(p.shape):
                raise ValueError(f"exp_avg of parameter {i} has shape {tuple(st['exp_avg'].shape)}, expected {tuple(p.shape)}")
            a["m"][o:o + n].view(p.shape).copy_(st["exp_avg"])
            a["v"][o:o + n].view(p.shape).copy_(st["exp_avg_sq"])
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError(f"parameters carry different step counts {sorted(steps)}; the fused step keeps one")
        self._step = steps.pop() if steps else 0


class GradSync:
    """Data-parallel gradient averaging over RCCL (`backend="nccl"` is RCCL on ROCm): all-reduce(avg) of contiguous slices
    ("buckets") of the flat gradient arenas.  Replaces DistributedDataParallel's per-bucket copies (reference
    CTClipTrainer.py:62-69,109-115; fired from :196).

    Overlap with backward: every parameter of THIS optimiser is tagged with `_ctclip_sync = self` (a per-optimiser
    registration: several trainers may live in one process) and reports when its gradient has been ISSUED into the arena
    -- through ops.grad_slot()/announce_grads() from the Functions that accumulate in place, through a post-accumulate hook
    for gradients autograd delivers.  A bucket whose parameters have all reported is all-reduced at once, so buckets leave
    in reverse-autograd order while the rest of backward runs.  Whatever has not reported when backward returns is reduced
    by all_reduce_grads().  With 7 xGMI peers a few large collectives keep every link busy; `bucket_mb` sizes them.

    More than one backward per step (gradient accumulation, an attribution backward through the model, a step aborted
    between backward and all_reduce_grads):
      * `with sync.no_sync():` -- reports are ignored, nothing is launched: the micro-batches before the last one;
      * a parameter that is about to be written again after its bucket has left (`before_write`, called by grad_slot() and a
        tensor hook BEFORE the gradient kernels are issued) first makes the current stream wait for that collective and
        re-arms the bucket: the slice then holds avg(g1), the new local gradient is added on top, and the second
        all-reduce(avg) yields avg(g1) + avg(g2) -- no write into a slice RCCL is still reducing, no element reduced twice;
      * HipAdam.zero_grad() joins whatever is still in flight and re-arms every bucket before it clears the arena."""

    def __init__(self, optimizer: HipAdam, bucket_mb: int = 64, group=None, overlap: bool = True):
        self.opt, self.group = optimizer, group
        self.bucket_elems = max(1, bucket_mb * (1 << 20) // 4)
        self.overlap = overlap
        self._buckets = None        # list of dict(ai, start, stop, n, pending, handle)
        self._bucket_of = {}
        self._hooks = []
        self._tagged = []
        self._defer = False
        listeners = getattr(optimizer, "_grad_listeners", None)
        if listeners is not None:
            listeners.append(self)

    def close(self):
        self.reset()
        for h in self._hooks:
            h.remove()
        self._hooks = []
        for p in self._tagged:
            if getattr(p, "_ctclip_sync", None) is self:
                del p._ctclip_sync
        self._tagged = []
        self._buckets, self._bucket_of = None, {}
        listeners = getattr(self.opt, "_grad_listeners", None)
        if listeners is not None and self in listeners:
            listeners.remove(self)

    def world(self) -> int:
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def active(self) -> bool:
        """Collectives run whenever a process group exists -- also at world size 1, where they are no-ops on the data but
        the RCCL path is the one exercised."""
        return dist.is_available() and dist.is_initialized()

    def no_sync(self):
        """Context manager for the micro-batches of a gradient-accumulation step that must not start collectives (DDP's
        `no_sync`): gradients only accumulate locally; the backward of the last micro-batch, outside of it, reduces."""
        sync = self

        class _NoSync:
            def __enter__(self_):
                self_.was, sync._defer = sync._defer, True

            def __exit__(self_, *exc):
                sync._defer = self_.was
                return False
        return _NoSync()

    # -- plan ---------------------------------------------------------------------------------------------------
    def _plan(self):
        arenas = getattr(self.opt, "_arenas", None)
        self._buckets, self._bucket_of = [], {}
        if arenas is None:                                   # arena-less stand-in (CPU tests): fixed-size slices
            for ai, g in enumerate(self.opt.flat_grads()):
                for s in range(0, g.numel(), self.bucket_elems):
                    self._buckets.append(dict(ai=ai, start=s, stop=min(g.numel(), s + self.bucket_elems), n=0,
                                              pending=0, handle=None))
            return
        if not self.opt._built:
            self.opt._build()
            arenas = self.opt._arenas                        # _build() replaces the list
        for ai, a in enumerate(arenas):
            if a is None:
                continue
            cur = None
            total = a["g"].numel()
            for i, (p, o) in enumerate(zip(a["params"], a["offs"])):
                if cur is not None and p.numel() >= self.bucket_elems:
                    cur = None                               # a tensor that fills a bucket by itself travels alone: it
                                                             # must not wait for small neighbours that finish much later
                if cur is None:
                    cur = dict(ai=ai, start=o, stop=o, n=0, pending=0, handle=None)
                    self._buckets.append(cur)
                cur["n"] += 1
                cur["stop"] = a["offs"][i + 1] if i + 1 < len(a["offs"]) else total
                self._bucket_of[id(p)] = cur
                if cur["stop"] - cur["start"] >= self.bucket_elems:
                    cur = None
                if self.overlap:
                    p._ctclip_sync = self
                    self._tagged.append(p)
                    if hasattr(p, "register_post_accumulate_grad_hook"):
                        # autograd-delivered gradients: told before the accumulation (tensor hook) and after it.  The
                        # engine also runs these hooks with an undefined gradient for a parameter whose Function returned
                        # None (the in-place accumulating ones): nothing is written then
                        self._hooks.append(p.register_hook(
                            lambda g, p=p: self.before_write(p) if g is not None else None))
                        self._hooks.append(p.register_post_accumulate_grad_hook(self.param_ready))
        for b in self._buckets:
            b["pending"] = b["n"]

    def prepare(self):
        """Plan the buckets and tag the parameters now (otherwise done by the first report / all_reduce_grads)."""
        if self._buckets is None and self.active():
            self._plan()

    def _flat(self, b):
        arenas = getattr(self.opt, "_arenas", None)
        g = arenas[b["ai"]]["g"] if arenas is not None else self.opt.flat_grads()[b["ai"]]
        return g[b["start"]:b["stop"]]

    # -- launch / join -------------------------------------------------------------------------------------------
    def _launch(self, b):
        chunk = self._flat(b)
        # the bucket may hold gradients written on the text stream AND on the main one: the stream that starts the collective waits
        # for the other first -- once per bucket, whatever the backend (a no-op when no second stream exists, e.g. on CPU)
        ops.join_side_streams()
        if chunk.is_cuda and dist.get_backend(self.group) == "nccl":
            b["handle"] = dist.all_reduce(chunk, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
        else:                                                # gloo: no AVG
            b["handle"] = (dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True), chunk)

    def _join(self, b):
        """Wait for b's collective (the current stream then orders after RCCL's) and re-arm the bucket."""
        h = b["handle"]
        if h is not None:
            if isinstance(h, tuple):
                h[0].wait()
                h[1].div_(self.world())
            else:
                h.wait()
        b["handle"] = None
        b["pending"] = b["n"]
        b["seen"] = set()

    def before_write(self, param):
        """`param`'s arena slot is about to be written by gradient kernels that have NOT been issued yet."""
        if self._buckets is None:
            return
        b = self._bucket_of.get(id(param))
        if b is not None and b["handle"] is not None:
            self._join(b)                                    # a second backward in this step: see the class docstring

    def param_ready(self, param):
        """`param`'s gradient kernels have been issued."""
        if self._defer or not self.active():
            return
        if self._buckets is None:
            self._plan()
        b = self._bucket_of.get(id(param))
        if b is None or b["handle"] is not None:
            return
        seen = b.setdefault("seen", set())
        if id(param) in seen:                                # the engine also runs the post-accumulate hook of a parameter
            return                                           # whose Function accumulated in place and returned None
        seen.add(id(param))
        b["pending"] -= 1
        if b["pending"] <= 0:
            self._launch(b)

    # kept for callers of the round-1 name
    early_reduce = param_ready

    def reset(self):
        """Join every collective still in flight and re-arm all buckets (HipAdam.zero_grad calls this before it clears the
        arena: a step that was aborted after its backward must not leave a reduce running over memory being zeroed)."""
        if self._buckets:
            for b in self._buckets:
                self._join(b)

    def all_reduce_grads(self):
        """Reduce what is left, wait for everything (the current stream then orders after RCCL's), reset for the next step."""
        if not self.active():
            return
        if self._buckets is None:
            self._plan()
        for b in self._buckets:
            if b["handle"] is None:
                self._launch(b)
        for b in self._buckets:
            self._join(b)
