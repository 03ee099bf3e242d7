"""world_size-2 `gloo` tests (CPU) of the data-parallel host logic: the gradient arena + bucketed all-reduce,
GatherWithGrad's forward/backward semantics (reference src/models/ctclip.py:10-41) and the gradient-scaling consequence
of SURVEY.md 8(e): after averaging, encoder gradients equal (1/W) * d(global loss) while `temperature` sees the
un-scaled gradient.  No HIP kernels run here; compute pieces are stood in by plain torch expressions of the same math
(the point is the collective wiring, which is identical on RCCL)."""
import os
import sys
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

WORLD = 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(fn, *args):
    port = _free_port()
    mp.spawn(_entry, args=(fn, port, args), nprocs=WORLD, join=True)


def _entry(rank, fn, port, args):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "ct-clip-ut_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(WORLD))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        fn(rank, *args)
    finally:
        dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------------------------
def _gather_semantics(rank):
    from models.ctclip import GatherWithGrad
    x = (torch.arange(6, dtype=torch.float32).reshape(3, 2) + 100 * rank).requires_grad_(True)
    g = GatherWithGrad.apply(x)
    assert g.shape == (3 * WORLD, 2)
    expect = torch.cat([torch.arange(6, dtype=torch.float32).reshape(3, 2) + 100 * r for r in range(WORLD)])
    assert torch.equal(g.detach(), expect)                                  # rank-major concatenation (ctclip.py:25)
    w = torch.arange(12, dtype=torch.float32).reshape(6, 2) + 1
    (g * w).sum().backward()
    assert torch.equal(x.grad, w[3 * rank:3 * rank + 3])                    # local slice, NO reduction (ctclip.py:38-39)


def test_gather_with_grad_forward_and_local_slice_backward():
    _run(_gather_semantics)


# ---------------------------------------------------------------------------------------------------------------------
class _FakeArenaOpt:
    """GradSync only needs .flat_grads(); the real HipAdam arena is cuda-only."""

    def __init__(self, grads):
        self._g = grads

    def flat_grads(self):
        return self._g


def _bucketed_allreduce(rank):
    from ctclip_hip.optim import GradSync
    torch.manual_seed(rank)
    g1, g2 = torch.randn(1000), torch.randn(37)
    want1 = [None, None]
    all1 = [torch.zeros(1000) for _ in range(WORLD)]
    all2 = [torch.zeros(37) for _ in range(WORLD)]
    dist.all_gather(all1, g1)
    dist.all_gather(all2, g2)
    sync = GradSync(_FakeArenaOpt([g1, g2]), bucket_mb=1)
    sync.bucket_elems = 256                                                  # force several buckets + a ragged tail
    sync.all_reduce_grads()
    torch.testing.assert_close(g1, sum(all1) / WORLD)
    torch.testing.assert_close(g2, sum(all2) / WORLD)


def test_gradsync_bucketed_average():
    _run(_bucketed_allreduce)


def _hook_driven_buckets(rank):
    """The real HipAdam arena (plumbing only, on CPU) + GradSync: parameters report as backward produces their gradients
    -- autograd-delivered ones through the post-accumulate hook, in-place accumulating Functions through
    ops.grad_slot()/announce_grads() -- and a bucket is all-reduced the moment its last parameter has reported, i.e.
    in reverse-autograd order while backward is still running.  A tensor as large as a bucket travels alone.  Every
    element is averaged exactly once."""
    from ctclip_hip import ops
    from ctclip_hip.optim import GradSync, HipAdam, mark_unused
    torch.manual_seed(0)                                                       # same weights on both ranks
    lin1, lin2, big = torch.nn.Linear(6, 5), torch.nn.Linear(5, 4), torch.nn.Linear(4, 64, bias=False)
    unused = torch.nn.Parameter(torch.zeros(3))
    mark_unused(unused)
    direct = torch.nn.Parameter(torch.zeros(8))                                # gradient written in place by a "kernel"
    params = [*lin1.parameters(), unused, *lin2.parameters(), *big.parameters(), direct]
    # reference first (no hooks installed yet): the plain local gradients, to be averaged with a bare all-reduce
    torch.manual_seed(10 + rank)                                               # different data per rank
    x = torch.randn(3, 6)
    y2 = big(lin2(torch.relu(lin1(x))))
    (y2 ** 2).sum().backward()
    want = {id(pr): pr.grad.clone() for pr in params if pr.grad is not None}
    want[id(direct)] = (2 * y2.detach()).sum() * torch.arange(8.0) * (rank + 1)
    for pr in params:
        pr.grad = None
    opt = HipAdam(params, lr=1e-3)
    sync = GradSync(opt, bucket_mb=1)
    sync.bucket_elems = 40                                                     # lin1 (36) + lin2.weight | lin2.bias.. | big alone
    opt.zero_grad()
    sync._plan()
    ranges = [(b["start"], b["stop"], b["n"]) for b in sync._buckets]
    big_off = next(o for pr, o in zip(opt._arenas[0]["params"], opt._arenas[0]["offs"]) if pr is big.weight)
    assert (big_off, big_off + 256, 1) in ranges, ranges                      # the 256-element tensor has its own bucket
    assert id(unused) not in sync._bucket_of
    launched = []
    real_launch = sync._launch
    sync._launch = lambda b: (launched.append((b["start"], b["stop"])), real_launch(b))[1]

    class InPlace(torch.autograd.Function):                                    # stands in for a HIP backward
        @staticmethod
        def forward(ctx, x, p):
            ctx.p = p
            return x * 1.0

        @staticmethod
        @ops.announces
        def backward(ctx, dy):
            slot, is_direct = ops.grad_slot(ctx.p)
            assert is_direct
            slot += dy.sum() * torch.arange(8.0) * (rank + 1)
            return dy, None

    y = big(lin2(torch.relu(lin1(x))))
    y = InPlace.apply(y, direct)
    during = []
    y.register_hook(lambda g: during.append(len(launched)))
    (y ** 2).sum().backward()
    n_early = len(launched)
    assert n_early >= 2, launched                                              # buckets left while backward was running
    assert launched[0][0] >= big_off                                          # reverse-autograd order: the tail of the arena first
    sync.all_reduce_grads()
    assert len(launched) == len(sync._buckets)
    a = opt._arenas[0]
    for pr, o in zip(a["params"], a["offs"]):
        w = want[id(pr)].clone()
        dist.all_reduce(w)
        torch.testing.assert_close(a["g"][o:o + pr.numel()].view(pr.shape), w / WORLD, rtol=1e-5, atol=1e-6)
    assert all(b["handle"] is None and b["pending"] == b["n"] for b in sync._buckets)   # armed for the next step
    sync.close()
    assert all(not hasattr(pr, "_ctclip_sync") for pr in params) and not opt._grad_listeners


def test_gradsync_buckets_leave_during_backward_and_average_once():
    _run(_hook_driven_buckets)


# ---------------------------------------------------------------------------------------------------------------------
def _fused_latent_gather(rank):
    """CTCLIP.forward gathers text and image latents with ONE collective over [B, 2L] (reference ctclip.py:123-124 uses two
    all_gathers): values and gradients equal those of the two separate GatherWithGrad calls."""
    from models.ctclip import GatherWithGrad
    torch.manual_seed(5 + rank)
    B, L = 3, 4
    t = torch.randn(B, L, requires_grad=True)
    i = torch.randn(B, L, requires_grad=True)
    t2, i2 = t.detach().clone().requires_grad_(True), i.detach().clone().requires_grad_(True)
    gt, gi = GatherWithGrad.apply(t), GatherWithGrad.apply(i)                  # the reference's two calls
    both = GatherWithGrad.apply(torch.cat((t2, i2), dim=1))                    # the fused one
    ft, fi = both[:, :L], both[:, L:]
    assert torch.equal(ft, gt) and torch.equal(fi, gi)
    w = torch.arange(float(WORLD * B * WORLD * B)).reshape(WORLD * B, WORLD * B)
    ((gi @ gt.t()) * w).sum().backward()
    ((fi @ ft.t()) * w).sum().backward()
    torch.testing.assert_close(t2.grad, t.grad)
    torch.testing.assert_close(i2.grad, i.grad)


def test_fused_latent_gather_equals_two_gathers():
    _run(_fused_latent_gather)


# ---------------------------------------------------------------------------------------------------------------------
def _avg_device_loss(rank):
    """CTClipTrainer.avg_device_loss (reference CTClipTrainer.py:156-162): the mean over ranks, as a python float."""
    from utils.CTClipTrainer import CTClipTrainer

    class Stub:
        class accelerator:
            device = torch.device("cpu")

    got = CTClipTrainer.avg_device_loss(Stub(), 1.0 + 2.0 * rank)
    assert isinstance(got, float) and abs(got - sum(1.0 + 2.0 * r for r in range(WORLD)) / WORLD) < 1e-6
    got = CTClipTrainer.avg_device_loss(Stub(), torch.tensor(3.0 * rank))
    assert abs(got - 1.5 * (WORLD - 1)) < 1e-6


def test_avg_device_loss_is_the_mean_over_ranks():
    _run(_avg_device_loss)


# ---------------------------------------------------------------------------------------------------------------------
def _vq_ema_async(rank):
    """the codebook statistics (bins, embed_sum) are summed over the ranks by ONE asynchronous collective over
    the flat buffer that holds both; flush_ema() joins it and applies the update exactly once (the apply kernel is stood
    in by a recorder here -- the wiring is what is tested)."""
    from ctclip_hip import ops, vq as vqmod
    q = vqmod.VectorQuantize(dim=8, codebook_size=16)
    flat = torch.full((16 * 9,), float(rank + 1))
    bins, esum = flat[:16], flat[16:].view(16, 8)
    q._pending_ema = (dist.all_reduce(flat, async_op=True), bins, esum)
    seen = []
    real = ops.vq_ema_apply
    ops.vq_ema_apply = lambda embed, cs, b, e, decay: seen.append((b.clone(), e.clone(), decay))
    try:
        q.flush_ema()
        q.flush_ema()                                                          # nothing pending: no second apply
        sd = q.state_dict()                                                    # flushes too (nothing pending)
    finally:
        ops.vq_ema_apply = real
    total = float(sum(r + 1 for r in range(WORLD)))
    assert len(seen) == 1 and seen[0][2] == 0.8
    assert torch.all(seen[0][0] == total) and torch.all(seen[0][1] == total)
    assert "_codebook.embed" in sd


def test_vq_codebook_statistics_reduce_in_the_background():
    _run(_vq_ema_async)


# ---------------------------------------------------------------------------------------------------------------------
def _global_contrastive_grad_scaling(rank):
    """Two ranks with B pairs each + all-gather == one process with the 2B batch, up to the reference's scaling quirk."""
    from models.ctclip import GatherWithGrad
    from oracle import ctclip_oracle as O
    B, L = 3, 8
    gen = torch.Generator().manual_seed(0)
    W_img, W_txt = torch.randn(L, 5, generator=gen), torch.randn(L, 4, generator=gen)
    feats_img, feats_txt = torch.randn(WORLD * B, 5, generator=gen), torch.randn(WORLD * B, 4, generator=gen)
    temp0 = torch.tensor(1.0)

    def loss_of(wi, wt, temp, fi, ft, gather):
        il = fi @ wi.t()
        tl = ft @ wt.t()
        il, tl = il / il.norm(dim=-1, keepdim=True), tl / tl.norm(dim=-1, keepdim=True)
        if gather:
            tl, il = GatherWithGrad.apply(tl), GatherWithGrad.apply(il)      # same order as ctclip.py:123-124
        return O.symmetric_info_nce(O.sim_matrix(il, tl, temp))

    # single-process reference on the concatenated batch
    wi, wt, tp = W_img.clone().requires_grad_(True), W_txt.clone().requires_grad_(True), temp0.clone().requires_grad_(True)
    ref = loss_of(wi, wt, tp, feats_img, feats_txt, gather=False)
    ref.backward()
    # this rank's shard
    wi2, wt2, tp2 = W_img.clone().requires_grad_(True), W_txt.clone().requires_grad_(True), temp0.clone().requires_grad_(True)
    sl = slice(rank * B, (rank + 1) * B)
    loss = loss_of(wi2, wt2, tp2, feats_img[sl], feats_txt[sl], gather=True)
    torch.testing.assert_close(loss.detach(), ref.detach())                   # every rank computes the SAME global loss
    loss.backward()
    for g in (wi2.grad, wt2.grad, tp2.grad):                                  # gradient averaging as GradSync does it
        dist.all_reduce(g)
        g /= WORLD
    torch.testing.assert_close(wi2.grad, wi.grad / WORLD)                     # encoders: (1/W) * d(global loss)
    torch.testing.assert_close(wt2.grad, wt.grad / WORLD)
    torch.testing.assert_close(tp2.grad, tp.grad)                             # temperature: un-scaled


def test_global_negatives_gradient_scaling_matches_reference_semantics():
    _run(_global_contrastive_grad_scaling)


# ---------------------------------------------------------------------------------------------------------------------
def _sharded_weak_scaling_units(rank):
    """bench.py's multi-rank accounting: ranks own disjoint synthetic pairs (seed 1234+rank) and report MAX time."""
    import bench
    vol_a, txt_a = bench.synthetic_batch(2, 8, 8, 16, 50, torch.device("cpu"), rank, dtype=torch.float32)
    vol_b, _ = bench.synthetic_batch(2, 8, 8, 16, 50, torch.device("cpu"), 1 - rank, dtype=torch.float32)
    assert not torch.equal(vol_a, vol_b)
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t) == float(WORLD)


def test_bench_shards_units_and_takes_max_time():
    _run(_sharded_weak_scaling_units)


def test_bench_self_launches_its_ranks_from_a_plain_shell():
    """`python bench.py --gpus 2` with no RANK / WORLD_SIZE in the environment (how the driver starts the scaling run): the
    parent starts two ranks through torch.distributed.run on 127.0.0.1, both rendezvous (gloo here) before anything touches a
    device, and the parent returns the children's exit code -- non-zero in this GPU-less container, where the training step
    refuses to run ("no HIP device visible": there is no CPU fallback).  On a GPU box the same command prints the JSON line:
    tests/test_hip_trainer.py::test_bench_two_gloo_ranks_share_the_gpu asserts its n_gpus / global_batch / negatives there.
    The parent counts devices from sysfs (bench.visible_gpus), never through the HIP runtime."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["CTCLIP_DIST_BACKEND"] = "gloo"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--small", "--steps", "1", "--warmup", "0",
                        "--batch", "2", "--lean"], env=env, capture_output=True, text=True, timeout=600)
    err = r.stderr
    assert "[bench] self-launch:" in err and "--nproc-per-node=2" in err, err[-2000:]
    assert "rank 0 of 2: process group up (gloo)" in err and "rank 1 of 2: process group up (gloo)" in err, err[-2000:]
    if not torch.cuda.is_available():
        assert r.returncode != 0 and "no HIP device visible" in err, err[-2000:]
        assert r.stdout.strip() == ""
    else:
        import json
        line = json.loads(r.stdout.strip().splitlines()[-1])
        assert r.returncode == 0 and line["n_gpus"] == 2


def test_self_launch_parent_never_touches_the_device_runtime(monkeypatch):
    """bench.self_launch (the parent of `python bench.py --gpus N`) must start its ranks from a process that has not initialised
    HIP: it counts devices from sysfs (bench.visible_gpus) and never calls torch.cuda.*; the visibility variables cut the
    count.  Checked by making every torch.cuda entry point it could reach raise."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench
    import subprocess

    def boom(*a, **k):
        raise AssertionError("self_launch touched torch.cuda")
    for name in ("device_count", "is_available", "init", "current_device", "set_device"):
        monkeypatch.setattr(torch.cuda, name, boom)
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env

        class R:
            returncode = 0
        return R()
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(bench, "visible_gpus", lambda: 8)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2"])
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        monkeypatch.delenv(k, raising=False)
    assert bench.self_launch(4) == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "2"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    monkeypatch.setattr(bench, "visible_gpus", lambda: 2)                  # fewer devices than ranks: refused, nothing started
    seen.clear()
    assert bench.self_launch(4) == 2 and not seen
    monkeypatch.undo()
    # the sysfs reader itself: the visibility variables cut the count, a box without KFD says None
    import glob as _glob
    nodes = _glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    n = bench.visible_gpus()
    assert (n is None) == (not nodes)


# ---------------------------------------------------------------------------------------------------------------------
def _join_called_once_per_bucket(rank):
    """GradSync._launch: the stream that starts a bucket's collective first waits for the other one (ops.join_side_streams --
    the text tower's gradients are written on a second HIP stream).  On CPU there are no streams and the call is a no-op, but
    it must be MADE exactly once per bucket launch, hook-driven launches and the tail of all_reduce_grads() alike, so that the
    first multi-GPU run cannot reduce a bucket that still has text-tower kernels in flight."""
    from ctclip_hip import ops
    from ctclip_hip.optim import GradSync
    calls = []
    real = ops.join_side_streams

    def counted():
        calls.append(1)
        return real()
    ops.join_side_streams = counted
    try:
        torch.manual_seed(rank)
        grads = [torch.randn(1000), torch.randn(37)]
        sync = GradSync(_FakeArenaOpt(grads), bucket_mb=1)
        sync.bucket_elems = 256                                                # several buckets + a ragged tail
        sync.all_reduce_grads()
        nb = len(sync._buckets)
        assert nb >= 2 and len(calls) == nb, (nb, len(calls))
        calls.clear()
        sync.all_reduce_grads()                                                # second step: once per bucket again
        assert len(calls) == nb
    finally:
        ops.join_side_streams = real


def test_gradsync_joins_the_text_stream_once_per_bucket():
    _run(_join_called_once_per_bucket)


# ---------------------------------------------------------------------------------------------------------------------
def _two_backwards_per_step(rank):
    """More than one backward between zero_grad() and step() (gradient accumulation, an attribution backward through the
    model): a bucket that has already left is joined and re-armed BEFORE the second backward writes into its slice
    (GradSync.before_write), so the final arena holds avg(g1 + g2) -- nothing is reduced twice, nothing is left local --
    and `no_sync()` keeps the first micro-batch from starting collectives at all.  zero_grad() joins what an aborted step
    left in flight."""
    from ctclip_hip import ops
    from ctclip_hip.optim import GradSync, HipAdam
    torch.manual_seed(0)                                                       # same weights on both ranks
    lin1, lin2 = torch.nn.Linear(6, 5), torch.nn.Linear(5, 64, bias=False)
    direct = torch.nn.Parameter(torch.zeros(8))
    params = [*lin1.parameters(), *lin2.parameters(), direct]
    torch.manual_seed(20 + rank)                                               # different data per rank
    xs = [torch.randn(3, 6), torch.randn(3, 6)]

    class InPlace(torch.autograd.Function):                                    # stands in for a HIP backward
        @staticmethod
        def forward(ctx, x, p):
            ctx.p = p
            return x * 1.0

        @staticmethod
        @ops.announces
        def backward(ctx, dy):
            slot, is_direct = ops.grad_slot(ctx.p)
            assert is_direct
            slot += dy.sum() * torch.arange(8.0) * (rank + 1)
            return dy, None

    def loss_of(x):
        return (InPlace.apply(lin2(torch.relu(lin1(x))), direct) ** 2).sum()

    opt = HipAdam(params, lr=1e-3)
    sync = GradSync(opt, bucket_mb=1)
    sync.bucket_elems = 40
    opt.zero_grad()
    sync.prepare()
    assert all(getattr(pr, "_ctclip_sync", None) is sync for pr in params)
    a = opt._arenas[0]

    def expected():                                                            # local sum of both micro-batches, then averaged
        loc = torch.zeros_like(a["g"])
        for x in xs:
            for pr in params:
                pr.grad = None
            direct.grad = torch.zeros(8)                                       # the in-place Function needs a slot to add into
            loss_of(x).backward()
            for pr, o in zip(a["params"], a["offs"]):
                if pr is direct:                                               # added in closed form below
                    continue
                loc[o:o + pr.numel()] += pr.grad.reshape(-1)
        return loc

    launched = []
    real_launch = sync._launch
    sync._launch = lambda b: (launched.append((b["start"], b["stop"])), real_launch(b))[1]

    # (a) two plain backwards: buckets leave during the first, are joined + re-armed by the second, leave again
    opt.zero_grad()
    loss_of(xs[0]).backward()
    n1 = len(launched)
    assert n1 >= 1 and any(b["handle"] is not None for b in sync._buckets)
    loss_of(xs[1]).backward()
    assert len(launched) > n1                                                  # re-armed buckets were reduced again
    sync.all_reduce_grads()
    got_a = a["g"].clone()
    assert all(b["handle"] is None and b["pending"] == b["n"] for b in sync._buckets)

    # (b) the same step with no_sync() around the first micro-batch: every bucket leaves exactly once
    opt.zero_grad()
    launched.clear()
    with sync.no_sync():
        loss_of(xs[0]).backward()
    assert not launched
    loss_of(xs[1]).backward()
    sync.all_reduce_grads()
    assert len(launched) == len(sync._buckets)
    got_b = a["g"].clone()

    # (c) a step aborted after backward: zero_grad() joins the collectives in flight before it clears the arena
    opt.zero_grad()
    loss_of(xs[0]).backward()
    assert any(b["handle"] is not None for b in sync._buckets)
    opt.zero_grad()
    assert all(b["handle"] is None and b["pending"] == b["n"] for b in sync._buckets)
    assert float(a["g"].abs().max()) == 0.0

    # expected: plain local sums (no hooks), one bare all-reduce
    sync.close()
    assert all(not hasattr(pr, "_ctclip_sync") for pr in params)
    want = expected()
    o_direct = next(o for pr, o in zip(a["params"], a["offs"]) if pr is direct)
    with torch.no_grad():
        for x in xs:
            want[o_direct:o_direct + 8] += (2 * lin2(torch.relu(lin1(x)))).sum() * torch.arange(8.0) * (rank + 1)
    dist.all_reduce(want)
    want /= WORLD
    torch.testing.assert_close(got_a, want, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(got_b, want, rtol=1e-5, atol=1e-6)


def test_gradsync_two_backwards_per_step_and_no_sync():
    _run(_two_backwards_per_step)


# ---------------------------------------------------------------------------------------------------------------------
def _occlusion_window_sharding(rank):
    """utils.visualizations.Visualizations._compute_occlusion (reference src/utils/visualizations.py:340-361,404-409): the
    window list is cut into equal contiguous slices (windows beyond a multiple of the world size are DROPPED, as the
    reference does), every rank scores its own slice, heat-map and count-map are summed onto rank 0 with one reduce each,
    and only rank 0 returns a map.  The scorer is a linear stand-in (sim = <w, volume>), so the importance of a window is
    known in closed form; the batched device-side accumulation and the collectives are the code under test."""
    import numpy as np
    import torch.nn.functional as F
    from utils.visualizations import Visualizations

    D, H, W = 8, 12, 12
    gen = torch.Generator().manual_seed(3)
    image = torch.randn(1, 1, D, H, W, generator=gen).clamp(-1, 1)
    weight = torch.rand(D, H, W, generator=gen)                               # >= 0: occluding (-> -1) never raises the score
    patch, stride = (4, 6, 6), (2, 3, 3)                                       # 3 x 3 x 3 = 27 windows: one is dropped at W = 2

    class Scorer(torch.nn.Module):
        gather_negatives = True

        def encode_text(self, tokens):
            return torch.zeros(1, 4)

        def forward(self, text, vol, cls):
            return ((vol[:, 0] * weight).sum(dim=(1, 2, 3)).reshape(-1, 1),)

    class Acc:
        is_main_process, process_index, num_processes, device = rank == 0, rank, WORLD, torch.device("cpu")

    model = Scorer()
    vis = Visualizations(model, Acc(), occlusion_batch=4)                      # 13 windows per rank: 3 full batches + 1
    got = vis._compute_occlusion(image, {"input_ids": torch.zeros(1, 2, dtype=torch.long)}, None, patch, stride, 0.0)
    assert model.gather_negatives is True                                      # restored
    if rank != 0:
        assert got is None
        return
    coords = [(d, h, w) for d in range(0, D - patch[0] + 1, stride[0]) for h in range(0, H - patch[1] + 1, stride[1])
              for w in range(0, W - patch[2] + 1, stride[2])]
    assert len(coords) == 27
    kept = coords[:(len(coords) // WORLD) * WORLD]                             # reference :352-356
    heat, count = torch.zeros(D, H, W), torch.zeros(D, H, W)
    vol = image[0, 0]
    for d, h, w in kept:
        sl = (slice(d, d + patch[0]), slice(h, h + patch[1]), slice(w, w + patch[2]))
        imp = float(((vol[sl] + 1) * weight[sl]).sum().clamp_min(0))          # original - occluded for the linear scorer
        heat[sl] += imp
        count[sl] += 1
    count[count == 0] = 1
    heat = heat / count
    heat = (heat - heat.min()) / (heat.max() - heat.min() + 1e-8)
    want = np.rot90(F.interpolate(heat[None, None], size=(D, H, W), mode="trilinear", align_corners=False)[0, 0].numpy(),
                    k=-1, axes=(1, 2))
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 1e-5
    assert (count == 1).any() and float(count.max()) > 1                       # overlapping windows were exercised


def test_occlusion_windows_are_sharded_and_reduced_to_rank_zero():
    _run(_occlusion_window_sharding)
