"""world_size-2 `gloo` tests (CPU) of the data-parallel host logic: the gradient arena + bucketed all-reduce,
GatherWithGrad's forward/backward semantics (reference src/models/ctclip.py:10-41) and the gradient-scaling consequence
of SURVEY.md 8(e): after averaging, encoder gradients equal (1/W) * d(global loss) while `temperature` sees the
un-scaled gradient.  No HIP kernels run here; compute pieces are stood in by plain torch expressions of the same math
(the point is the collective wiring, which is identical on RCCL)."""
import os
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


def _early_reduce_then_rest(rank):
    """A parameter whose gradient is announced during backward is all-reduced early; all_reduce_grads() then covers exactly
    the remaining gaps of the arena, so every element is averaged exactly once."""
    from ctclip_hip.optim import GradSync

    class Arena:
        _built = True

        def __init__(self):
            torch.manual_seed(100 + rank)
            self.params = [torch.nn.Parameter(torch.zeros(n)) for n in (40, 300, 17)]
            self.offs = [0, 40, 340]
            self.g = torch.randn(360)
            self._arenas = [None, dict(g=self.g, params=self.params, offs=self.offs)]

        def flat_grads(self):
            return [self.g]

    ar = Arena()
    everyone = [torch.zeros(360) for _ in range(WORLD)]
    dist.all_gather(everyone, ar.g.clone())
    sync = GradSync(ar, overlap=False)
    sync.bucket_elems = 64
    sync.early_reduce(ar.params[1])                                            # the middle slice goes first
    assert len(sync._early) == 1 and sync._early[0][1:3] == (40, 340)
    sync.all_reduce_grads()
    torch.testing.assert_close(ar.g, sum(everyone) / WORLD)
    assert sync._early == []


def test_gradsync_early_slice_plus_gaps_average_once():
    _run(_early_reduce_then_rest)


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
