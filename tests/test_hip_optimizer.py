"""GPU parity of the fused clip + Adam / AdamW step (`ctclip_sumsq_accum`, `ctclip_adam_step` through HipAdam; reference
src/utils/optimizer.py:42-54, src/utils/CTClipTrainer.py:199-202):

  * against tests/golden/optimizer.npz -- three steps of the reference's own `get_optimizer` (Adam for wd == 0, AdamW with
    the ndim >= 2 / < 2 weight-decay groups otherwise) on fixed gradients, 1e-5;
  * with the global-norm clip, against torch.nn.utils.clip_grad_norm_ + torch.optim.Adam/AdamW on the CPU (the calls the
    reference makes), over several tensors of sizes that are not multiples of the arena's 4-element alignment;
  * continuing a torch.optim.Adam checkpoint (the reference's `optim` state dict) for a third step.
"""
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda"


def close(a, b, rtol=1e-5, atol=1e-7):
    torch.testing.assert_close(a.detach().cpu(), b.detach().cpu(), rtol=rtol, atol=atol)


@pytest.mark.parametrize("tag,wd", [("adam", 0.0), ("adamw", 1e-2)])
def test_hip_adam_replays_the_reference_factory_trajectories(tag, wd):
    from utils.optimizer import get_optimizer
    g = load_golden("optimizer")
    pw = torch.nn.Parameter(g[f"{tag}.w0"].clone().to(DEV))
    pb = torch.nn.Parameter(g[f"{tag}.b0"].clone().to(DEV))
    opt = get_optimizer([pw, pb], lr=1e-2, wd=wd)                       # the call the golden script made on the reference
    assert str(g[f"{tag}.class"]) == ("Adam" if wd == 0 else "AdamW")
    assert len(opt.param_groups) == (1 if wd == 0 else 2)
    for s in range(3):
        opt.zero_grad()
        pw.grad.copy_(g[f"{tag}.gw{s}"])
        pb.grad.copy_(g[f"{tag}.gb{s}"])
        opt.step()
        close(pw, g[f"{tag}.w{s+1}"])
        close(pb, g[f"{tag}.b{s+1}"])
    # gradients assigned as NEW tensors (p.grad = t, what a torch.optim user does) are adopted, not ignored
    pw2 = torch.nn.Parameter(g[f"{tag}.w0"].clone().to(DEV))
    pb2 = torch.nn.Parameter(g[f"{tag}.b0"].clone().to(DEV))
    opt2 = get_optimizer([pw2, pb2], lr=1e-2, wd=wd)
    pw2.grad, pb2.grad = g[f"{tag}.gw0"].to(DEV), g[f"{tag}.gb0"].to(DEV)
    opt2.step()
    close(pw2, g[f"{tag}.w1"])
    close(pb2, g[f"{tag}.b1"])


@pytest.mark.parametrize("wd", [0.0, 3e-2])
@pytest.mark.parametrize("max_norm", [0.5, 1e4, None])
def test_fused_clip_and_adam_vs_torch(wd, max_norm):
    """clip_grad_norm_(params, max_norm) then optim.step() (CTClipTrainer.py:199-202) for 3 steps; max_norm 0.5 clips
    (gradient norm ~ 30), 1e4 does not, None skips the norm pass."""
    from utils.optimizer import get_optimizer
    torch.manual_seed(0)
    shapes = [(37, 19), (5,), (3, 7, 2), (1,), (130,), (64, 64)]
    init = [torch.randn(s) for s in shapes]
    ref_ps = [torch.nn.Parameter(t.clone()) for t in init]
    if wd == 0:
        ref = torch.optim.Adam(ref_ps, lr=3e-3, betas=(0.9, 0.99), eps=1e-8)
    else:
        ref = torch.optim.AdamW([{"params": [p for p in ref_ps if p.ndim >= 2]},
                                 {"params": [p for p in ref_ps if p.ndim < 2], "weight_decay": 0}],
                                lr=3e-3, weight_decay=wd, betas=(0.9, 0.99), eps=1e-8)
    ps = [torch.nn.Parameter(t.clone().to(DEV)) for t in init]
    opt = get_optimizer(ps, lr=3e-3, wd=wd)
    for s in range(3):
        grads = [torch.randn(sh) * (1.0 + s) for sh in shapes]
        for p, gr in zip(ref_ps, grads):
            p.grad = gr.clone()
        norm = torch.nn.utils.clip_grad_norm_(ref_ps, max_norm) if max_norm else None
        ref.step()
        opt.zero_grad()
        for p, gr in zip(ps, grads):
            p.grad.copy_(gr)
        opt.step(max_grad_norm=max_norm)
        if max_norm:
            assert abs(opt.grad_norm() - float(norm)) <= 1e-5 * float(norm)
        for p, r in zip(ps, ref_ps):
            close(p, r, rtol=2e-5, atol=1e-6)
    # the bf16 / transposed weight shadows are invalidated by the step
    from ctclip_hip import ops
    e0 = ops._weight_epoch
    opt.zero_grad()
    opt.step()
    assert ops._weight_epoch == e0 + 1


def test_hip_adam_continues_a_torch_adam_checkpoint():
    """reference CTClipTrainer.py:136-154: `optim` is torch.optim.Adam's state dict over model.parameters()."""
    from ctclip_hip.optim import HipAdam, mark_unused
    torch.manual_seed(1)
    shapes = [(9, 4), (6,), (11,), (2, 3, 5)]
    ref_ps = [torch.nn.Parameter(torch.randn(s)) for s in shapes]
    ref = torch.optim.Adam(ref_ps, lr=1e-2, betas=(0.9, 0.99))
    for _ in range(2):
        for i, p in enumerate(ref_ps):
            p.grad = None if i == 2 else torch.randn(p.shape)            # index 2: a parameter that is never used
        ref.step()
    ps = [torch.nn.Parameter(p.detach().clone().to(DEV)) for p in ref_ps]
    mark_unused(ps[2])
    opt = HipAdam(ps, lr=1.0)
    opt.load_state_dict(ref.state_dict())
    grads = [torch.randn(s) for s in shapes]
    for i, (p, gr) in enumerate(zip(ref_ps, grads)):
        p.grad = None if i == 2 else gr.clone()
    ref.step()
    opt.zero_grad()
    for i, (p, gr) in enumerate(zip(ps, grads)):
        if i != 2:
            p.grad.copy_(gr)
    opt.step()
    for p, r in zip(ps, ref_ps):
        close(p, r, rtol=2e-5, atol=1e-6)
    out = opt.state_dict()
    assert sorted(out["state"]) == [0, 1, 3] and float(out["state"][0]["step"]) == 3.0
    close(out["state"][3]["exp_avg"], ref.state_dict()["state"][3]["exp_avg"], rtol=1e-5, atol=1e-7)
    close(out["state"][3]["exp_avg_sq"], ref.state_dict()["state"][3]["exp_avg_sq"], rtol=1e-5, atol=1e-8)
