"""GPU tests of the trainer surface beyond one step (reference src/utils/CTClipTrainer.py:206-304, src/train_ctclip.py:17-60)
and of the RCCL code paths of the data-parallel design at world size 1: on one GPU the
collectives move no data between ranks, but `backend="nccl"` IS RCCL on ROCm, so every branch that the 8-GPU run takes
(`all_gather_into_tensor`, async AVG all-reduce of arena buckets issued from the weight-gradient stream during backward,
the background codebook-statistics reduce, the loss average) executes here once and must leave the step's result unchanged.
"""
import math
import os
import runpy
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_example_script_trains_and_evaluates(tmp_path, monkeypatch, capsys):
    """examples/train_ctclip_synthetic.py = the reference's train_ctclip.py with synthetic data: trainer.train() runs an
    epoch of two steps, evaluate(0) after the first step and evaluate(1) after the epoch, best checkpoint saved."""
    monkeypatch.setenv("CTCLIP_EXAMPLE_SMALL", "1")
    monkeypatch.setenv("CTCLIP_EXAMPLE_RESULTS", str(tmp_path))
    monkeypatch.setenv("CTCLIP_EXAMPLE_STEPS", "2")
    monkeypatch.setenv("CTCLIP_EXAMPLE_EPOCHS", "1")
    ns = runpy.run_path(os.path.join(ROOT, "examples", "train_ctclip_synthetic.py"), run_name="__main__")
    trainer = ns["trainer"]
    out = capsys.readouterr().out
    assert "Training started" in out and "Epoch 1 | Step 2/2" in out and "Training completed" in out
    assert trainer.global_step == 2
    assert len(trainer.valid_losses) == 2 and all(math.isfinite(v) for v in trainer.valid_losses)    # evaluate(0), evaluate(1)
    assert len(trainer.train_losses["epochs"]) == 2                     # first-step loss + the epoch average (reference :278-286)
    assert len(trainer.train_losses["steps"]) >= 2
    ckpt = trainer.results_folder / "best_checkpoint.pt"
    assert ckpt.exists() and (trainer.results_folder / "architecture.txt").exists()
    pkg = torch.load(ckpt, map_location="cpu", weights_only=False)
    assert set(pkg["model"]) == set(trainer.model.state_dict())
    # the optimiser entry is torch.optim.Adam's format over ALL model parameters, as the reference writes it
    n_params = len(list(trainer.model.parameters()))
    assert pkg["optim"]["param_groups"][0]["params"] == list(range(n_params))
    st = next(iter(pkg["optim"]["state"].values()))
    assert set(st) == {"step", "exp_avg", "exp_avg_sq"}
    # evaluate() on its own, eval mode, no parameter change
    before = {k: v.clone() for k, v in trainer.model.state_dict().items()}
    v = trainer.evaluate(7)
    assert math.isfinite(v) and len(trainer.valid_losses) == 3
    for k, t in trainer.model.state_dict().items():
        assert torch.equal(t, before[k]), k
    # ... and the checkpoint loads into the REFERENCE's optimiser class (torch.optim.Adam over model.parameters())
    ref_opt = torch.optim.Adam([torch.nn.Parameter(p.detach().cpu().clone()) for p in trainer.model.parameters()], lr=1.0)
    ref_opt.load_state_dict(pkg["optim"])
    assert ref_opt.param_groups[0]["lr"] == 1.25e-5 and ref_opt.param_groups[0]["betas"] == (0.9, 0.99)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_rccl_branches_run_at_world_size_one():
    import torch.distributed as dist
    from test_hip_model import _config1
    from ctclip_hip import ops
    from utils.CTClipTrainer import CTClipTrainer
    # single-process result first
    clip, data, _ = _config1()
    trainer = CTClipTrainer(clip, batch_size=4, results_folder=None)
    losses, codes = [], []
    for txt, vol in data:
        losses.append(trainer.train_step((vol, txt)))
        codes.append(clip.visual_transformer.vq.last_indices.clone())
    ref_state = {k: v.detach().clone() for k, v in clip.state_dict().items()}
    trainer.grad_sync.close()
    assert not trainer.accelerator.distributed

    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        clip2, data2, _ = _config1()
        tr2 = CTClipTrainer(clip2, batch_size=4, results_folder=None)
        assert tr2.accelerator.distributed and tr2.accelerator.num_processes == 1
        sync = tr2.grad_sync
        launched = []
        real = sync._launch
        sync._launch = lambda b: (launched.append((b["start"], b["stop"])), real(b))[1]
        losses2, codes2 = [], []
        for txt, vol in data2:
            n0 = len(launched)
            losses2.append(tr2.train_step((vol, txt)))
            codes2.append(clip2.visual_transformer.vq.last_indices.clone())
            assert len(launched) - n0 == len(sync._buckets)              # every bucket reduced exactly once per step
        state_after_two = {k: v.detach().clone() for k, v in clip2.state_dict().items()}
        assert sync._bucket_of and all(b["handle"] is None for b in sync._buckets)
        assert tr2.model.visual_transformer.vq._pending_ema is None      # the background codebook reduce was joined
        # one more step, with the text tower forced onto ONE stream: the collectives leave exactly once either way
        text_was = ops._text_stream["on"]
        ops._text_stream["on"] = False
        try:
            n0 = len(launched)
            extra_loss = tr2.train_step((data2[0][1], data2[0][0]))
            assert len(launched) - n0 == len(sync._buckets) and math.isfinite(extra_loss)
        finally:
            ops._text_stream["on"] = text_was
        clip2.load_state_dict(state_after_two)                            # the comparison below is about the two default steps
        avg = tr2.avg_device_loss(losses2[-1])
        assert abs(avg - losses2[-1]) < 1e-6
        for i, (a, b) in enumerate(zip(losses, losses2)):
            print(f"  loss single-process {a:.7f}  through RCCL {b:.7f}")
            # f32 atomics reorder sums, nothing else differs -- unless that noise flipped a VQ near-tie in this step
            # (same_trajectory below bounds how many may), which moves the loss of a 64-token toy by up to ~1e-3
            same_codes = torch.equal(codes[i], codes2[i])
            assert abs(a - b) <= (1e-5 if same_codes else 5e-3) * abs(a)
        from test_hip_model import same_trajectory
        same_trajectory(ref_state, clip2.state_dict(), codes, codes2, 1.25e-5, 2, "single process vs RCCL at world size 1")
        sync.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("local", [False, True])
def test_bench_two_gloo_ranks_share_the_gpu(local):
    """`python bench.py --gpus 2` from a plain shell (how the driver starts the scaling run), rehearsed on ONE device: the parent
    counts devices from sysfs and starts two ranks, which rendezvous over gloo (CTCLIP_DIST_BACKEND) and both train on cuda:0 --
    hook-driven gradient buckets, the fused latent all-gather (BASELINE configs[3]) or `--local-negatives` (configs[2]), the
    max-over-ranks timing -- and rank 0 prints the ONE JSON line: n_gpus, global_batch and `negatives` must say what ran."""
    import json
    import subprocess
    import sys
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["CTCLIP_DIST_BACKEND"] = "gloo"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--small", "--steps", "2", "--warmup", "1", "--batch", "2",
           "--lean"] + (["--local-negatives"] if local else [])
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]                                # exactly one line on stdout: the JSON
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["steps"] == 2 and line["warmup"] == 1
    cfg = line["config"]
    assert cfg["per_gpu_batch"] == 2 and cfg["global_batch"] == 4 and cfg["parallelism"] == "dp2"
    assert cfg["negatives"] == ("local" if local else "global (all-gather)")
    assert math.isfinite(cfg["final_loss"]) and line["value"] > 0
    assert "rank 1 of 2: process group up (gloo)" in r.stderr
