"""SURVEY 8(f) "next" rows on the GPU: batched occlusion + integrated gradients (f1), checkpoint interchange / resume
(f3).  The file name sorts after test_hip_model / _optimizer / _production / _trainer on purpose: a failure in a next-row
test under `pytest -x` must not hide the 8(a)/(e) parity evidence that those files hold."""
import math

import pytest
import torch

from test_hip_model import DEV, _config1, check, same_trajectory

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------------------------------- SURVEY 8(f) row f1
def test_batched_occlusion_sensitivity_vs_serial_and_oracle():
    """utils.visualizations.Visualizations._compute_occlusion (reference src/utils/visualizations.py:335-424): windows
    scored in device-side batches with the text side encoded once must give the heat-map of the reference's serial
    one-forward-per-window loop -- (a) against the same HIP model run with batch 1, (b) against the f32 oracle restatement
    of the reference loop."""
    import numpy as np
    from oracle import ctclip_oracle as O
    from utils.visualizations import Visualizations

    class Acc:                                     # the attributes of accelerate.Accelerator the class reads
        is_main_process, process_index, num_processes, device = True, 0, 1, torch.device(DEV)

    clip, data, cfg = _config1()
    st = {k: v.clone() for k, v in clip.state_dict().items()}
    txt, vol = data[0]
    txt1 = {k: v[:1] for k, v in txt.items()}
    image = vol[:1]
    patch, stride = (32, 32, 32), (32, 32, 16)     # 2 x 2 x 3 = 12 windows, overlapping along w
    clip = clip.to(DEV)
    maps = {}
    for b in (5, 1):                               # 5: two full batches + a ragged one; 1: the serial schedule
        vis = Visualizations(clip, Acc(), occlusion_batch=b)
        maps[b] = vis._compute_occlusion(image, txt1, None, patch, stride, 0.0)
    assert maps[5].shape == (64, 64, 64)
    check("occlusion batched vs serial (HIP)", torch.from_numpy(maps[5].copy()), torch.from_numpy(maps[1].copy()), 2e-2)
    _, _, ref = O.occlusion_heatmap(txt1, image, st, cfg, patch, stride, 0.0)
    a, b_ = maps[5].reshape(-1).astype(np.float64), ref.reshape(-1).astype(np.float64)
    corr = float(np.corrcoef(a, b_)[0, 1])
    print(f"  occlusion heat-map vs oracle: correlation {corr:.5f}, max |diff| {np.abs(a - b_).max():.3e}")
    # free-running comparison: every occluded copy goes through the VQ arg-max, whose genuine near-ties flip under bf16
    # noise (the caveat of test_config1_vs_oracle), so the maps agree in shape, not to rounding
    assert corr >= 0.93 and np.abs(a - b_).max() <= 0.3
    # precomputed text embeddings (the reference's `text_embeds` branch, :371-372,384-385)
    emb = clip.encode_text({k: v.to(DEV) for k, v in txt1.items()}).detach()
    m2 = Visualizations(clip, Acc(), occlusion_batch=4)._compute_occlusion(image, None, emb, patch, stride, 0.0)
    check("occlusion with text_embeds", torch.from_numpy(m2.copy()), torch.from_numpy(maps[5].copy()), 2e-2)


# ------------------------------------------------------------------------------------------- SURVEY 8(f) row f3
def test_checkpoint_resume_continues_the_run(tmp_path):
    """save_model / load_model (reference src/utils/CTClipTrainer.py:136-154) carry the optimiser moments and -- beyond
    the reference -- the global step: a trainer restored from the checkpoint takes the same next step as the original.

    What is asserted, in this order: (1) the restore itself is EXACT -- weights, buffers (codebook, cluster sizes), both
    Adam moment arenas and the step counter of the restored trainer equal the saving trainer's bit for bit; (2) the
    resumed step's loss equals the uninterrupted one; (3) the post-step weights agree per element within
    max(4 lr, 1e-3 of the tensor's peak).  (3) cannot be bit-equality: split-K weight gradients are sums of f32 atomics
    (include/ctclip_hip.h lists the order-dependent outputs), and Adam moves an element whose gradient is rounding noise
    (zero-initialised biases, BERT key biases the softmax is invariant to) by ~lr * sign(noise)."""
    from utils.CTClipTrainer import CTClipTrainer
    clip, data, _ = _config1()
    trainer = CTClipTrainer(clip, batch_size=4, results_folder=str(tmp_path))
    (txt0, vol0), (txt1, vol1) = data
    trainer.train_step((vol0, txt0))
    trainer.save_model("ckpt.pt")
    ckpt = trainer.results_folder / "ckpt.pt"
    assert ckpt.exists() and (trainer.results_folder / "architecture.txt").exists()
    saved_state = {k: v.detach().clone() for k, v in trainer.model.state_dict().items()}
    saved_arenas = [None if a is None else (a["p"].clone(), a["m"].clone(), a["v"].clone()) for a in trainer.optim._arenas]
    saved_step = trainer.optim._step
    loss_a = trainer.train_step((vol1, txt1))
    codes_a = trainer.model.visual_transformer.vq.last_indices.clone()
    after_a = {k: v.detach().clone() for k, v in trainer.model.state_dict().items()}
    trainer.grad_sync.close()

    clip_b, _, _ = _config1()                      # fresh model + trainer, then restore
    with torch.no_grad():
        for p in clip_b.parameters():
            p.add_(0.01)                           # make sure the restore is what brings the weights back
    trainer_b = CTClipTrainer(clip_b, batch_size=4, results_folder=None)
    trainer_b.load_model(ckpt)
    assert trainer_b.global_step == 1 and trainer_b.optim._step == saved_step == 1
    # (1) exact restore
    for k, v in trainer_b.model.state_dict().items():
        assert torch.equal(v, saved_state[k]), f"{k} is not restored bit for bit"
    assert len(trainer_b.optim._arenas) == len(saved_arenas)
    for a, s in zip(trainer_b.optim._arenas, saved_arenas):
        assert (a is None) == (s is None)
        if a is not None:
            for name, t in zip("pmv", s):
                assert torch.equal(a[name], t), f"optimiser arena '{name}' is not restored bit for bit"
    # (2) the resumed step
    loss_b = trainer_b.train_step((vol1, txt1))
    codes_b = trainer_b.model.visual_transformer.vq.last_indices.clone()
    assert trainer_b.global_step == 2
    print(f"  resumed step: loss {loss_b:.7f} vs uninterrupted {loss_a:.7f}")
    # both runs start this step from identical bits, so the forward -- and with it the code decisions and the loss -- can
    # only differ by the order of the split-K sums of the 294 912 -> 512 projection (~1e-7)
    assert torch.equal(codes_a, codes_b)
    assert abs(loss_a - loss_b) <= 1e-5 * abs(loss_a)
    # (3) post-step weights, per element, the offender named
    same_trajectory(after_a, trainer_b.model.state_dict(), [codes_a], [codes_b], 1.25e-5, 2, "resumed vs uninterrupted step")
    with pytest.raises(FileNotFoundError):
        trainer_b.load_model(tmp_path / "missing.pt")
    trainer_b.grad_sync.close()


def test_integrated_gradients_vs_oracle():
    """utils.visualizations (reference src/utils/visualizations.py:851-910): batched interpolation points, input gradient
    through ctclip_patch_ln_bwd_dx, against the oracle's autograd over the same path (VQ codes free-running)."""
    import numpy as np
    from oracle import ctclip_oracle as O
    from utils.visualizations import Visualizations

    class Acc:
        is_main_process, process_index, num_processes, device = True, 0, 1, torch.device(DEV)

    clip, data, cfg = _config1()
    st = {k: v.clone() for k, v in clip.state_dict().items()}
    txt, vol = data[0]
    txt1 = {k: v[:1] for k, v in txt.items()}
    image = vol[:1]
    steps = 6
    avg_o, map_o = O.integrated_gradients(txt1, image, st, cfg, steps=steps)
    clip = clip.to(DEV)
    vis = Visualizations(clip, Acc())
    avg, diff = vis._integrated_gradients(image, txt1, steps=steps, ig_batch=4)       # 4 + 2: a ragged last batch
    avg1, _ = vis._integrated_gradients(image, txt1, steps=steps, ig_batch=1)         # the reference's serial schedule
    check("IG batched vs serial (HIP)", avg, avg1, 1e-3)
    a, b_ = avg.cpu().reshape(-1).double(), avg_o.reshape(-1).double()
    cos = float((a @ b_) / (a.norm() * b_.norm()))
    print(f"  IG average input gradient vs oracle: cosine {cos:.5f}, norm ratio {float(a.norm() / b_.norm()):.4f}")
    assert cos >= 0.97 and 0.9 <= float(a.norm() / b_.norm()) <= 1.1
    m = vis.visualize_integrated_gradients(image, txt1, steps=steps, ig_batch=4)
    assert m.shape == map_o.shape and float(m.max()) <= 1.0 + 1e-6
    agree = float(((m > 0) == (map_o > 0)).mean())
    print(f"  IG top-decile mask agreement with the oracle map: {agree:.4f}")
    assert agree >= 0.93


def test_deterministic_mode_makes_the_training_step_bitwise_reproducible(tmp_path):
    """torch.use_deterministic_algorithms(True) -- what the reference's attribution code sets at import
    (src/utils/visualizations.py:29-39) -- selects the ordered form of the one sum whose fast form is order-dependent (the
    relative-position d(bias)); every other reduction of the step is reproducible unconditionally (two-stage partial sums,
    split-K through a workspace, sorted codebook statistics, owner-summed embedding gradients).  Then: two runs of the same
    two training steps end in bit-identical states, and a run resumed from a checkpoint taken after step 1 equals the
    uninterrupted run bit for bit -- losses, weights, codebook, Adam moments."""
    from utils.CTClipTrainer import CTClipTrainer
    was = torch.are_deterministic_algorithms_enabled()
    torch.use_deterministic_algorithms(True)
    try:
        def run(resume_after_first=False):
            clip, data, _ = _config1()
            trainer = CTClipTrainer(clip, batch_size=4, results_folder=str(tmp_path))
            (txt0, vol0), (txt1, vol1) = data
            losses = [trainer.train_step((vol0, txt0))]
            if resume_after_first:
                trainer.save_model("det.pt")
                ckpt = trainer.results_folder / "det.pt"
                trainer.grad_sync.close()
                clip2, _, _ = _config1()
                trainer = CTClipTrainer(clip2, batch_size=4, results_folder=None)
                trainer.load_model(ckpt)
            losses.append(trainer.train_step((vol1, txt1)))
            state = {k: v.detach().clone() for k, v in trainer.model.state_dict().items()}
            moments = [None if a is None else (a["m"].clone(), a["v"].clone()) for a in trainer.optim._arenas]
            trainer.grad_sync.close()
            return losses, state, moments

        la, sa, ma = run()
        for label, (lb, sb, mb) in (("second run", run()), ("resumed run", run(resume_after_first=True))):
            assert la == lb, (label, la, lb)
            for k in sa:
                assert torch.equal(sa[k], sb[k]), f"{label}: {k} differs"
            for x, y in zip(ma, mb):
                assert (x is None) == (y is None)
                if x is not None:
                    assert torch.equal(x[0], y[0]) and torch.equal(x[1], y[1]), f"{label}: Adam moments differ"
        print(f"  two runs and a resumed run of two training steps: bit-identical (losses {la})")
    finally:
        torch.use_deterministic_algorithms(was)


def test_text_tower_on_its_own_stream_changes_no_bit(tmp_path):
    """CTCLIP.forward runs the text tower on a second HIP stream next to the image tower (ops.fork_text_stream) and its backward
    follows there; the optimiser joins the streams.  Only the ORDER in time of independent kernels changes: under deterministic
    algorithms two training steps end in the same bits with the fork on and off -- losses, weights, codebook, Adam moments."""
    from ctclip_hip import ops
    from utils.CTClipTrainer import CTClipTrainer
    was_det, was_on = torch.are_deterministic_algorithms_enabled(), ops._text_stream["on"]
    torch.use_deterministic_algorithms(True)
    try:
        def run(fork):
            ops._text_stream["on"] = fork
            clip, data, _ = _config1()
            trainer = CTClipTrainer(clip, batch_size=4, results_folder=None)
            losses = [trainer.train_step((vol, txt)) for txt, vol in data]
            torch.cuda.synchronize()
            state = {k: v.detach().clone() for k, v in trainer.model.state_dict().items()}
            moments = [None if a is None else (a["m"].clone(), a["v"].clone()) for a in trainer.optim._arenas]
            trainer.grad_sync.close()
            return losses, state, moments

        la, sa, ma = run(False)
        lb, sb, mb = run(True)
        assert ops._text_stream["side"], "the fork never ran"
        assert la == lb, (la, lb)
        for k in sa:
            assert torch.equal(sa[k], sb[k]), f"{k} differs with the text tower on its own stream"
        for x, y in zip(ma, mb):
            assert (x is None) == (y is None)
            if x is not None:
                assert torch.equal(x[0], y[0]) and torch.equal(x[1], y[1]), "Adam moments differ"
        print(f"  one stream vs text tower on its own stream: bit-identical after two steps (losses {la})")
    finally:
        ops._text_stream["on"] = was_on
        torch.use_deterministic_algorithms(was_det)
