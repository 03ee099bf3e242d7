"""CT-CLIP contrastive trainer, MI355X-native.  Drop-in for the reference's `utils.CTClipTrainer`
(src/utils/CTClipTrainer.py:33-304) on the hot path: `train_step`, `loss_function`, `avg_device_loss`, `train`,
`save_model` / `load_model`, with the same constructor keywords.

Differences by design:
  * one process per GPU over `torch.distributed` (backend "nccl" == RCCL on ROCm) instead of Accelerate + DDP:
    gradients live in one flat arena and are averaged by a few large RCCL all-reduces (ctclip_hip.optim.GradSync);
  * bf16 MFMA compute with f32 master weights / accumulation instead of fp16 autocast + GradScaler
    (reference :67,269): no loss scaling is needed;
  * grad-clip(0.5) + Adam run as one fused HIP step without a host sync (reference :199-202);
  * the real-data pipeline (NIfTI datasets, CSV reports: TrainDataset/InferenceDataset/preprocess, reference :85-105)
    and the evaluation plots (metrics.py) are out of scope: pass `train_dl` / `valid_dl` iterables that yield
    `(images, texts)` where `texts` is either a list of strings (needs `tokenizer`) or an already tokenised mapping.
"""
from __future__ import annotations

import os
import time
from datetime import datetime, timedelta
from pathlib import Path
from typing import Iterable, Optional

import torch
import torch.distributed as dist
from torch import nn

from ctclip_hip import ops
from ctclip_hip.optim import GradSync, mark_unused
from models.ctclip import CTCLIP
from utils.optimizer import get_optimizer

# parameters that never receive a gradient on the CT-CLIP path: the reason the reference needs
# DistributedDataParallel(find_unused_parameters=True) (CTClipTrainer.py:64).  They are excluded statically.
STATICALLY_UNUSED = ("text_transformer.pooler.", "visual_transformer.to_patch_emb_first_frame.", ".context_norm.",
                     ".null_kv")


class _Runtime:
    """The few attributes of accelerate.Accelerator the reference code reads (CTClipTrainer.py:70,122,137,160)."""

    def __init__(self):
        self.distributed = dist.is_available() and dist.is_initialized()
        if not self.distributed and "RANK" in os.environ and "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) > 1:
            # "nccl" is RCCL on ROCm (reference :65).  CTCLIP_DIST_BACKEND=gloo lets several ranks share one GPU for
            # rehearsals of the multi-rank path (RCCL refuses duplicate devices).
            backend = os.environ.get("CTCLIP_DIST_BACKEND", "nccl" if torch.cuda.is_available() else "gloo")
            dist.init_process_group(backend=backend, timeout=timedelta(seconds=36000))
            self.distributed = True
        self.process_index = dist.get_rank() if self.distributed else 0
        self.num_processes = dist.get_world_size() if self.distributed else 1
        local = int(os.environ.get("LOCAL_RANK", 0))
        if not torch.cuda.is_available():
            raise RuntimeError("CTClipTrainer: no HIP device visible; the training step has no CPU fallback")
        local = local % max(1, torch.cuda.device_count())                     # shared-GPU rehearsal: ranks wrap around
        self.device = torch.device("cuda", local)
        torch.cuda.set_device(self.device)
        self.is_main_process = self.process_index == 0
        self.sync_gradients = True

    def unwrap_model(self, m):
        return m

    def get_state_dict(self, m):
        return m.state_dict()


class CTClipTrainer(nn.Module):
    def __init__(self, model: CTCLIP, batch_size: int = 1, data_train: Optional[str] = None,
                 data_valid: Optional[str] = None, train_reports: Optional[str] = None,
                 valid_reports: Optional[str] = None, valid_labels: Optional[str] = None,
                 train_metadata: Optional[str] = None, valid_metadata: Optional[str] = None, tokenizer=None,
                 lr: float = 1.25e-5, wd: float = 0.0, max_grad_norm: float = 0.5, results_folder: str = "./results",
                 num_workers: int = 8, num_epochs: int = 10, num_save_split: int = 5, num_train_samples: int = 100,
                 num_valid_samples: int = 20, save_best_model: bool = False, *, train_dl: Optional[Iterable] = None,
                 valid_dl: Optional[Iterable] = None, max_text_length: int = 512):
        super().__init__()
        self.accelerator = _Runtime()
        self.maybe_print = print if self.accelerator.is_main_process else (lambda *a, **k: None)
        if train_dl is None and data_train is not None:
            raise NotImplementedError("the NIfTI/CSV dataset pipeline (reference TrainDataset/InferenceDataset) is out of "
                                      "scope of this build; pass train_dl=/valid_dl= iterables of (images, texts)")
        self.model = model.to(self.accelerator.device)
        self.model.accelerator = self.accelerator                              # reference :73
        self.tokenizer = tokenizer
        self.max_text_length = max_text_length
        self.num_epochs, self.num_save_split, self.batch_size = num_epochs, num_save_split, batch_size
        self.max_grad_norm, self.save_best_model = max_grad_norm, save_best_model
        self.train_dl, self.valid_dl = train_dl, valid_dl

        # every parameter goes to the optimiser, as in the reference (:107), so the parameter indices of an "optim"
        # checkpoint are interchangeable; the statically unused ones are tagged and get no arena slot / Adam state
        # (torch's Adam creates none for a parameter whose grad stays None either)
        for n, p in self.model.named_parameters():
            if any(tag in n for tag in STATICALLY_UNUSED):
                mark_unused(p)
        self.optim = get_optimizer(self.model.parameters(), lr=lr, wd=wd)      # reference :107
        self.grad_sync = GradSync(self.optim)
        self.grad_sync.prepare()               # with a process group: plan the buckets, tag this optimiser's parameters
        if self.accelerator.distributed:                                       # DDP broadcasts rank-0 weights at wrap time
            for t in list(self.model.parameters()) + list(self.model.buffers()):
                dist.broadcast(t.data, src=0)

        self.metrics, self.train_losses, self.valid_losses = [], {}, []
        self.best_score = float("inf")
        self.global_step = 0
        self.results_folder = None
        if self.accelerator.process_index == 0 and results_folder:
            root = Path(results_folder) / datetime.now().strftime("%d-%m-%Y")
            root.mkdir(parents=True, exist_ok=True)
            idx = len([d for d in root.iterdir() if d.is_dir()]) + 1
            self.results_folder = root / str(idx)
            self.results_folder.mkdir(parents=True, exist_ok=True)

    # ---- checkpoints (reference :136-154) ---------------------------------------------------------------------
    def save_model(self, name, log=None):
        if self.accelerator.is_main_process and self.results_folder is not None:
            # "model" / "optim" as in the reference (:140-143); the rest is what a real resume needs and the reference
            # drops: global step, loss history, RNG streams
            pkg = {"model": self.model.state_dict(), "optim": self.optim.state_dict(), "step": self.global_step,
                   "train_losses": self.train_losses, "valid_losses": self.valid_losses, "best_score": self.best_score,
                   "rng": {"torch": torch.get_rng_state(), "cuda": torch.cuda.get_rng_state(self.accelerator.device)}}
            torch.save(pkg, str(self.results_folder / name))
            with open(self.results_folder / "architecture.txt", "w") as f:
                f.write(str(self.model))
            if log:
                self.maybe_print(log)

    def load_model(self, path):
        path = Path(path)
        if not path.exists():
            raise FileNotFoundError(f"Checkpoint not found at {path}")
        pkg = torch.load(path, map_location=self.accelerator.device, weights_only=False)
        self.model.load_state_dict(pkg["model"])
        self.optim.load_state_dict(pkg["optim"])
        ops.bump_weight_epoch()
        self.global_step = int(pkg.get("step", 0))
        self.train_losses = pkg.get("train_losses", self.train_losses)
        self.valid_losses = pkg.get("valid_losses", self.valid_losses)
        self.best_score = pkg.get("best_score", self.best_score)
        if "rng" in pkg:
            torch.set_rng_state(pkg["rng"]["torch"].cpu())
            torch.cuda.set_rng_state(pkg["rng"]["cuda"].cpu(), self.accelerator.device)

    # ---- loss bookkeeping (reference :156-175) -----------------------------------------------------------------
    def avg_device_loss(self, loss):
        """Mean of a per-rank scalar over the ranks (reference :156-162: gather_for_metrics + mean), as a python float."""
        t = torch.as_tensor(float(loss), dtype=torch.float32, device=self.accelerator.device)
        if dist.is_available() and dist.is_initialized():
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            t = t / dist.get_world_size()
        return t.item()

    def loss_function(self, sim_matrix, targets=None):
        if targets is not None:
            raise NotImplementedError("only the diagonal targets of the reference training loop are implemented")
        return ops.InfoNCEFn.apply(sim_matrix)

    def _tokens(self, texts):
        dev = self.accelerator.device
        if isinstance(texts, dict) or hasattr(texts, "keys"):
            return {k: v.to(dev) for k, v in texts.items()}
        if self.tokenizer is None:
            raise RuntimeError("string reports need a tokenizer (the CXR-BERT tokenizer is not bundled); "
                               "pass tokenizer= or yield tokenised mappings")
        tok = self.tokenizer(list(texts), return_tensors="pt", padding="max_length", truncation=True,
                             max_length=self.max_text_length)                # reference :186-192
        return {k: v.to(dev) for k, v in tok.items()}

    # ---- one optimisation step (reference :177-204) ------------------------------------------------------------
    def train_step(self, batch, return_tensor: bool = False):
        self.model.train()
        self.optim.zero_grad()
        images, texts = batch
        images = images.to(self.accelerator.device, non_blocking=True)
        text_tokens = self._tokens(texts)
        sim_matrix, *_ = self.model(text_tokens, images)
        loss = self.loss_function(sim_matrix)
        loss.backward()
        ops.join_side_streams()
        self.grad_sync.all_reduce_grads()
        self.optim.step(max_grad_norm=self.max_grad_norm if self.max_grad_norm else None)
        vq = getattr(getattr(self.model, "visual_transformer", None), "vq", None)
        if hasattr(vq, "flush_ema"):
            vq.flush_ema()                      # codebook statistics were all-reduced under the backward pass
        self.global_step += 1
        return loss.detach() if return_tensor else loss.item()

    @torch.no_grad()
    def evaluate(self, epoch):
        if self.valid_dl is None:
            return None
        self.model.eval()
        total, n = 0.0, 0
        for batch in self.valid_dl:
            images, texts = batch[0], batch[1]
            sim, *_ = self.model(self._tokens(texts), images.to(self.accelerator.device))
            total += self.loss_function(sim).item()
            n += 1
        avg = self.avg_device_loss(total / max(n, 1))
        self.valid_losses.append(avg)
        self.maybe_print(f"Epoch {epoch} - Validation Loss: {avg:.4f}")
        if self.accelerator.is_main_process and (epoch == 0 or (avg < self.best_score and self.save_best_model)):
            self.best_score = avg
            self.save_model("best_checkpoint.pt", f"New best model saved at epoch {epoch}, validation loss {avg:.4f}")
        return avg

    def train(self):
        """Reference :252-304: epochs of train_step + per-step averaged loss, evaluate(0) after the very first step and
        evaluate(epoch) after every epoch (which saves `best_checkpoint.pt` as the reference does, :238-244)."""
        if self.train_dl is None:
            raise RuntimeError("no training data: pass train_dl=")
        start = time.time()
        n_batches = len(self.train_dl) if hasattr(self.train_dl, "__len__") else None
        save_at = max(1, n_batches // self.num_save_split) if n_batches else 1                  # reference :257
        self.maybe_print("Training started")
        durations = []
        for epoch in range(1, self.num_epochs + 1):
            t0 = time.time()
            self.maybe_print(f"\nStarting Epoch {epoch}/{self.num_epochs}")
            sampler = getattr(self.train_dl, "sampler", None)
            if hasattr(sampler, "set_epoch"):
                sampler.set_epoch(epoch)                                                         # reference :265
            total, steps = 0.0, 0
            for step, batch in enumerate(self.train_dl, start=1):
                loss = self.train_step(batch)
                total += loss
                steps += 1
                avg_step = self.avg_device_loss(loss)                                            # reference :272
                if step % save_at == 0:
                    self.train_losses.setdefault("steps", []).append(avg_step)
                if epoch == 1 and step == 1:                                                     # reference :278-281
                    self.train_losses.setdefault("epochs", []).append(avg_step)
                    self.train_losses.setdefault("steps", []).append(avg_step)
                    self.evaluate(0)
                of = f"/{n_batches}" if n_batches else ""
                self.maybe_print(f"Epoch {epoch} | Step {step}{of} | Avg Loss: {avg_step:.6f}")
            avg_epoch = self.avg_device_loss(total / max(steps, 1))
            self.train_losses.setdefault("epochs", []).append(avg_epoch)
            durations.append(time.time() - t0)
            self.maybe_print(f"Epoch {epoch} completed. Average Loss: {avg_epoch:.6f}")
            self.maybe_print(f"Time taken for Epoch {epoch}: {timedelta(seconds=durations[-1])}")
            t1 = time.time()
            self.evaluate(epoch)
            self.maybe_print(f"Time taken for evaluation: {timedelta(seconds=time.time() - t1)}")
        self.maybe_print("Training completed")
        self.maybe_print(f"Total Training Time: {timedelta(seconds=time.time() - start)}")
        if durations:
            self.maybe_print(f"Average Epoch Duration: {timedelta(seconds=sum(durations) / len(durations))}")
