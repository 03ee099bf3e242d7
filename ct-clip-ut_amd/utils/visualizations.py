"""Attribution harness, MI355X-native: SURVEY.md §8(f) row f1 -- occlusion sensitivity.

Mirror of the reference's `utils.visualizations.Visualizations` (src/utils/visualizations.py:73-105) for the one
visualisation that is a pure function of the hot path's forward: `_compute_occlusion` (:335-424) and its driver
`visualize_occlusion_sensitivity` (:1029-1083, without the GIF rendering).

The reference slides a voxel window over the volume and runs ONE B=1 forward per window (12 167 forwards at the default
window (20,40,40) / stride (10,20,20) on a 240x480x480 volume, ~10 forwards/s, a `.item()` sync and two host numpy
updates per window).  Here:
  * the text side is encoded once (CLS output cached and passed as `text_embeds`, ctclip.py:107);
  * windows are processed `occlusion_batch` at a time: the occluded copies are built on the device, one batched no-grad
    forward through the HIP encoder scores them all (sim [B,1]);
  * importance, heat-map and count-map accumulate on the device without a host sync;
  * ranks take contiguous slices of the window list exactly as the reference does (:352-362, extra windows dropped) and
    the maps are reduced to rank 0 with one RCCL reduce each (:404-407).
Same return value: the normalised, thresholded, rot90'd heat-map on the main process, None elsewhere.

`visualize_integrated_gradients` (:851-910) is here too: the `steps` interpolation points go through batched forward +
backward passes whose input gradient comes from ctclip_patch_ln_bwd_dx.

Out of scope here (NotImplementedError): attention maps / rollout, Grad-CAM and the matplotlib overlays."""
from __future__ import annotations

from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist
import torch.nn.functional as F


class Visualizations:
    def __init__(self, model, accelerator, dataset=None, dist_dataloader=None, batch_size: int = 1, results_folder=None,
                 diff_embeds_folder=None, tokenizer=None, occlusion_batch: int = 32, max_windows=None):
        self.model = model.module if hasattr(model, "module") else model
        self.accelerator = accelerator
        self.dataset, self.dist_dataloader, self.batch_size = dataset, dist_dataloader, batch_size
        self.tokenizer = tokenizer
        self.results_folder = Path(results_folder) if results_folder is not None else None
        self.diff_embeds_folder = diff_embeds_folder
        self.maybe_print = print if self.accelerator.is_main_process else (lambda *a, **k: None)
        self.rank = self.accelerator.process_index
        self.world_size = self.accelerator.num_processes
        self.occlusion_batch = int(occlusion_batch)
        self.max_windows = max_windows            # benchmarking aid: score only the first N windows of the rank's slice

    def _results_subdirectory(self, visualization_name):                     # reference :108-123
        sub = self.results_folder / visualization_name
        sub.mkdir(parents=True, exist_ok=True)
        idx = len([d for d in sub.iterdir() if d.is_dir()]) + 1
        sub = sub / str(idx)
        sub.mkdir(parents=True, exist_ok=True)
        return sub

    # ---- reference :335-424 ------------------------------------------------------------------------------------
    @torch.no_grad()
    def _compute_occlusion(self, image, text_tokens, text_embeds, patch_size, stride, threshold):
        dev = self.accelerator.device
        model = self.model
        image = image.to(dev)
        _, _, D, H, W = image.shape
        coords = [(d, h, w)
                  for d in range(0, D - patch_size[0] + 1, stride[0])
                  for h in range(0, H - patch_size[1] + 1, stride[1])
                  for w in range(0, W - patch_size[2] + 1, stride[2])]
        per_rank = len(coords) // self.world_size                            # :352-356: extra windows are dropped
        coords = coords[:per_rank * self.world_size][self.rank * per_rank:(self.rank + 1) * per_rank]
        if self.max_windows is not None:
            coords = coords[:self.max_windows]
        self.maybe_print(f"[Rank {self.rank}] Total patches to go through: {len(coords)}")

        was_training, gathered = model.training, model.gather_negatives
        model.eval()
        model.gather_negatives = False        # every rank scores its own windows; sim[rank,rank] of the reference == sim[i,0] here
        try:
            if isinstance(text_embeds, torch.Tensor) and text_embeds.ndim > 1:
                cls = text_embeds.to(dev)
            else:
                cls = model.encode_text({k: v.to(dev) for k, v in text_tokens.items()})      # encoded once, not per window
            cls = cls[:1]
            original = model(None, image, cls)[0][0, 0].float()              # :370-375
            heat = torch.zeros(D, H, W, dtype=torch.float32, device=dev)
            count = torch.zeros(D, H, W, dtype=torch.float32, device=dev)
            B = max(1, self.occlusion_batch)
            for i0 in range(0, len(coords), B):
                chunk = coords[i0:i0 + B]
                occluded = image.expand(len(chunk), -1, -1, -1, -1).clone()
                for j, (d, h, w) in enumerate(chunk):                        # :380-381
                    occluded[j, :, d:d + patch_size[0], h:h + patch_size[1], w:w + patch_size[2]] = -1
                scores = model(None, occluded, cls)[0][:, 0].float()         # one batched forward, sim [len(chunk), 1]
                importance = (original - scores).clamp_min(0)                # :390
                for j, (d, h, w) in enumerate(chunk):                        # :391-392, on the device, no host sync
                    heat[d:d + patch_size[0], h:h + patch_size[1], w:w + patch_size[2]] += importance[j]
                    count[d:d + patch_size[0], h:h + patch_size[1], w:w + patch_size[2]] += 1
        finally:
            model.gather_negatives = gathered
            model.train(was_training)
        if self.world_size > 1:                                              # :404-407
            dist.reduce(heat, dst=0, op=dist.ReduceOp.SUM)
            dist.reduce(count, dst=0, op=dist.ReduceOp.SUM)
        if not self.accelerator.is_main_process:
            return None
        count[count == 0] = 1                                                # :409-424
        heat = heat / count
        heat = (heat - heat.min()) / (heat.max() - heat.min() + 1e-8)
        full = F.interpolate(heat[None, None], size=(D, H, W), mode="trilinear", align_corners=False)[0, 0]
        full = full.cpu().numpy()
        full[full < threshold] = 0
        return np.rot90(full, k=-1, axes=(1, 2))

    # ---- reference :1029-1083 (heat-maps and .npy files; the GIF overlays are out of scope) ------------------------
    def visualize_occlusion_sensitivity(self, image, text_tokens, labels=None, scan_name="scan", original_scan_path=None,
                                        patch_size=(20, 40, 40), stride=(10, 20, 20), use_text_embeds=False, prompt="",
                                        pathologies=None):
        threshold = 0.0
        heatmaps = {}
        if use_text_embeds:
            emb = np.load(self.diff_embeds_folder, allow_pickle=True).item()
            dev = self.accelerator.device
            tensors = {k: torch.tensor(v, dtype=torch.float32, device=dev).unsqueeze(0) for k, v in emb.items()}
            names = pathologies if pathologies is not None else sorted(tensors)
            positive = [names[i] for i in (labels == 1).nonzero(as_tuple=True)[0].tolist()]
            for name in positive:
                self.maybe_print("Processing pathology:", name)
                heatmaps[name] = self._compute_occlusion(image, text_tokens, tensors[name], patch_size, stride, threshold)
        else:
            heatmaps[prompt or "report"] = self._compute_occlusion(image, text_tokens, None, patch_size, stride, threshold)
        if self.accelerator.is_main_process and self.results_folder is not None:
            out = self._results_subdirectory("occlusion")
            np.save(out / f"{scan_name}_{str(patch_size)}_{str(stride)}_{prompt}_heatmaps.npy", heatmaps)
        return heatmaps

    # ---- reference :851-910 ----------------------------------------------------------------------------------
    def _integrated_gradients(self, image, text_tokens, steps=50, text_embeds=None, ig_batch=10):
        """Average d sim / d volume over `steps` points of the straight path from the all-ones baseline to the image
        (reference :852-878).  The reference runs `steps` B=1 forward+backward passes and re-encodes the report each time;
        here the text side is encoded once and `ig_batch` interpolation points go through one batched forward + backward
        (the samples of a batch are independent, so d(sum_i sim[i,0]) / d(x_i) is each point's own gradient).
        Needs d(volume) of the tubelet embedding: ctclip_patch_ln_bwd_dx."""
        dev = self.accelerator.device
        model = self.model
        image = image.to(dev).float()
        baseline = torch.ones_like(image)
        diff = image - baseline
        was_training, gathered = model.training, model.gather_negatives
        model.eval()
        model.gather_negatives = False
        try:
            with torch.no_grad():
                if isinstance(text_embeds, torch.Tensor) and text_embeds.ndim > 1:
                    cls = text_embeds.to(dev)[:1]
                else:
                    cls = model.encode_text({k: v.to(dev) for k, v in text_tokens.items()})[:1]
            alphas = torch.linspace(0, 1, steps, device=dev)
            total = torch.zeros_like(image)
            # This is an INPUT-gradient pass: the parameter gradients it leaves in the arena are thrown away.  Under a process
            # group the trainer's GradSync would start bucket all-reduces as they are produced -- unmatched if only some ranks
            # attribute, and racing the zero_grad below -- so every synchroniser the parameters are tagged with is put into
            # no_sync() for the duration (DDP's rule for a backward that must not communicate).
            import contextlib
            syncs = {id(s_): s_ for s_ in (getattr(p, "_ctclip_sync", None) for p in model.parameters()) if s_ is not None}
            with contextlib.ExitStack() as stack:
                for s_ in syncs.values():
                    stack.enter_context(s_.no_sync())
                for i0 in range(0, steps, ig_batch):
                    a = alphas[i0:i0 + ig_batch].view(-1, 1, 1, 1, 1)
                    x = (baseline + a * diff).detach().requires_grad_()
                    with torch.enable_grad():
                        sim = model(None, x, cls)[0]
                        sim[:, 0].sum().backward()
                    total += x.grad.sum(dim=0, keepdim=True)
                    model.zero_grad(set_to_none=False)
        finally:
            model.gather_negatives = gathered
            model.train(was_training)
        return total / steps, diff

    def visualize_integrated_gradients(self, image, text_tokens, labels=None, scan_name="scan", original_scan_path=None,
                                       steps=50, ig_batch=10):
        avg, diff = self._integrated_gradients(image, text_tokens, steps=steps, ig_batch=ig_batch)
        ig = (diff * avg).squeeze().relu()                                   # :879-898
        ig = (ig - ig.min()) / (ig.max() + 1e-8)
        ig = ig.cpu().numpy()
        ig = np.where(ig >= np.quantile(ig, 0.90), ig, 0.0)
        ig = ig ** 0.05
        ig = ig / (ig.max() + 1e-8)
        ig = np.rot90(ig, k=-1, axes=(1, 2))
        if self.accelerator.is_main_process and self.results_folder is not None:
            out = self._results_subdirectory("integrated_gradients")
            np.save(out / f"{scan_name}.npy", ig)
        return ig

    def visualize(self, **kwargs):
        raise NotImplementedError("only occlusion sensitivity (visualize_occlusion_sensitivity / _compute_occlusion) is "
                                  "and integrated gradients (visualize_integrated_gradients) are implemented on the MI355X "
                                  "path; attention maps and Grad-CAM are next rows (SURVEY.md 8f)")
