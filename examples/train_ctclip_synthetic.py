#!/usr/bin/env python3
"""The reference's entry script (src/train_ctclip.py:17-60) on the MI355X-native modules, with synthetic data.

Run from the repo root on a machine with an MI355X:

    python examples/train_ctclip_synthetic.py                      # one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        examples/train_ctclip_synthetic.py                         # 8 GPUs, RCCL gradient all-reduce + latent all-gather

`ct-clip-ut_amd/` plays the role of the reference's `src/` directory on sys.path: the imports, the CTViT / CTCLIP /
CTClipTrainer constructor calls and `trainer.train()` below are the reference script's own lines.  Only what cannot exist
offline differs, and each difference is marked:
  (1) `BertModel.from_pretrained("microsoft/BiomedVLP-CXR-BERT-specialized")` needs the network: a random-init
      `BertModel(BertConfig())` of the same architecture (BERT-base, vocab 30522) stands in;
  (2) `clip.load(".../ctclip_v2.pt", strict=False)` runs only if CTCLIP_PRETRAINED points at a checkpoint;
  (3) the NIfTI/CSV dataset arguments are replaced by `train_dl=` / `valid_dl=` iterables of synthetic
      (volume, tokenised report) batches (real-data IO is out of scope);
  (4) the unused `monai` imports of the reference script (:3-4) are dropped (monai is not installed).
CTCLIP_EXAMPLE_SMALL=1 shrinks the model and volumes so the script finishes in seconds (used by the test-suite).
"""
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ct-clip-ut_amd"))

import torch
from models.ctclip import CTCLIP
from utils.CTClipTrainer import CTClipTrainer
from utils.ctvit import CTViT
from transformers import BertModel, BertConfig
from transformers.utils import logging
from torch import nn  # noqa: F401

warnings.simplefilter("ignore")
logging.set_verbosity_error()
torch.set_printoptions(profile="default")
torch.autograd.set_detect_anomaly(False)

SMALL = os.environ.get("CTCLIP_EXAMPLE_SMALL") == "1"
torch.manual_seed(0)

text_encoder = BertModel(BertConfig() if not SMALL else BertConfig(                                   # (1)
    hidden_size=64, num_hidden_layers=2, num_attention_heads=2, intermediate_size=128, vocab_size=211,
    max_position_embeddings=64))

vit_encoder = CTViT(
    dim = 512,
    codebook_size = 8192,
    image_size = 480,
    patch_size = 20,
    temporal_patch_size = 10,
    spatial_depth = 4,
    temporal_depth = 4,
    dim_head = 32,
    heads = 8
) if not SMALL else CTViT(dim=64, codebook_size=256, image_size=64, patch_size=16, temporal_patch_size=16,
                          spatial_depth=2, temporal_depth=2, dim_head=32, heads=2)

clip = CTCLIP(
    text_encoder = text_encoder,
    image_encoder = vit_encoder,
    dim_text = 768 if not SMALL else 64,
    dim_image = 294912 if not SMALL else 4 * 4 * 64,
    dim_latent = 512 if not SMALL else 32
)

if os.environ.get("CTCLIP_PRETRAINED"):                                                               # (2)
    clip.load(os.environ["CTCLIP_PRETRAINED"], strict=False)


class SyntheticPairs:                                                                                 # (3)
    """A re-iterable, sized stand-in for the reference's DataLoader: `n_batches` batches of (volumes in the value range
    of preprocess.py:135-149, tokenised reports)."""

    def __init__(self, n_batches, batch_size, seed, depth, size, length, vocab):
        self.args = (n_batches, batch_size, seed, depth, size, length, vocab)

    def __len__(self):
        return self.args[0]

    def __iter__(self):
        n, b, seed, depth, size, length, vocab = self.args
        g = torch.Generator().manual_seed(seed + 1000 * int(os.environ.get("RANK", 0)))
        for _ in range(n):
            vol = (torch.randn(b, 1, depth, size, size, generator=g) * 0.5).clamp_(-1, 1)
            ids = torch.randint(0, vocab, (b, length), generator=g)
            lens = torch.randint(length // 4, length + 1, (b,), generator=g)
            mask = (torch.arange(length)[None] < lens[:, None]).long()
            yield vol, {"input_ids": ids, "token_type_ids": torch.zeros_like(ids), "attention_mask": mask}


batch_size = 1 if not SMALL else 4
shape = dict(depth=240, size=480, length=128, vocab=30522) if not SMALL else dict(depth=64, size=64, length=32, vocab=211)
steps = int(os.environ.get("CTCLIP_EXAMPLE_STEPS", 4 if not SMALL else 2))

trainer = CTClipTrainer(
    clip,
    train_dl = SyntheticPairs(steps, batch_size, 1234, **shape),                                       # (3)
    valid_dl = SyntheticPairs(1, batch_size, 4321, **shape),
    results_folder = os.environ.get("CTCLIP_EXAMPLE_RESULTS", "./results/train/ctclip"),
    batch_size = batch_size,
    num_workers = 4,
    num_epochs = int(os.environ.get("CTCLIP_EXAMPLE_EPOCHS", 1)),
    num_save_split = 1,
    num_train_samples = 5000,
    num_valid_samples = 1000,
    save_best_model = True
)

trainer.train()
