import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "ct-clip-ut_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")
# widen the dispatch gate of the pipelined GEMM so its ragged / split-K / transposed-operand paths are exercised.
# CTCLIP_TEST_PRODUCT_GATES=1 leaves the gates as a training run has them (the model / production tests are run that way once
# per round: profiles/r04_tests_product_gates.txt)
if not os.environ.get("CTCLIP_TEST_PRODUCT_GATES"):
    os.environ.setdefault("CTCLIP_GEMM_V2_ALL", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


def load_golden(name):
    """Load tests/golden/<name>.npz as {key: torch tensor}."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    out = {}
    for k in z.files:
        a = z[k]
        out[k] = torch.from_numpy(a) if a.dtype.kind in "fiub" else a
    return out


def sub(d, prefix):
    return {k[len(prefix):]: v for k, v in d.items() if k.startswith(prefix)}


@pytest.fixture(scope="session")
def golden():
    return load_golden
