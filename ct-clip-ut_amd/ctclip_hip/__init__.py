"""ctclip_hip: Python binding of libctclip_hip.so (hand-written gfx950 kernels, C ABI in include/ctclip_hip.h).

There is no CPU or PyTorch-eager fallback: importing `ctclip_hip.lib.hip` without the built library, or
calling an op with CPU tensors, raises.
"""
from .lib import hip, library_path, HipLibraryMissing  # noqa: F401
