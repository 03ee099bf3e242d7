"""Build libctclip_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m ctclip_hip.build            # from ct-clip-ut_amd/
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(PKG), "csrc")
OBJ = os.path.join(PKG, "_build")
LIB = os.path.join(PKG, "libctclip_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-munsafe-fp-atomics", "-fPIC", "-std=c++17", "-Wno-unused-value"]
# diagnostic builds only (e.g. -DCTCLIP_G3_STAMPS for tools/gemm_timeline.py); objects go to their own directory
EXTRA = os.environ.get("CTCLIP_EXTRA_HIPCC_FLAGS", "").split()
if EXTRA:
    OBJ = os.path.join(PKG, "_build_diag")
    LIB = os.path.join(PKG, "libctclip_hip_diag.so")


def _newer(src, dst, extra=()):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(s) > t for s in (src, *extra))


def _compile(src):
    obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    if _newer(src, obj, hdrs):
        cmd = ["hipcc", *FLAGS, *EXTRA, "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    return obj


def build(verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(_compile, srcs))
    if any(_newer(o, LIB) for o in objs):
        r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"built {LIB} from {len(srcs)} sources")
    return LIB


if __name__ == "__main__":
    build()
    sys.exit(0)
