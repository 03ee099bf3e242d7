"""ctypes loader for libctclip_hip.so.  Prototypes are parsed from include/ctclip_hip.h so the header is
the single source of truth for the C ABI."""
import ctypes
import os
import re

import torch

PKG = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(PKG))
HEADER = os.path.join(REPO, "include", "ctclip_hip.h")


class HipLibraryMissing(RuntimeError):
    pass


def library_path():
    # CTCLIP_HIP_LIB selects another build of the same ABI (diagnostic builds made with CTCLIP_EXTRA_HIPCC_FLAGS)
    return os.environ.get("CTCLIP_HIP_LIB") or os.path.join(PKG, "libctclip_hip.so")


_CTYPES = {"int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float}


def parse_header(path=HEADER):
    """-> {name: [(ctype, argname, is_pointer)]} for every `int ctclip_*(...)` declaration."""
    txt = open(path).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    protos = {}
    for m in re.finditer(r"\bint\s+(ctclip_\w+)\s*\(([^;]*?)\)\s*;", txt, flags=re.S):
        args = []
        for a in m.group(2).split(","):
            a = " ".join(a.split())
            ptr = "*" in a
            name = re.findall(r"(\w+)\s*$", a)[0]
            base = a.replace("const", "").replace("*", " ").split()[0]
            args.append((ctypes.c_void_p if ptr else _CTYPES[base], name, ptr))
        protos[m.group(1)] = args
    return protos


PARTIALS_FLOATS = 1 << 21          # CTCLIP_PARTIALS_FLOATS of include/ctclip_hip.h
_partials = {}


def partials_scratch():
    """The scratch behind the `partials` argument of the two-stage reductions: one buffer per (device, stream) -- kernels
    of one stream run in order, so consecutive calls may share it."""
    key = (torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream)
    t = _partials.get(key)
    if t is None:
        t = _partials[key] = torch.empty(PARTIALS_FLOATS, dtype=torch.float32, device="cuda")
    return t


SPLITK_WS_FLOATS = 36 << 20        # CTCLIP_SPLITK_WS_FLOATS
_splitk = {}


def splitk_scratch():
    """The workspace behind `splitk_ws` of ctclip_gemm_bf16 (split-K partial products, summed in split order): one per
    (device, stream), like partials_scratch()."""
    key = (torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream)
    t = _splitk.get(key)
    if t is None:
        t = _splitk[key] = torch.empty(SPLITK_WS_FLOATS, dtype=torch.float32, device="cuda")
    return t


class _Hip:
    """Attribute access gives a checked wrapper: tensors -> device pointers, None -> NULL, the trailing
    `stream` argument defaults to torch's current stream (and a `partials` argument, or the `splitk_ws` /
    `splitk_ws_floats` pair, just before it to this stream's scratch buffers), a non-zero return raises RuntimeError."""

    def __init__(self):
        self._dll = None
        self._protos = None
        self._fns = {}
        self._timed = {}        # kernel name -> list of (start_event, end_event, work) while timing is on

    # -- optional per-launch HIP-event timing (bench.py roofline): events are recorded on the launch stream ----
    def time_kernel(self, name, work_fn):
        """Record a HIP event pair around every `name` launch.  work_fn(*args) -> algorithmic work of that launch: a number,
        or a dict of numbers plus an optional "tag" (launches are then also grouped by tag); None = do not time this one."""
        full = name if name.startswith("ctclip_") else "ctclip_" + name
        self._timed[full] = {"work_fn": work_fn, "events": []}
        self._fns.pop(name, None)
        self._fns.pop(full, None)

    def stop_timing(self):
        """-> {name: {"launches", "total_ms", "work", "items": [(ms, work), ...]}}; `work` is the sum of numeric work (or of
        each numeric field of dict work)."""
        out = {}
        for k, v in self._timed.items():
            torch.cuda.synchronize()
            items = [(a.elapsed_time(b), w) for a, b, w in v["events"]]
            if items and isinstance(items[0][1], dict):
                keys = [f for f, x in items[0][1].items() if isinstance(x, (int, float))]
                work = {f: float(sum(w[f] for _, w in items)) for f in keys}
            else:
                work = float(sum(w for _, w in items))
            out[k] = {"launches": len(items), "total_ms": float(sum(ms for ms, _ in items)), "work": work, "items": items}
        self._timed = {}
        self._fns = {}
        return out

    def _ensure(self):
        if self._dll is not None:
            return
        path = library_path()
        if not os.path.exists(path):
            raise HipLibraryMissing(
                f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the CT-CLIP hot path.")
        self._dll = ctypes.CDLL(path)
        self._protos = parse_header()

    def symbols(self):
        self._ensure()
        return sorted(self._protos)

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        self._ensure()
        if name in self._fns:
            return self._fns[name]
        full = name if name.startswith("ctclip_") else "ctclip_" + name
        if full not in self._protos:
            raise AttributeError(f"{full} is not declared in include/ctclip_hip.h")
        proto = self._protos[full]
        cfn = getattr(self._dll, full)
        cfn.restype = ctypes.c_int
        cfn.argtypes = [t for t, _, _ in proto]
        nargs = len(proto)
        has_stream = proto[-1][1] == "stream"
        has_partials = has_stream and nargs >= 2 and proto[-2][1] == "partials"
        has_splitk = has_stream and nargs >= 3 and proto[-3][1] == "splitk_ws" and proto[-2][1] == "splitk_ws_floats"

        # test hook CTCLIP_TEST_POISON_LDS=1: every launch is preceded by ctclip_probe_lds_fill with a NaN pattern -- LDS is not
        # cleared between kernels, so a kernel that reads a word it never wrote (idle threads of a block rounded up to whole waves,
        # halo columns) otherwise sees whatever the previous tenant left, which is benign until another process shares the device
        poison = has_stream and os.environ.get("CTCLIP_TEST_POISON_LDS") and full != "ctclip_probe_lds_fill"

        def call(*args):
            if poison:
                dev = torch.cuda.current_device()
                if dev not in _poison_sink:
                    _poison_sink[dev] = torch.zeros(4, dtype=torch.int32, device=f"cuda:{dev}")
                self.probe_lds_fill(0x7FC00000, _poison_sink[dev])
            if has_partials and len(args) == nargs - 2:
                args = (*args, partials_scratch())
            if has_splitk and len(args) == nargs - 3:
                args = (*args, splitk_scratch(), SPLITK_WS_FLOATS)
            if has_stream and len(args) == nargs - 1:
                args = (*args, torch.cuda.current_stream().cuda_stream)
            if len(args) != nargs:
                raise TypeError(f"{full} takes {nargs} arguments ({[n for _, n, _ in proto]}), got {len(args)}")
            conv = []
            for a, (_, an, ptr) in zip(args, proto):
                if ptr:
                    if a is None:
                        conv.append(None)
                    elif isinstance(a, torch.Tensor):
                        if not a.is_cuda:
                            raise RuntimeError(f"{full}: argument `{an}` is a CPU tensor; the HIP path has no CPU fallback")
                        conv.append(a.data_ptr())
                    else:
                        conv.append(int(a))
                else:
                    conv.append(a)
            timed = self._timed.get(full)
            work = timed["work_fn"](*args) if timed is not None else None
            if isinstance(work, dict):                       # which stream the launch goes to (the text tower has its own)
                work = dict(work, stream=str(torch.cuda.current_stream().cuda_stream))
            if work is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                err = cfn(*conv)
                e1.record()
                timed["events"].append((e0, e1, work))
            else:
                err = cfn(*conv)
            if err != 0:
                raise RuntimeError(f"{full} failed with hipError_t {err}")

        call.__name__ = full
        self._fns[name] = call
        return call


_poison_sink = {}
hip = _Hip()
