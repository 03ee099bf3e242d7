"""Yardstick: ctclip_gemm_bf16 against torch.matmul (hipBLASLt / rocBLAS) on the GEMM shapes of the CT-CLIP step, INTERLEAVED
in one process on one device on the same random operands (the guide's rules 24 / 25), median and best of R rounds.
Nothing in the product path calls torch.matmul.

    B=32 R=7 python3 tools/gemm_vs_vendor.py                  # table
    POWER=5 python3 tools/gemm_vs_vendor.py                   # + a POWER-second loop of each side with board power / sclk
                                                              #   sampled from sysfs (hwmon power1_average, pp_dpm_sclk)
"""
import glob, os, statistics, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ct-clip-ut_amd"))
import torch
from ctclip_hip.lib import hip
from ctclip_hip.ops import _splits_for

B = int(os.environ.get("B", 32))
R = int(os.environ.get("R", 7))
POWER = float(os.environ.get("POWER", 0))
T = 13824 * B
dev = "cuda"
SHAPES = [  # name, M, N, K, akm, bkm, c_fp32
    ("sq4096", 4096, 4096, 4096, 1, 1, 0),
    ("sq8192", 8192, 8192, 8192, 1, 1, 0),
    ("ff1 fwd", T, 2816, 512, 1, 1, 0),
    ("ff2 fwd", T, 512, 1408, 1, 1, 0),
    ("q fwd", T, 256, 512, 1, 1, 0),
    ("kv fwd", T, 512, 512, 1, 1, 0),
    ("out fwd", T, 512, 256, 1, 1, 0),
    ("ff2 dgrad", T, 1408, 512, 1, 1, 0),
    ("ff1 dgrad", T, 512, 2816, 1, 1, 0),
    ("ff1 wgrad", 2816, 512, T, 0, 0, 1),
    ("ff2 wgrad", 512, 1408, T, 0, 0, 1),
    ("kv wgrad", 512, 512, T, 0, 0, 1),
]
only = os.environ.get("ONLY")
if only:
    SHAPES = [s for s in SHAPES if any(o in s[0] for o in only.split(","))]


def timed(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def _sysfs():
    """hwmon power and pp_dpm_sclk of THIS process's GPU: the box shows all eight cards in sysfs, so the card is matched by the
    PCI address torch reports for device 0."""
    pr = torch.cuda.get_device_properties(0)
    want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}"
    for card in sorted(glob.glob("/sys/class/drm/card[0-9]*")):
        try:
            dev = os.path.realpath(os.path.join(card, "device"))
        except OSError:
            continue
        if want not in dev:
            continue
        pw = glob.glob(os.path.join(dev, "hwmon/hwmon*/power1_average")) + glob.glob(os.path.join(dev, "hwmon/hwmon*/power1_input"))
        cap = glob.glob(os.path.join(dev, "hwmon/hwmon*/power1_cap"))
        sc = os.path.join(dev, "pp_dpm_sclk")
        return (pw[0] if pw else None), (sc if os.path.exists(sc) else None), (cap[0] if cap else None), dev
    return None, None, None, None


def sample_power(stop, out):
    pw, sc, _, _ = _sysfs()
    while not stop.is_set():
        p = f = None
        try:
            if pw:
                p = int(open(pw).read()) / 1e6
            if sc:
                for ln in open(sc):
                    if "*" in ln:
                        f = int(ln.split(":")[1].strip().lower().replace("mhz", "").replace("*", ""))
        except Exception:
            pass
        out.append((p, f))
        time.sleep(0.1)


def power_loop(fn, seconds):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    one = timed(fn, 10) / 1e3
    n = max(10, int(seconds / one))
    stop, out = threading.Event(), []
    th = threading.Thread(target=sample_power, args=(stop, out))
    th.start()
    ms = timed(fn, n)
    stop.set()
    th.join()
    ps = [p for p, _ in out[len(out) // 4:] if p is not None]
    fs = [f for _, f in out[len(out) // 4:] if f is not None]
    return ms, (statistics.mean(ps) if ps else float("nan")), (max(ps) if ps else float("nan")), \
        (statistics.mean(fs) if fs else float("nan"))


if POWER > 0:
    _pw, _sc, _cap, _dev = _sysfs()
    capw = (int(open(_cap).read()) / 1e6) if _cap else float("nan")
    print(f"# power samples: {_pw} (cap {capw:.0f} W), clock: {_sc}", flush=True)
print(f"# {torch.cuda.get_device_name(0)}  B={B} (tokens {T})  rounds {R}  torch {torch.__version__}  randn operands", flush=True)
print(f"# {'shape':12s} {'M':>8s} {'N':>5s} {'K':>8s} | ours TF med  best | vendor TF med  best | ours/vendor", flush=True)
for name, M, N, K, akm, bkm, cf in SHAPES:
    A = torch.randn((M, K) if akm else (K, M), device=dev).to(torch.bfloat16)
    W = torch.randn((N, K) if bkm else (K, N), device=dev).to(torch.bfloat16)
    C = torch.zeros(M, N, device=dev, dtype=torch.float32 if cf else torch.bfloat16)
    split = 1 if akm else _splits_for(M, N, K)
    acc = 0 if akm else 1

    def ours():
        hip.gemm_bf16(A, W, C, None, None, M, N, K, A.stride(0), W.stride(0), N, 0, akm, bkm, cf, split, acc, 1.0, 0)
    if akm:
        Wt = W.t()
        Cv = torch.empty(M, N, device=dev, dtype=torch.bfloat16)

        def vendor():
            torch.matmul(A, Wt, out=Cv)
    else:
        At = A.t()
        Cv = torch.empty(M, N, device=dev, dtype=torch.bfloat16)

        def vendor():
            torch.matmul(At, W, out=Cv)
    fl = 2.0 * M * N * K
    n = max(3, min(50, int(0.03 / (fl / 1.0e15))))
    for _ in range(3):
        ours(); vendor()
    to, tv = [], []
    for r in range(R):
        to.append(timed(ours, n))
        tv.append(timed(vendor, n))
    tf = lambda ms: fl / ms / 1e9
    mo, mv = statistics.median(to), statistics.median(tv)
    print(f"  {name:12s} {M:8d} {N:5d} {K:8d} | {tf(mo):8.0f} {tf(min(to)):6.0f}   | {tf(mv):8.0f} {tf(min(tv)):6.0f}     | {mv / mo:5.2f}", flush=True)
    if POWER > 0 and name in ("sq4096", "sq8192", "ff1 dgrad", "ff2 fwd", "ff1 wgrad"):
        for who, fn in (("ours", ours), ("vendor", vendor)):
            ms, pm, px, fm = power_loop(fn, POWER)
            print(f"      power {who:6s}: {tf(ms):6.0f} TF over {POWER:.0f} s   board {pm:6.0f} W mean {px:6.0f} W max   sclk(sysfs) {fm:5.0f} MHz", flush=True)
    del A, W, C, Cv
