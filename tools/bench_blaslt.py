"""Yardstick only: what torch.matmul (hipBLASLt / rocBLAS) reaches on the GEMM shapes of the CT-CLIP step.  Nothing in the
product path calls it; it shows how far the hand-written kernels are from the vendor library on the same box."""
import os, torch
B = int(os.environ.get("B", 32))
T = 13824 * B
dev = "cuda"
def t(fn, n=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for name, M, N, K in (("ff1 fwd", T, 2816, 512), ("ff2 fwd", T, 512, 1408), ("kv fwd", T, 512, 512), ("q fwd", T, 256, 512),
                      ("ff1 dgrad", T, 512, 2816), ("sq4096", 4096, 4096, 4096), ("sq8192", 8192, 8192, 8192)):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = torch.randn(N, K, device=dev).to(torch.bfloat16)
    ms = t(lambda: torch.matmul(a, w.t()))
    print(f"{name:12s} NT  {ms*1e3:9.1f} us {2.0*M*N*K/ms/1e9:8.1f} TFLOP/s", flush=True)
    del a, w
for name, M, N in (("ff1 wgrad", 2816, 512), ("ff2 wgrad", 512, 1408), ("kv wgrad", 512, 512)):
    dy = torch.randn(T, M, device=dev).to(torch.bfloat16)
    x = torch.randn(T, N, device=dev).to(torch.bfloat16)
    ms = t(lambda: torch.matmul(dy.t(), x))
    print(f"{name:12s} TN  {ms*1e3:9.1f} us {2.0*M*N*T/ms/1e9:8.1f} TFLOP/s", flush=True)
    del dy, x
