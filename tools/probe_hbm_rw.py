"""Pure-read, pure-write and copy rates of this box's HBM (torch elementwise kernels over 4 GiB), for reading the epilogue numbers."""
import torch
n = 1 << 30
x = torch.empty(n, dtype=torch.float32, device="cuda")
y = torch.empty(n, dtype=torch.float32, device="cuda")
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
gb = n * 4 / 1e9
print(f"fill  (write only) {gb / t(lambda: x.zero_()):7.2f} TB/s")
print(f"sum   (read only)  {gb / t(lambda: x.sum()):7.2f} TB/s")
print(f"copy  (read+write) {2 * gb / t(lambda: y.copy_(x)):7.2f} TB/s")
print(f"add_  (read+write same buffer) {2 * gb / t(lambda: x.add_(1.0)):7.2f} TB/s")
