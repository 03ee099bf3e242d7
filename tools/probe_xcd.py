"""Which XCD does block b of a 1-D grid run on, and in which order does an XCD start its blocks?  (The XCD-aware block maps of
csrc/common.h: xcd_remap and the VQ code-group map assume XCD = b % 8 and in-order dispatch inside an XCD.)
Compiles a 30-line kernel with hipcc on the GPU box; every block records HW_REG_XCC_ID, its CU and its start time, then spins
for ~20 us so that a round of one-per-CU workgroups (128 KiB of LDS each) stays resident."""
import ctypes, os, subprocess, sys, tempfile
import torch

SRC = r'''
#include <hip/hip_runtime.h>
extern "C" __global__ __launch_bounds__(512) void probe(unsigned long long* out, int spin) {
  extern __shared__ char smem[];
  unsigned xcc, hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  const unsigned long long t0 = wall_clock64();
  if (threadIdx.x == 0) {
    out[blockIdx.x * 3 + 0] = xcc & 0xf;
    out[blockIdx.x * 3 + 1] = hwid;
    out[blockIdx.x * 3 + 2] = t0;
    smem[0] = 1;
  }
  while (wall_clock64() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(8);
}
extern "C" int launch(void* out, int blocks, int lds, int spin) {
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL(probe, dim3(blocks), dim3(512), lds, 0, (unsigned long long*)out, spin);
  return (int)hipDeviceSynchronize();
}
'''
d = tempfile.mkdtemp()
open(os.path.join(d, "p.hip"), "w").write(SRC)
so = os.path.join(d, "p.so")
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(d, "p.hip")])
lib = ctypes.CDLL(so)
blocks = int(os.environ.get("BLOCKS", 1024))
out = torch.zeros(blocks, 3, dtype=torch.int64, device="cuda")
rc = lib.launch(ctypes.c_void_p(out.data_ptr()), blocks, 128 * 1024, 2000)      # wall clock is 100 MHz: 2000 ticks = 20 us
assert rc == 0, rc
o = out.cpu()
xcc, hw, t = o[:, 0], o[:, 1], o[:, 2] - o[:, 2].min()
b = torch.arange(blocks)
print(f"{blocks} blocks of 512 threads, 128 KiB LDS; XCD == block % 8 for {(xcc == b % 8).float().mean():.4f} of the blocks")
print("XCD of blocks 0..31:", xcc[:32].tolist())
cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5)
for x in range(2):
    mine = b[xcc == x]
    order = mine[torch.argsort(t[mine], stable=True)]
    print(f"XCD {x}: {len(mine)} blocks on {len(set(cu[mine].tolist()))} distinct (se, sh, cu); XCD-local index (block >> 3) in start order:")
    print("   ", (order >> 3).tolist()[:80])
    print("    start time (us) of those:", [round(float(v) / 100, 1) for v in t[order][:80].tolist()])
