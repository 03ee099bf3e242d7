// Volume ingest: the tensor part of reference src/utils/preprocess.py:84-152 (`process_file`,
// model_type "ctclip") as ONE pass on the device:
//   raw scan [H, W, D] (nibabel order, f32 or i16)  ->  HU = slope * raw + intercept  ->  permute to [D, H, W]
//   -> trilinear resample to the target spacing (F.interpolate(size=..., align_corners=False), :34-36)
//   -> clamp [-1000, 1000] / 1000 (:135-136)  ->  centre crop / symmetric pad with -1 to (D_t, H_t, W_t) (:38-80,143-145)
//   -> [1, D_t, H_t, W_t] in bf16 (what the patch embedding reads) or f32.
// The reference materialises five full-size f32 intermediates on the host and ships 221 MB to the device; here each output
// voxel is produced from its 8 source taps directly.  The raw scan is D-fastest and the output W-fastest, so a workgroup
// owns a 32 (d) x 32 (w) tile of one output row h: lanes run along d for the (nearly contiguous) gathers and the tile is
// turned through LDS for 64-byte row stores.
#include "common.h"

namespace {

struct IngestArgs {
  const void* raw; void* out;
  int H, W, D;            // raw extents (nibabel order [H][W][D])
  int rD, rH, rW;         // resampled extents
  int oD, oH, oW;         // output extents
  int offD, offH, offW;   // resampled index = output index + off (crop: +start, pad: -pad_before)
  float slope, intercept, pad_value;
  float sD, sH, sW;       // in / out scale per axis (align_corners = False)
};

__device__ __forceinline__ void src_index(int dst, float scale, int in_size, int& i0, int& i1, float& l1) {
  float s = scale * ((float)dst + 0.5f) - 0.5f;          // PyTorch area_pixel_compute_source_index, align_corners=False
  s = s < 0.f ? 0.f : s;
  i0 = (int)s;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l1 = s - (float)i0;
}

template <typename RAW> __device__ __forceinline__ float raw_at(const void* p, long i);
template <> __device__ __forceinline__ float raw_at<float>(const void* p, long i) { return ((const float*)p)[i]; }
template <> __device__ __forceinline__ float raw_at<short>(const void* p, long i) { return (float)((const short*)p)[i]; }

template <typename RAW, bool OUT16>
__global__ __launch_bounds__(256) void ingest_kernel(IngestArgs a) {
  __shared__ float tile[32][33];                          // [w][d]
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int od0 = blockIdx.x * 32, ow0 = blockIdx.y * 32, oh = blockIdx.z;
  const int od = od0 + tx;
  const int rd = od + a.offD, rh = oh + a.offH;
  const bool row_ok = od < a.oD && rd >= 0 && rd < a.rD && rh >= 0 && rh < a.rH;
  int d0 = 0, d1 = 0, h0 = 0, h1 = 0;
  float ld = 0.f, lh = 0.f;
  if (row_ok) {
    src_index(rd, a.sD, a.D, d0, d1, ld);
    src_index(rh, a.sH, a.H, h0, h1, lh);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int wl = ty + 8 * k, ow = ow0 + wl, rw = ow + a.offW;
    float v = a.pad_value;
    if (row_ok && ow < a.oW && rw >= 0 && rw < a.rW) {
      int w0, w1;
      float lw;
      src_index(rw, a.sW, a.W, w0, w1, lw);
      // raw[h][w][d]; HU transform on every tap first, as the reference does before interpolating (:124-125)
      auto tap = [&](int h, int w, int d) { return fmaf(a.slope, raw_at<RAW>(a.raw, ((long)h * a.W + w) * a.D + d), a.intercept); };
      const float c000 = tap(h0, w0, d0), c001 = tap(h0, w1, d0), c010 = tap(h1, w0, d0), c011 = tap(h1, w1, d0);
      const float c100 = tap(h0, w0, d1), c101 = tap(h0, w1, d1), c110 = tap(h1, w0, d1), c111 = tap(h1, w1, d1);
      const float l0d = 1.f - ld, l0h = 1.f - lh, l0w = 1.f - lw;
      // upsample_trilinear3d: d outermost, then h, then w
      v = l0d * (l0h * (l0w * c000 + lw * c001) + lh * (l0w * c010 + lw * c011)) +
          ld * (l0h * (l0w * c100 + lw * c101) + lh * (l0w * c110 + lw * c111));
      v = fminf(fmaxf(v, -1000.f), 1000.f) / 1000.0f;
    }
    tile[wl][tx] = v;
  }
  __syncthreads();
  // write: lanes along w
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int dl = ty + 8 * k, d = od0 + dl, ow = ow0 + tx;
    if (d < a.oD && ow < a.oW) {
      const long o = ((long)d * a.oH + oh) * a.oW + ow;
      const float v = tile[tx][dl];
      if (OUT16) ((bf16_t*)a.out)[o] = f32_to_bf16(v);
      else ((float*)a.out)[o] = v;
    }
  }
}

}  // namespace

extern "C" int ctclip_ingest_volume(const void* raw, int raw_is_i16, int H, int W, int D, float slope, float intercept,
                                    int rD, int rH, int rW, int oD, int oH, int oW, float pad_value, void* out, int out_bf16,
                                    void* stream) {
  if (H <= 0 || W <= 0 || D <= 0 || rD <= 0 || rH <= 0 || rW <= 0 || oD <= 0 || oH <= 0 || oW <= 0) return (int)hipErrorInvalidValue;
  if (oH > 65535 || (oW + 31) / 32 > 65535) return (int)hipErrorInvalidValue;
  IngestArgs a{};
  a.raw = raw; a.out = out; a.H = H; a.W = W; a.D = D; a.rD = rD; a.rH = rH; a.rW = rW; a.oD = oD; a.oH = oH; a.oW = oW;
  auto off = [](int r, int o) { return r > o ? (r - o) / 2 : -((o - r) / 2); };   // preprocess.py:62,72-73
  a.offD = off(rD, oD); a.offH = off(rH, oH); a.offW = off(rW, oW);
  a.slope = slope; a.intercept = intercept; a.pad_value = pad_value;
  a.sD = (float)D / (float)rD; a.sH = (float)H / (float)rH; a.sW = (float)W / (float)rW;
  dim3 grid((oD + 31) / 32, (oW + 31) / 32, oH), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (raw_is_i16) {
    if (out_bf16) hipLaunchKernelGGL((ingest_kernel<short, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((ingest_kernel<short, false>), grid, block, 0, st, a);
  } else {
    if (out_bf16) hipLaunchKernelGGL((ingest_kernel<float, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((ingest_kernel<float, false>), grid, block, 0, st, a);
  }
  CTCLIP_CHECK_LAUNCH();
}
