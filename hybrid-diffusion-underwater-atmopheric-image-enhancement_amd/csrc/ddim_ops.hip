// Elementwise kernels of the image-conditioned DDIM / ancestral sampler of the reference's second tree
// (diffusion/Diffusion.py:217-269, diffusion/Model.py:110-168, 446-515).  HBM-bound streaming kernels; compiled with
// -ffp-contract=off so that the update rounds like the reference's separate tensor ops.
#include "common.h"

using namespace hdiff;

#pragma clang fp contract(off)

namespace {

inline int grid_for(int64_t n, int per_thread = 1) {
  int64_t blocks = (n + (int64_t)256 * per_thread - 1) / ((int64_t)256 * per_thread);
  if (blocks > 256 * 32) blocks = 256 * 32;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

// Diffusion.py:259-263 with eta = 0:
//   y0 = (y - eps * sqrt(1 - at)) / sqrt(at) ;  y' = sqrt(at_next) * y0 + c2 * eps        (c1 * randn = 0 is dropped: x + 0 = x)
// tab[k] = {sqrt(1 - at), sqrt(at), sqrt(at_next), c2} of DDIM step k, formed on the host with the reference's fp32 ops.
// y and y_next may be the same buffer (the sampler updates in place): neither is __restrict__.
__global__ void ddim_step_kernel(const float* y, const float* __restrict__ eps, float* y_next,
                                 const float* __restrict__ tab, const int32_t* __restrict__ step_ptr, int nsteps,
                                 int32_t* __restrict__ nan_flag, int64_t n) {
  int k = *step_ptr;
  k = k < 0 ? 0 : (k >= nsteps ? nsteps - 1 : k);          // never index outside the table, whatever the counter holds
  const float s1m = tab[4 * k + 0], sa = tab[4 * k + 1], san = tab[4 * k + 2], c2 = tab[4 * k + 3];
  bool bad = false;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float e = eps[i];
    const float y0 = (y[i] - e * s1m) / sa;
    const float v = san * y0 + c2 * e;
    bad |= (v != v);
    y_next[i] = v;
  }
  if (__any(bad)) {
    if ((threadIdx.x & 63) == 0) atomicOr(nan_flag, 1);
  }
}

__global__ void fill_from_table_kernel(int64_t* dst, const int32_t* __restrict__ table, const int32_t* __restrict__ idx,
                                       int table_len, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  int k = *idx;
  k = k < 0 ? 0 : (k >= table_len ? table_len - 1 : k);
  if (i < n) dst[i] = (int64_t)table[k];
}

// F.interpolate(mode="nearest"): src = min(floor(dst * (float)in / out), in - 1), computed in float like ATen does
__global__ void resize_nearest_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int OH, int OW,
                                      float sy, float sx, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int ox = (int)(i % OW);
    const int64_t r = i / OW;
    const int oy = (int)(r % OH);
    const int64_t bc = r / OH;
    const int iy = min((int)floorf(oy * sy), H - 1), ix = min((int)floorf(ox * sx), W - 1);
    y[i] = x[(bc * H + iy) * W + ix];
  }
}

// torch.cat([a, b], dim=1) of per-sample blocks of n0 and n1 floats (the 3 + 3 channel sampler input, Diffusion.py:229,252:
// the head conv's two-pointer input needs 4-channel-aligned halves, so this one seam is materialised -- 24 B per pixel)
__global__ void concat2_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int64_t n0,
                               int64_t n1, int64_t total) {
  const int64_t per = n0 + n1;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t s = i / per, r = i - s * per;
    out[i] = r < n0 ? a[s * n0 + r] : b[s * n1 + (r - n0)];
  }
}

// nn.AdaptiveAvgPool2d((1, 1)): one wave per (sample, channel) plane
__global__ void avgpool_global_kernel(const float* __restrict__ x, float* __restrict__ y, int BC, int HW) {
  const int plane = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (plane >= BC) return;
  const float* p = x + (size_t)plane * HW;
  float s = 0.f;
  for (int i = threadIdx.x & 63; i < HW; i += 64) s += p[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) y[plane] = s / (float)HW;
}

}  // namespace

extern "C" {

int hdiff_ddim_step(const float* y, const float* eps, float* y_next, const float* tab, const int32_t* step_ptr, int nsteps,
                    int32_t* nan_flag, int64_t n, hdiff_stream_t stream) {
  HDIFF_CHECK_ARG(y && eps && y_next && tab && step_ptr && nan_flag && n > 0 && nsteps > 0, "ddim_step: bad arguments");
  (void)hipGetLastError();  // drop any stale error left by another HIP user in this thread
  hipLaunchKernelGGL(ddim_step_kernel, dim3(grid_for(n, 4)), dim3(256), 0, (hipStream_t)stream, y, eps, y_next, tab, step_ptr,
                     nsteps, nan_flag, n);
  HDIFF_CHECK_LAUNCH("ddim_step_kernel");
  return HDIFF_OK;
}

int hdiff_fill_from_table(int64_t* dst, const int32_t* table, const int32_t* idx, int table_len, int n,
                          hdiff_stream_t stream) {
  HDIFF_CHECK_ARG(dst && table && idx && n > 0 && table_len > 0, "fill_from_table: bad arguments");
  (void)hipGetLastError();
  hipLaunchKernelGGL(fill_from_table_kernel, dim3(cdiv(n, 64)), dim3(64), 0, (hipStream_t)stream, dst, table, idx, table_len,
                     n);
  HDIFF_CHECK_LAUNCH("fill_from_table_kernel");
  return HDIFF_OK;
}

int hdiff_resize_nearest(const float* x, float* y, int BC, int H, int W, int OH, int OW, hdiff_stream_t stream) {
  HDIFF_CHECK_ARG(x && y && BC > 0 && H > 0 && W > 0 && OH > 0 && OW > 0, "resize_nearest: bad arguments");
  (void)hipGetLastError();
  const int64_t n = (int64_t)BC * OH * OW;
  hipLaunchKernelGGL(resize_nearest_kernel, dim3(grid_for(n, 4)), dim3(256), 0, (hipStream_t)stream, x, y, H, W, OH, OW,
                     (float)H / (float)OH, (float)W / (float)OW, n);
  HDIFF_CHECK_LAUNCH("resize_nearest_kernel");
  return HDIFF_OK;
}

int hdiff_concat2(const float* a, const float* b, float* out, int B, int64_t n0, int64_t n1, hdiff_stream_t stream) {
  HDIFF_CHECK_ARG(a && b && out && B > 0 && n0 > 0 && n1 > 0, "concat2: bad arguments");
  (void)hipGetLastError();
  const int64_t total = (int64_t)B * (n0 + n1);
  hipLaunchKernelGGL(concat2_kernel, dim3(grid_for(total, 4)), dim3(256), 0, (hipStream_t)stream, a, b, out, n0, n1, total);
  HDIFF_CHECK_LAUNCH("concat2_kernel");
  return HDIFF_OK;
}

int hdiff_avgpool_global(const float* x, float* y, int BC, int HW, hdiff_stream_t stream) {
  HDIFF_CHECK_ARG(x && y && BC > 0 && HW > 0, "avgpool_global: bad arguments");
  (void)hipGetLastError();
  hipLaunchKernelGGL(avgpool_global_kernel, dim3(cdiv(BC, 4)), dim3(256), 0, (hipStream_t)stream, x, y, BC, HW);
  HDIFF_CHECK_LAUNCH("avgpool_global_kernel");
  return HDIFF_OK;
}

}  // extern "C"
