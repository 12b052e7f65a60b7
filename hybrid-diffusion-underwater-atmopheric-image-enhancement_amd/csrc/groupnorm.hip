// GroupNorm(32, C) statistics over NCHW activations (reference: nn.GroupNorm at ModelCondition.py:170,184,249).
//
// HBM-bound streaming reduction.  The normalisation + affine + Swish is NOT applied here: hdiff_gn_finalize folds
// (mean, rstd, gamma, beta) into a per-(sample, channel) scale/shift that the consuming convolution applies while it
// stages its input patch (conv_igemm.hip), so the activation is read once for the statistics and never rewritten.
//
// The input may be the virtual channel concat of two tensors (skip connections, ModelCondition.py:271); a group may
// straddle the seam (384 channels / 32 groups = 12 per group, seam at 256), so sources are chosen per channel.
// Each (sample, group) is split over `nsplit` workgroups; every workgroup produces (count, mean, M2) of its slice in one
// pass over shifted data, and the partials are merged with Chan's formula.
#include "common.h"

using namespace hdiff;

namespace {

constexpr int GN_THREADS = 512;

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < GN_THREADS / 64; ++i) t += red[i];
  return t;
}

// Merge the partials of one (sample, group) and write the per-channel scale / shift; every lane does the same arithmetic
// (Chan merge, sequential over the few partials), lanes share the channels.
__device__ __forceinline__ void gn_finalize_group(const float* __restrict__ ws, int bg, int C, int G, int nsplit,
                                                  const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                  float* __restrict__ scale, float* __restrict__ shift,
                                                  float* __restrict__ mean_out, float* __restrict__ rstd_out, int lane) {
  const int b = bg / G, g = bg - b * G;
  const int cpg = C / G;
  float n = 0.f, mean = 0.f, m2 = 0.f;
  for (int s = 0; s < nsplit; ++s) {
    const float* p = ws + ((size_t)bg * nsplit + s) * 3;
    const float nb = p[0], mb = p[1], m2b = p[2];
    if (nb > 0.f) {
      const float tot = n + nb;
      const float delta = mb - mean;
      mean += delta * (nb / tot);
      m2 += m2b + delta * delta * (n * nb / tot);
      n = tot;
    }
  }
  const float var = m2 / n;
  const float rstd = rsqrtf(var + eps);
  if (lane == 0) {
    if (mean_out) mean_out[bg] = mean;
    if (rstd_out) rstd_out[bg] = rstd;
  }
  for (int cc = lane; cc < cpg; cc += 64) {
    const int c = g * cpg + cc;
    const float sc = rstd * gamma[c];
    scale[b * C + c] = sc;
    shift[b * C + c] = beta[c] - mean * sc;
  }
}

__global__ __launch_bounds__(GN_THREADS) void gn_stats_kernel(const float* __restrict__ x0, const float* __restrict__ x1,
                                                              int C0, int C1, int HW, int G, int nsplit,
                                                              float* __restrict__ ws, int fused,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              float eps, float* __restrict__ scale, float* __restrict__ shift) {
  __shared__ float red[GN_THREADS / 64];
  const int bg = blockIdx.x, split = blockIdx.y;
  const int b = bg / G, g = bg - b * G;
  const int C = C0 + C1, cpg = C / G;
  // slice of every channel plane handled by this split, in float4 units when HW % 4 == 0
  const bool vec = (HW & 3) == 0;
  const int units = vec ? HW >> 2 : HW;
  const int per = (units + nsplit - 1) / nsplit;
  const int lo = split * per, hi = min(units, lo + per);
  const int cnt_units = max(0, hi - lo);
  const float count = (float)cnt_units * (vec ? 4.f : 1.f) * (float)cpg;

  // ONE pass: sums of d = x - K and of d^2 with K = the slice's first element (a value near the mean keeps the
  // subtraction mean^2 - mean(x^2)-style cancellation out of fp32: d is O(std), so M2 = S2 - S1^2/n loses nothing)
  float K = 0.f;
  if (cnt_units > 0) {
    const int c = g * cpg;
    const float* plane = (c < C0) ? x0 + ((size_t)b * C0 + c) * HW : x1 + ((size_t)b * C1 + (c - C0)) * HW;
    K = plane[vec ? 4 * lo : lo];
  }
  float s1 = 0.f, s2 = 0.f;
  for (int cc = 0; cc < cpg; ++cc) {
    const int c = g * cpg + cc;
    const float* plane = (c < C0) ? x0 + ((size_t)b * C0 + c) * HW : x1 + ((size_t)b * C1 + (c - C0)) * HW;
    if (vec) {
      const float4* p4 = reinterpret_cast<const float4*>(plane);
      for (int i = lo + threadIdx.x; i < hi; i += GN_THREADS) {
        const float4 v = p4[i];
        const float a = v.x - K, bb = v.y - K, cq = v.z - K, d = v.w - K;
        s1 += (a + bb) + (cq + d);
        s2 += (a * a + bb * bb) + (cq * cq + d * d);
      }
    } else {
      for (int i = lo + threadIdx.x; i < hi; i += GN_THREADS) {
        const float a = plane[i] - K;
        s1 += a;
        s2 += a * a;
      }
    }
  }
  s1 = block_sum(s1, red);
  s2 = block_sum(s2, red);
  const float mean = (count > 0.f) ? K + s1 / count : 0.f;
  const float m2 = (count > 0.f) ? fmaxf(s2 - s1 * s1 / count, 0.f) : 0.f;
  if (threadIdx.x == 0) {
    float* o = ws + ((size_t)bg * nsplit + split) * 3;
    o[0] = count;
    o[1] = mean;
    o[2] = m2;
  }
  // Fused finalize (hdiff_gn_scale_shift with nsplit == 1): this workgroup holds the only partial of its (sample, group),
  // so it folds (mean, rstd, gamma, beta) into scale / shift itself -- no second launch, no hand-off between workgroups.
  // (A last-arriving-workgroup hand-off for nsplit > 1 was measured and dropped: its agent-scope release writes back the
  // XCD's whole dirty L2 -- the activation the producing conv has just written -- at every workgroup's end: the 0.11 ms
  // statistics pass of a 537 MB tensor took 0.46 ms.)
  if (fused) {
    __syncthreads();            // thread 0's partial is visible to the first wave
    if (threadIdx.x < 64)
      gn_finalize_group(ws, bg, C, G, 1, gamma, beta, eps, scale, shift, nullptr, nullptr, threadIdx.x);
  }
}

// One wave per (sample, group).
__global__ void gn_finalize_kernel(const float* __restrict__ ws, int B, int C, int G, int nsplit,
                                   const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                   float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ mean_out,
                                   float* __restrict__ rstd_out) {
  const int bg = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (bg >= B * G) return;
  gn_finalize_group(ws, bg, C, G, nsplit, gamma, beta, eps, scale, shift, mean_out, rstd_out, threadIdx.x & 63);
}

// y = x * scale[b][c] + shift[b][c]: GroupNorm WITHOUT the Swish (the reference's AttnBlock normalises and projects,
// ModelCondition.py:103-107; every other GroupNorm of the path is followed by Swish and fused into a conv prologue)
__global__ void gn_affine_apply_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                       const float* __restrict__ shift, float* __restrict__ y, int HW) {
  const int bc = blockIdx.x;
  const float sc = scale[bc], sh = shift[bc];
  const float* xp = x + (size_t)bc * HW;
  float* yp = y + (size_t)bc * HW;
  for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < HW; i += gridDim.y * blockDim.x) yp[i] = fmaf(xp[i], sc, sh);
}

__global__ void gn_swish_apply_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                      const float* __restrict__ shift, float* __restrict__ y, int HW) {
  const int bc = blockIdx.x;   // plane index on x: B*C may exceed 65 535
  const float sc = scale[bc], sh = shift[bc];
  const float* xp = x + (size_t)bc * HW;
  float* yp = y + (size_t)bc * HW;
  for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < HW; i += gridDim.y * blockDim.x)
    yp[i] = swishf(fmaf(xp[i], sc, sh));
}

}  // namespace

extern "C" int hdiff_gn_stats(const float* x0, const float* x1, int C0, int C1, int B, int HW, int G, int nsplit,
                              float* ws, hdiff_stream_t stream) {
  HDIFF_CHECK_ARG(x0 && ws, "gn_stats: null pointer");
  HDIFF_CHECK_ARG(C1 == 0 || x1, "gn_stats: C1 > 0 without x1");
  HDIFF_CHECK_ARG(G > 0 && (C0 + C1) % G == 0, "gn_stats: channels %d not divisible by %d groups", C0 + C1, G);
  HDIFF_CHECK_ARG(B > 0 && HW > 0 && nsplit >= 1 && nsplit <= 1024, "gn_stats: bad sizes");
  (void)hipGetLastError();  // drop any stale error left by another HIP user in this thread
  hipLaunchKernelGGL(gn_stats_kernel, dim3(B * G, nsplit), dim3(GN_THREADS), 0, (hipStream_t)stream, x0, x1, C0, C1, HW,
                     G, nsplit, ws, 0, (const float*)nullptr, (const float*)nullptr, 0.f, (float*)nullptr, (float*)nullptr);
  HDIFF_CHECK_LAUNCH("gn_stats_kernel");
  return HDIFF_OK;
}

extern "C" int hdiff_gn_scale_shift(const float* x0, const float* x1, int C0, int C1, int B, int HW, int G, int nsplit,
                                    float* ws, const float* gamma, const float* beta, float eps, float* scale, float* shift,
                                    hdiff_stream_t stream) {
  HDIFF_CHECK_ARG(x0 && ws && gamma && beta && scale && shift, "gn_scale_shift: null pointer");
  HDIFF_CHECK_ARG(C1 == 0 || x1, "gn_scale_shift: C1 > 0 without x1");
  HDIFF_CHECK_ARG(G > 0 && (C0 + C1) % G == 0, "gn_scale_shift: channels %d not divisible by %d groups", C0 + C1, G);
  HDIFF_CHECK_ARG(B > 0 && HW > 0 && nsplit >= 1 && nsplit <= 1024, "gn_scale_shift: bad sizes");
  (void)hipGetLastError();  // drop any stale error left by another HIP user in this thread
  hipLaunchKernelGGL(gn_stats_kernel, dim3(B * G, nsplit), dim3(GN_THREADS), 0, (hipStream_t)stream, x0, x1, C0, C1, HW,
                     G, nsplit, ws, nsplit == 1 ? 1 : 0, gamma, beta, eps, scale, shift);
  HDIFF_CHECK_LAUNCH("gn_stats_kernel");
  if (nsplit > 1) {            // several partials per (sample, group): the small merge kernel behind the streaming pass
    const int waves_per_block = 4;
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(cdiv(B * G, waves_per_block)), dim3(64 * waves_per_block), 0,
                       (hipStream_t)stream, ws, B, C0 + C1, G, nsplit, gamma, beta, eps, scale, shift, (float*)nullptr,
                       (float*)nullptr);
    HDIFF_CHECK_LAUNCH("gn_finalize_kernel");
  }
  return HDIFF_OK;
}

extern "C" int hdiff_gn_finalize(const float* ws, int B, int C, int G, int nsplit, const float* gamma, const float* beta,
                                 float eps, float* scale, float* shift, float* mean_out, float* rstd_out,
                                 hdiff_stream_t stream) {
  HDIFF_CHECK_ARG(ws && gamma && beta && scale && shift, "gn_finalize: null pointer");
  HDIFF_CHECK_ARG(G > 0 && C % G == 0, "gn_finalize: channels %d not divisible by %d groups", C, G);
  const int waves_per_block = 4;
  (void)hipGetLastError();  // drop any stale error left by another HIP user in this thread
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(cdiv(B * G, waves_per_block)), dim3(64 * waves_per_block), 0,
                     (hipStream_t)stream, ws, B, C, G, nsplit, gamma, beta, eps, scale, shift, mean_out, rstd_out);
  HDIFF_CHECK_LAUNCH("gn_finalize_kernel");
  return HDIFF_OK;
}

extern "C" int hdiff_gn_swish_apply(const float* x, const float* scale, const float* shift, float* y, int B, int C,
                                    int HW, hdiff_stream_t stream) {
  HDIFF_CHECK_ARG(x && scale && shift && y, "gn_swish_apply: null pointer");
  const int bx = cdiv(HW, 256) < 64 ? cdiv(HW, 256) : 64;
  (void)hipGetLastError();  // drop any stale error left by another HIP user in this thread
  hipLaunchKernelGGL(gn_swish_apply_kernel, dim3(B * C, bx), dim3(256), 0, (hipStream_t)stream, x, scale, shift, y, HW);
  HDIFF_CHECK_LAUNCH("gn_swish_apply_kernel");
  return HDIFF_OK;
}

extern "C" int hdiff_gn_affine_apply(const float* x, const float* scale, const float* shift, float* y, int B, int C, int HW,
                                     hdiff_stream_t stream) {
  HDIFF_CHECK_ARG(x && scale && shift && y, "gn_affine_apply: null pointer");
  const int bx = cdiv(HW, 256) < 64 ? cdiv(HW, 256) : 64;
  (void)hipGetLastError();  // drop any stale error left by another HIP user in this thread
  hipLaunchKernelGGL(gn_affine_apply_kernel, dim3(B * C, bx), dim3(256), 0, (hipStream_t)stream, x, scale, shift, y, HW);
  HDIFF_CHECK_LAUNCH("gn_affine_apply_kernel");
  return HDIFF_OK;
}
