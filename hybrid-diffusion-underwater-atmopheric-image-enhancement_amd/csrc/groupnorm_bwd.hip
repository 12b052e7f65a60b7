// Backward of GroupNorm(32, C) + Swish as fused into the convolution prologue (reference: autograd of
// nn.GroupNorm -> Swish, ModelCondition.py:170-171, 184-185, 249-250).  HBM-bound: two streaming passes.
//   y = (x - mean) * rstd * gamma + beta,  a = y * sigmoid(y),  dy = dA * sigmoid(y) * (1 + y * (1 - sigmoid(y)))
//   dgamma[c] = sum dy * xhat,  dbeta[c] = sum dy
//   dx = rstd * (gamma * dy - mean_g(gamma * dy) - xhat * mean_g(gamma * dy * xhat))      (means over the group)
// pass 1: per (sample, channel) plane  p1 = sum dy, p2 = sum dy * xhat
// pass 2: tiny -- group sums s1, s2 and the parameter gradients
// pass 3: elementwise dx (dy recomputed), written to the two halves of the virtual concat input
#include "common.h"

using namespace hdiff;

namespace {

constexpr int T = 256;

__device__ __forceinline__ float block_sum256(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

__device__ __forceinline__ float dswish_times(float da, float y) {
  const float sg = 1.0f / (1.0f + __expf(-y));
  return da * sg * (1.0f + y * (1.0f - sg));
}

template <bool SWISH>
__global__ __launch_bounds__(T) void gn_bwd_reduce_kernel(const float* __restrict__ x0, const float* __restrict__ x1, int C0,
                                                          int C1, int HW, int G, const float* __restrict__ dA,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ p1, float* __restrict__ p2) {
  __shared__ float red[4];
  const int C = C0 + C1, cpg = C / G;
  const int bc = blockIdx.x, b = bc / C, c = bc - b * C;
  const int g = c / cpg;
  const float mu = mean[b * G + g], rs = rstd[b * G + g], ga = gamma[c], be = beta[c];
  const float* xp = (c < C0) ? x0 + ((size_t)b * C0 + c) * HW : x1 + ((size_t)b * C1 + (c - C0)) * HW;
  const float* dp = dA + (size_t)bc * HW;
  float s1 = 0.f, s2 = 0.f;
  for (int i = threadIdx.x; i < HW; i += T) {
    const float xh = (xp[i] - mu) * rs;
    const float dy = SWISH ? dswish_times(dp[i], fmaf(xh, ga, be)) : dp[i];
    s1 += dy;
    s2 = fmaf(dy, xh, s2);
  }
  s1 = block_sum256(s1, red);
  s2 = block_sum256(s2, red);
  if (threadIdx.x == 0) {
    p1[bc] = s1;
    p2[bc] = s2;
  }
}

// one thread per channel for the parameter gradients, one per (b, g) for the group sums
__global__ void gn_bwd_finalize_kernel(const float* __restrict__ p1, const float* __restrict__ p2,
                                       const float* __restrict__ gamma, int B, int C, int G, float* __restrict__ gs1,
                                       float* __restrict__ gs2, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int cpg = C / G;
  if (i < C) {
    float a = 0.f, bsum = 0.f;
    for (int b = 0; b < B; ++b) {
      a += p2[b * C + i];
      bsum += p1[b * C + i];
    }
    dgamma[i] = a;
    dbeta[i] = bsum;
  }
  if (i < B * G) {
    const int b = i / G, g = i - b * G;
    float s1 = 0.f, s2 = 0.f;
    for (int cc = 0; cc < cpg; ++cc) {
      const int c = g * cpg + cc;
      s1 = fmaf(gamma[c], p1[b * C + c], s1);
      s2 = fmaf(gamma[c], p2[b * C + c], s2);
    }
    gs1[i] = s1;
    gs2[i] = s2;
  }
}

template <bool SWISH>
__global__ __launch_bounds__(T) void gn_bwd_apply_kernel(const float* __restrict__ x0, const float* __restrict__ x1, int C0,
                                                         int C1, int HW, int G, const float* __restrict__ dA,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const float* __restrict__ gs1, const float* __restrict__ gs2,
                                                         float* __restrict__ dx0, float* __restrict__ dx1) {
  const int C = C0 + C1, cpg = C / G;
  const int bc = blockIdx.x, b = bc / C, c = bc - b * C;   // plane index on x: B*C may exceed 65 535
  const int g = c / cpg;
  const float mu = mean[b * G + g], rs = rstd[b * G + g], ga = gamma[c], be = beta[c];
  const float inv_n = 1.0f / ((float)cpg * (float)HW);
  const float m1 = gs1[b * G + g] * inv_n, m2 = gs2[b * G + g] * inv_n;
  const float* xp = (c < C0) ? x0 + ((size_t)b * C0 + c) * HW : x1 + ((size_t)b * C1 + (c - C0)) * HW;
  float* op = (c < C0) ? dx0 + ((size_t)b * C0 + c) * HW : dx1 + ((size_t)b * C1 + (c - C0)) * HW;
  const float* dp = dA + (size_t)bc * HW;
  for (int i = blockIdx.y * T + threadIdx.x; i < HW; i += gridDim.y * T) {
    const float xh = (xp[i] - mu) * rs;
    const float dy = SWISH ? dswish_times(dp[i], fmaf(xh, ga, be)) : dp[i];
    op[i] = rs * (ga * dy - m1 - xh * m2);
  }
}

// dvec[b][c] = sum_hw dy[b][c][:]  and  dbias[c] = sum_b dvec[b][c]   (one block per channel, samples in order)
__global__ __launch_bounds__(T) void bias_addvec_grad_kernel(const float* __restrict__ dy, int B, int C, int HW,
                                                             float* __restrict__ dvec, float* __restrict__ dbias) {
  __shared__ float red[4];
  const int c = blockIdx.x;
  float total = 0.f;
  for (int b = 0; b < B; ++b) {
    const float* p = dy + ((size_t)b * C + c) * HW;
    float s = 0.f;
    for (int i = threadIdx.x; i < HW; i += T) s += p[i];
    s = block_sum256(s, red);
    if (threadIdx.x == 0 && dvec) dvec[b * C + c] = s;
    total += s;
  }
  if (threadIdx.x == 0 && dbias) dbias[c] = total;
}

// The same sums with one workgroup per (channel, sample) plane and 16-byte loads (the single-kernel form above walks the
// batch inside C workgroups: 1.2 TB/s at C = 128); dbias is then a fixed-order sum over the batch of the per-sample sums.
__global__ __launch_bounds__(T) void plane_sum_kernel(const float* __restrict__ dy, int C, int HW, float* __restrict__ dvec) {
  __shared__ float red[4];
  const int c = blockIdx.x, b = blockIdx.y;
  const float* p = dy + ((size_t)b * C + c) * HW;
  float s = 0.f;
  if ((HW & 3) == 0) {
    const float4* p4 = reinterpret_cast<const float4*>(p);
    for (int i = threadIdx.x; i < (HW >> 2); i += T) {
      const float4 v = p4[i];
      s += (v.x + v.y) + (v.z + v.w);
    }
  } else {
    for (int i = threadIdx.x; i < HW; i += T) s += p[i];
  }
  s = block_sum256(s, red);
  if (threadIdx.x == 0) dvec[b * C + c] = s;
}

__global__ void batch_sum_kernel(const float* __restrict__ dvec, int B, int C, float* __restrict__ dbias) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float t = 0.f;
  for (int b = 0; b < B; ++b) t += dvec[b * C + c];
  dbias[c] = t;
}

}  // namespace

namespace {
template <bool SWISH>
int gn_bwd_launch(const float* x0, const float* x1, int C0, int C1, int B, int HW, int G, const float* dA, const float* mean,
                  const float* rstd, const float* gamma, const float* beta, float* ws, float* dx0, float* dx1, float* dgamma,
                  float* dbeta, hipStream_t s) {
  const int C = C0 + C1;
  float* p1 = ws;                 // [B][C]
  float* p2 = ws + (size_t)B * C; // [B][C]
  float* gs1 = p2 + (size_t)B * C; // [B][G]
  float* gs2 = gs1 + (size_t)B * G;
  (void)hipGetLastError();  // drop any stale error left by another HIP user in this thread
  hipLaunchKernelGGL(gn_bwd_reduce_kernel<SWISH>, dim3(B * C), dim3(T), 0, s, x0, x1, C0, C1, HW, G, dA, mean, rstd, gamma, beta,
                     p1, p2);
  const int n = (C > B * G ? C : B * G);
  hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, p1, p2, gamma, B, C, G, gs1, gs2, dgamma,
                     dbeta);
  const int bx = cdiv(HW, T) < 32 ? cdiv(HW, T) : 32;
  hipLaunchKernelGGL(gn_bwd_apply_kernel<SWISH>, dim3(B * C, bx), dim3(T), 0, s, x0, x1, C0, C1, HW, G, dA, mean, rstd, gamma,
                     beta, gs1, gs2, dx0, dx1);
  HDIFF_CHECK_LAUNCH("gn backward kernels");
  return HDIFF_OK;
}
}  // namespace

extern "C" int hdiff_gn_swish_bwd(const float* x0, const float* x1, int C0, int C1, int B, int HW, int G, const float* dA,
                                  const float* mean, const float* rstd, const float* gamma, const float* beta, float* ws,
                                  float* dx0, float* dx1, float* dgamma, float* dbeta, hdiff_stream_t stream) {
  HDIFF_CHECK_ARG(x0 && dA && mean && rstd && gamma && beta && ws && dx0 && dgamma && dbeta, "gn_swish_bwd: null pointer");
  HDIFF_CHECK_ARG(C1 == 0 || (x1 && dx1), "gn_swish_bwd: C1 > 0 without x1/dx1");
  HDIFF_CHECK_ARG(G > 0 && (C0 + C1) % G == 0 && B > 0 && HW > 0, "gn_swish_bwd: bad sizes");
  return gn_bwd_launch<true>(x0, x1, C0, C1, B, HW, G, dA, mean, rstd, gamma, beta, ws, dx0, dx1, dgamma, dbeta,
                             (hipStream_t)stream);
}

extern "C" int hdiff_gn_affine_bwd(const float* x, int C, int B, int HW, int G, const float* dY, const float* mean,
                                   const float* rstd, const float* gamma, const float* beta, float* ws, float* dx, float* dgamma,
                                   float* dbeta, hdiff_stream_t stream) {
  HDIFF_CHECK_ARG(x && dY && mean && rstd && gamma && beta && ws && dx && dgamma && dbeta, "gn_affine_bwd: null pointer");
  HDIFF_CHECK_ARG(G > 0 && C > 0 && C % G == 0 && B > 0 && HW > 0, "gn_affine_bwd: bad sizes");
  return gn_bwd_launch<false>(x, nullptr, C, 0, B, HW, G, dY, mean, rstd, gamma, beta, ws, dx, nullptr, dgamma, dbeta,
                              (hipStream_t)stream);
}

extern "C" int hdiff_bias_addvec_grad(const float* dy, int B, int C, int HW, float* dvec, float* dbias,
                                      hdiff_stream_t stream) {
  HDIFF_CHECK_ARG(dy && (dvec || dbias), "bias_addvec_grad: null pointer");
  (void)hipGetLastError();  // drop any stale error left by another HIP user in this thread
  if (dvec != nullptr) {
    hipLaunchKernelGGL(plane_sum_kernel, dim3(C, B), dim3(T), 0, (hipStream_t)stream, dy, C, HW, dvec);
    if (dbias != nullptr)
      hipLaunchKernelGGL(batch_sum_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, dvec, B, C, dbias);
  } else {
    hipLaunchKernelGGL(bias_addvec_grad_kernel, dim3(C), dim3(T), 0, (hipStream_t)stream, dy, B, C, HW, dvec, dbias);
  }
  HDIFF_CHECK_LAUNCH("bias_addvec_grad_kernel");
  return HDIFF_OK;
}
