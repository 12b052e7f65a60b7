// Single-head self-attention with a WIDE head (d_head = C > 64), gfx950: the core of the reference's AttnBlock
// (ModelCondition.py:92-120: softmax(q k^T * C^-1/2) v over all H*W positions with ONE head of width in_ch).
//
// AttnBlock is dead code in the reference (UNet never instantiates it), so this kernel is
// written for clarity, not for the roofline: one workgroup per query row, the row of scores lives in LDS.
//   pass 1  s[k] = scale * sum_c q[c] K[c][k]          threads stride the keys (coalesced rows of K), running max
//   pass 2  p[k] = exp(s[k] - max), row sum
//   pass 3  out[c] = sum_k p[k] V[c][k] / rowsum        waves stride the channels, lanes stride the keys (wave reduction)
// Heads of width <= 64 go to the flash kernels (hdiff_mha_flash_fwd with heads = 1).  qkv is [B][3C][L], out [B][C][L].
// hdiff_mha_wide_bwd (any width) is the backward of both: a row pass for dQ, a column pass for dK and dV.
#include "common.h"

using namespace hdiff;

namespace {

constexpr int WT = 256;

__global__ __launch_bounds__(WT) void mha_wide_rows_kernel(const float* __restrict__ qkv, float* __restrict__ out, int C, int L,
                                                           float scale) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sq = smem;            // [C]
  float* ss = smem + C;        // [L]
  __shared__ float red[WT / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = blockIdx.x, b = blockIdx.y;
  const float* qb = qkv + (size_t)b * 3 * C * L;
  const float* kb = qb + (size_t)C * L;
  const float* vb = kb + (size_t)C * L;
  for (int c = tid; c < C; c += WT) sq[c] = qb[(size_t)c * L + q] * scale;
  __syncthreads();
  float m = -__builtin_inff();
  for (int k = tid; k < L; k += WT) {
    float s = 0.f;
    for (int c = 0; c < C; ++c) s = fmaf(sq[c], kb[(size_t)c * L + k], s);
    ss[k] = s;
    m = fmaxf(m, s);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float l = 0.f;
  for (int k = tid; k < L; k += WT) {
    const float p = __expf(ss[k] - m);
    ss[k] = p;
    l += p;
  }
  l = wave_sum(l);
  if (lane == 0) red[wave] = l;
  __syncthreads();
  const float inv = 1.0f / ((red[0] + red[1]) + (red[2] + red[3]));
  for (int c = wave; c < C; c += WT / 64) {
    const float* vr = vb + (size_t)c * L;
    float a = 0.f;
    for (int k = lane; k < L; k += 64) a = fmaf(ss[k], vr[k], a);
    a = wave_sum(a);
    if (lane == 0) out[((size_t)b * C + c) * L + q] = a * inv;
  }
}

// ---- backward, in the same plain style (AttnBlock is dead code in the reference: this exists so that the class trains).
// Row pass, one workgroup per query q: recomputes p[k] = softmax_k(s), dP[k] = dO[q] . V[k], delta = sum_k p dP,
// dS[k] = p (dP - delta);  dQ[q] = scale * sum_k dS[k] K[k];  leaves lse[q] = max + log(sum) and delta[q] for the column pass.
__global__ __launch_bounds__(WT) void mha_wide_bwd_rows_kernel(const float* __restrict__ qkv, const float* __restrict__ d_o,
                                                               float* __restrict__ dqkv, float* __restrict__ lse,
                                                               float* __restrict__ delta, int C, int L, float scale) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sq = smem;                // [C] q * scale
  float* sdo = smem + C;           // [C] dO[q]
  float* sp = smem + 2 * C;        // [L] scores -> p
  float* sds = smem + 2 * C + L;   // [L] dP -> dS
  __shared__ float red[WT / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = blockIdx.x, b = blockIdx.y;
  const float* qb = qkv + (size_t)b * 3 * C * L;
  const float* kb = qb + (size_t)C * L;
  const float* vb = kb + (size_t)C * L;
  const float* dob = d_o + (size_t)b * C * L;
  for (int c = tid; c < C; c += WT) {
    sq[c] = qb[(size_t)c * L + q] * scale;
    sdo[c] = dob[(size_t)c * L + q];
  }
  __syncthreads();
  float m = -__builtin_inff();
  for (int k = tid; k < L; k += WT) {
    float s = 0.f, dp = 0.f;
    for (int c = 0; c < C; ++c) {
      s = fmaf(sq[c], kb[(size_t)c * L + k], s);
      dp = fmaf(sdo[c], vb[(size_t)c * L + k], dp);
    }
    sp[k] = s;
    sds[k] = dp;
    m = fmaxf(m, s);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float l = 0.f;
  for (int k = tid; k < L; k += WT) {
    const float p = __expf(sp[k] - m);
    sp[k] = p;
    l += p;
  }
  l = wave_sum(l);
  if (lane == 0) red[wave] = l;
  __syncthreads();
  const float lsum = (red[0] + red[1]) + (red[2] + red[3]);
  const float inv = 1.0f / lsum;
  __syncthreads();
  float dl = 0.f;
  for (int k = tid; k < L; k += WT) {
    const float p = sp[k] * inv;
    sp[k] = p;
    dl = fmaf(p, sds[k], dl);
  }
  dl = wave_sum(dl);
  if (lane == 0) red[wave] = dl;
  __syncthreads();
  const float del = (red[0] + red[1]) + (red[2] + red[3]);
  for (int k = tid; k < L; k += WT) sds[k] = sp[k] * (sds[k] - del);
  if (tid == 0) {
    lse[(size_t)b * L + q] = m + __logf(lsum);
    delta[(size_t)b * L + q] = del;
  }
  __syncthreads();
  float* dqb = dqkv + (size_t)b * 3 * C * L;
  for (int c = wave; c < C; c += WT / 64) {
    const float* kr = kb + (size_t)c * L;
    float a = 0.f;
    for (int k = lane; k < L; k += 64) a = fmaf(sds[k], kr[k], a);
    a = wave_sum(a);
    if (lane == 0) dqb[(size_t)c * L + q] = a * scale;
  }
}

// Column pass, one workgroup per key k: p[i] = exp(scale q_i . k - lse[i]), dS[i] = p (dO_i . v - delta[i]) over all queries i;
// dV[k] = sum_i p[i] dO[i];  dK[k] = scale * sum_i dS[i] Q[i].
__global__ __launch_bounds__(WT) void mha_wide_bwd_cols_kernel(const float* __restrict__ qkv, const float* __restrict__ d_o,
                                                               float* __restrict__ dqkv, const float* __restrict__ lse,
                                                               const float* __restrict__ delta, int C, int L, float scale) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sk = smem;                // [C] k * scale
  float* sv = smem + C;            // [C] v
  float* sp = smem + 2 * C;        // [L]
  float* sds = smem + 2 * C + L;   // [L]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int k = blockIdx.x, b = blockIdx.y;
  const float* qb = qkv + (size_t)b * 3 * C * L;
  const float* kb = qb + (size_t)C * L;
  const float* vb = kb + (size_t)C * L;
  const float* dob = d_o + (size_t)b * C * L;
  for (int c = tid; c < C; c += WT) {
    sk[c] = kb[(size_t)c * L + k] * scale;
    sv[c] = vb[(size_t)c * L + k];
  }
  __syncthreads();
  for (int i = tid; i < L; i += WT) {
    float s = 0.f, dp = 0.f;
    for (int c = 0; c < C; ++c) {
      s = fmaf(sk[c], qb[(size_t)c * L + i], s);
      dp = fmaf(sv[c], dob[(size_t)c * L + i], dp);
    }
    const float p = __expf(s - lse[(size_t)b * L + i]);
    sp[i] = p;
    sds[i] = p * (dp - delta[(size_t)b * L + i]);
  }
  __syncthreads();
  float* dkb = dqkv + ((size_t)b * 3 + 1) * C * L;
  float* dvb = dkb + (size_t)C * L;
  for (int c = wave; c < C; c += WT / 64) {
    const float* qr = qb + (size_t)c * L;
    const float* dor = dob + (size_t)c * L;
    float ak = 0.f, av = 0.f;
    for (int i = lane; i < L; i += 64) {
      ak = fmaf(sds[i], qr[i], ak);
      av = fmaf(sp[i], dor[i], av);
    }
    ak = wave_sum(ak);
    av = wave_sum(av);
    if (lane == 0) {
      dkb[(size_t)c * L + k] = ak * scale;
      dvb[(size_t)c * L + k] = av;
    }
  }
}

}  // namespace

extern "C" int hdiff_mha_wide_bwd(const float* qkv, const float* d_o, float* dqkv, float* ws, int B, int C, int L,
                                  hdiff_stream_t stream) {
  HDIFF_CHECK_ARG(qkv && d_o && dqkv && ws, "mha_wide_bwd: null pointer");
  HDIFF_CHECK_ARG(B > 0 && B <= 65535 && C > 0 && L > 0, "mha_wide_bwd: bad sizes B=%d C=%d L=%d", B, C, L);
  const size_t lds = (size_t)(2 * C + 2 * L) * sizeof(float);
  HDIFF_CHECK_ARG(lds <= 150 * 1024, "mha_wide_bwd: two rows of %d scores + 2 x %d channels do not fit in LDS", L, C);
  static uint64_t attr_mask = 0;
  if (first_use_on_device(attr_mask)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mha_wide_bwd_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mha_wide_bwd_cols_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
  }
  float* lse = ws;                       // [B][L]
  float* delta = ws + (size_t)B * L;     // [B][L]
  const float scale = 1.0f / sqrtf((float)C);
  (void)hipGetLastError();  // drop any stale error left by another HIP user in this thread
  hipLaunchKernelGGL(mha_wide_bwd_rows_kernel, dim3(L, B), dim3(WT), lds, (hipStream_t)stream, qkv, d_o, dqkv, lse, delta, C, L,
                     scale);
  hipLaunchKernelGGL(mha_wide_bwd_cols_kernel, dim3(L, B), dim3(WT), lds, (hipStream_t)stream, qkv, d_o, dqkv, lse, delta, C, L,
                     scale);
  HDIFF_CHECK_LAUNCH("mha_wide_bwd kernels");
  return HDIFF_OK;
}

extern "C" int hdiff_mha_wide_fwd(const float* qkv, float* o, int B, int C, int L, hdiff_stream_t stream) {
  HDIFF_CHECK_ARG(qkv && o, "mha_wide_fwd: null pointer");
  HDIFF_CHECK_ARG(B > 0 && B <= 65535 && C > 0 && L > 0, "mha_wide_fwd: bad sizes B=%d C=%d L=%d", B, C, L);
  const size_t lds = (size_t)(C + L) * sizeof(float);
  HDIFF_CHECK_ARG(lds <= 150 * 1024, "mha_wide_fwd: a row of %d scores + %d channels does not fit in LDS", L, C);
  static uint64_t attr_mask = 0;
  if (first_use_on_device(attr_mask))
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mha_wide_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
  (void)hipGetLastError();  // drop any stale error left by another HIP user in this thread
  hipLaunchKernelGGL(mha_wide_rows_kernel, dim3(L, B), dim3(WT), lds, (hipStream_t)stream, qkv, o, C, L,
                     1.0f / sqrtf((float)C));
  HDIFF_CHECK_LAUNCH("mha_wide_rows_kernel");
  return HDIFF_OK;
}
