// Single-head self-attention with a WIDE head (d_head = C > 64), gfx950: the core of the reference's AttnBlock
// (ModelCondition.py:92-120: softmax(q k^T * C^-1/2) v over all H*W positions with ONE head of width in_ch).
//
// AttnBlock is dead code in the reference (UNet never instantiates it), so this kernel is
// written for clarity, not for the roofline: one workgroup per query row, the row of scores lives in LDS.
//   pass 1  s[k] = scale * sum_c q[c] K[c][k]          threads stride the keys (coalesced rows of K), running max
//   pass 2  p[k] = exp(s[k] - max), row sum
//   pass 3  out[c] = sum_k p[k] V[c][k] / rowsum        waves stride the channels, lanes stride the keys (wave reduction)
// Heads of width <= 64 go to the flash kernels (hdiff_mha_flash_fwd with heads = 1).  qkv is [B][3C][L], out [B][C][L].
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

}  // namespace

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
