// Backward of the flash self-attention core (attention.hip), fp32 MFMA 16x16x4, gfx950.
//
// Reference: autograd through nn.MultiheadAttention(C, 8)(h, h, h) (ModelCondition.py:189, 204-208) in
// TrainCondition.py:60 (loss.backward()).  P is recomputed from Q, K and the forward's log2-domain log-sum-exp
// (lse2 = m + log2 l), never stored:  p = exp2(s2 - lse2),  dP = dO V^T,  dS = P o (dP - delta),  delta = rowsum(dO o O).
//
// Two kernels, no atomics, bitwise reproducible:
//   dQ kernel   (same geometry as the forward: a wave owns query tiles, streams 64-key K/V tiles through LDS)
//               S^T = K Q^T ; dP^T = V dO^T ; dS^T = P^T o (dP^T - delta) ; dQ^T += K^T dS^T      (3 MFMA products)
//   dKdV kernel (roles swapped: a wave owns key tiles, streams 64-query Q/dO tiles + lse/delta through LDS)
//               S = Q K^T ; dP = dO V^T ; dV^T += dO^T P ; dK^T += Q^T dS                            (4 MFMA products)
// In both, the 16x16 accumulator of the first product is directly the B operand of the accumulating product (MFMA
// number r consumes register r), exactly as in the forward.  All tensors keep the [B][3C][L] / [B][C][L] layouts.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

using namespace hdiff;

namespace {

constexpr int KT = 64;
constexpr int KROW = KT + 4;
constexpr int ATT_THREADS = 256;

// delta[b][h][q] = sum_d dO[b][h*D+d][q] * O[b][h*D+d][q]
__global__ void mha_delta_kernel(const float* __restrict__ o, const float* __restrict__ d_o, float* __restrict__ delta,
                                 int C, int D, int L, int total) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int q = idx % L;
  const int bh = idx / L;
  const int heads = C / D;
  const int b = bh / heads, h = bh - b * heads;
  const size_t base = ((size_t)b * C + (size_t)h * D) * L + q;
  float s = 0.f;
  for (int d = 0; d < D; ++d) s = fmaf(o[base + (size_t)d * L], d_o[base + (size_t)d * L], s);
  delta[idx] = s;
}

// Stage rows [row0, row0+nrows) x 64 columns starting at col0 of a [rows][L] slab into LDS rows of stride KROW
// (zero beyond L).  All 256 threads take part; 16-byte accesses when L % 4 == 0.
__device__ __forceinline__ void stage_rows(const float* __restrict__ src, int L, int col0, int nrows, float* dst, int tid,
                                           bool vec_ok) {
  for (int idx = tid; idx < nrows * (KT / 4); idx += ATT_THREADS) {
    const int row = idx >> 4, seg = idx & 15;
    const int col = col0 + seg * 4;
    const float* s = src + (size_t)row * L + col;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (vec_ok) {
      if (col < L) v = *reinterpret_cast<const float4*>(s);
    } else {
      if (col + 0 < L) v.x = s[0];
      if (col + 1 < L) v.y = s[1];
      if (col + 2 < L) v.z = s[2];
      if (col + 3 < L) v.w = s[3];
    }
    *reinterpret_cast<float4*>(&dst[row * KROW + seg * 4]) = v;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// dQ
// ---------------------------------------------------------------------------------------------------------------------
template <int D, int NQ>
__global__ __launch_bounds__(ATT_THREADS) void mha_bwd_dq_kernel(const float* __restrict__ qkv, const float* __restrict__ d_o,
                                                                 const float* __restrict__ lse2,
                                                                 const float* __restrict__ delta, float* __restrict__ dqkv,
                                                                 int C, int L, float qscale, float inv_sqrt_d) {
  constexpr int KS = D / 4;
  constexpr int MT = (D + 15) / 16;
  constexpr int DP = MT * 16;
  __shared__ __attribute__((aligned(16))) float sK[DP * KROW];
  __shared__ __attribute__((aligned(16))) float sV[DP * KROW];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  const TileId tile = xcd_tile();
  const int head = tile.head, b = tile.b, heads = gridDim.y;
  const int qblk0 = tile.x * (64 * NQ) + wave * (16 * NQ);
  const float* qbase = qkv + ((size_t)b * 3 * C + (size_t)head * D) * L;
  const float* kbase = qbase + (size_t)C * L;
  const float* vbase = kbase + (size_t)C * L;
  const float* dobase = d_o + ((size_t)b * C + (size_t)head * D) * L;
  const float* lbase = lse2 + ((size_t)b * heads + head) * L;
  const float* dbase = delta + ((size_t)b * heads + head) * L;
  const bool vec_ok = (L & 3) == 0;
  const int ntiles = (L + KT - 1) / KT;

  for (int idx = tid; idx < (DP - D) * KROW; idx += ATT_THREADS) {
    sK[D * KROW + idx] = 0.f;
    sV[D * KROW + idx] = 0.f;
  }

  float qf[NQ][KS], dof[NQ][KS], lse[NQ], dl[NQ];
  f32x4 dQ[MT][NQ];
#pragma unroll
  for (int qt = 0; qt < NQ; ++qt) {
    const int q = qblk0 + qt * 16 + i16;
    const bool ok = q < L;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      qf[qt][s] = ok ? qbase[(size_t)(4 * s + g) * L + q] * qscale : 0.f;
      dof[qt][s] = ok ? dobase[(size_t)(4 * s + g) * L + q] : 0.f;
    }
    lse[qt] = ok ? lbase[q] : 0.f;
    dl[qt] = ok ? dbase[q] : 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) dQ[mt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  for (int t = 0; t < ntiles; ++t) {
    __syncthreads();
    stage_rows(kbase, L, t * KT, D, sK, tid, vec_ok);
    stage_rows(vbase, L, t * KT, D, sV, tid, vec_ok);
    __syncthreads();
    const bool ragged = (t * KT + KT > L);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      float kf[KS], vkf[KS], kvf[MT][4];
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        kf[s] = sK[(4 * s + g) * KROW + ks * 16 + i16];
        vkf[s] = sV[(4 * s + g) * KROW + ks * 16 + i16];
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const float4 v4 = *reinterpret_cast<const float4*>(&sK[(mt * 16 + i16) * KROW + ks * 16 + 4 * g]);
        kvf[mt][0] = v4.x; kvf[mt][1] = v4.y; kvf[mt][2] = v4.z; kvf[mt][3] = v4.w;
      }
#pragma unroll
      for (int qt = 0; qt < NQ; ++qt) {
        f32x4 S = {0.f, 0.f, 0.f, 0.f}, dP = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          S = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[s], qf[qt][s], S, 0, 0, 0);
          dP = __builtin_amdgcn_mfma_f32_16x16x4f32(vkf[s], dof[qt][s], dP, 0, 0, 0);
        }
        f32x4 dS;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float p = __builtin_amdgcn_exp2f(S[r] - lse[qt]);
          if (ragged && (t * KT + ks * 16 + 4 * g + r >= L)) p = 0.f;
          dS[r] = p * (dP[r] - dl[qt]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            dQ[mt][qt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kvf[mt][r], dS[r], dQ[mt][qt], 0, 0, 0);
      }
    }
  }

  float* obase = dqkv + ((size_t)b * 3 * C + (size_t)head * D) * L;     // Q third
#pragma unroll
  for (int qt = 0; qt < NQ; ++qt) {
    const int q = qblk0 + qt * 16 + i16;
    if (q < L) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int d = mt * 16 + 4 * g + r;
          if (d < D) obase[(size_t)d * L + q] = dQ[mt][qt][r] * inv_sqrt_d;
        }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// dK, dV
// ---------------------------------------------------------------------------------------------------------------------
template <int D, int NK>
__global__ __launch_bounds__(ATT_THREADS) void mha_bwd_dkv_kernel(const float* __restrict__ qkv,
                                                                  const float* __restrict__ d_o,
                                                                  const float* __restrict__ lse2,
                                                                  const float* __restrict__ delta,
                                                                  float* __restrict__ dqkv, int C, int L, float qscale,
                                                                  float inv_sqrt_d) {
  constexpr int KS = D / 4;
  constexpr int MT = (D + 15) / 16;
  constexpr int DP = MT * 16;
  __shared__ __attribute__((aligned(16))) float sQ[DP * KROW];
  __shared__ __attribute__((aligned(16))) float sO[DP * KROW];
  __shared__ __attribute__((aligned(16))) float sL[KT];
  __shared__ __attribute__((aligned(16))) float sD[KT];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  const TileId tile = xcd_tile();
  const int head = tile.head, b = tile.b, heads = gridDim.y;
  const int kblk0 = tile.x * (64 * NK) + wave * (16 * NK);
  const float* qbase = qkv + ((size_t)b * 3 * C + (size_t)head * D) * L;
  const float* kbase = qbase + (size_t)C * L;
  const float* vbase = kbase + (size_t)C * L;
  const float* dobase = d_o + ((size_t)b * C + (size_t)head * D) * L;
  const float* lbase = lse2 + ((size_t)b * heads + head) * L;
  const float* dbase = delta + ((size_t)b * heads + head) * L;
  const bool vec_ok = (L & 3) == 0;
  const int ntiles = (L + KT - 1) / KT;

  for (int idx = tid; idx < (DP - D) * KROW; idx += ATT_THREADS) {
    sQ[D * KROW + idx] = 0.f;
    sO[D * KROW + idx] = 0.f;
  }

  // this wave's keys live on the lanes: B operands of S = Q K^T (pre-scaled) and dP = dO V^T
  float kreg[NK][KS], vreg[NK][KS];
  f32x4 dK[MT][NK], dV[MT][NK];
#pragma unroll
  for (int kt = 0; kt < NK; ++kt) {
    const int key = kblk0 + kt * 16 + i16;
    const bool ok = key < L;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      kreg[kt][s] = ok ? kbase[(size_t)(4 * s + g) * L + key] * qscale : 0.f;
      vreg[kt][s] = ok ? vbase[(size_t)(4 * s + g) * L + key] : 0.f;
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      dK[mt][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
      dV[mt][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }

  for (int t = 0; t < ntiles; ++t) {
    __syncthreads();
    stage_rows(qbase, L, t * KT, D, sQ, tid, vec_ok);
    stage_rows(dobase, L, t * KT, D, sO, tid, vec_ok);
    if (tid < KT) {
      const int q = t * KT + tid;
      sL[tid] = (q < L) ? lbase[q] : __builtin_inff();   // +inf: p = exp2(s - inf) = 0 for queries beyond L
      sD[tid] = (q < L) ? dbase[q] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int qs = 0; qs < 4; ++qs) {
      float qa[KS], doa[KS], qvf[MT][4], dovf[MT][4];
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        qa[s] = sQ[(4 * s + g) * KROW + qs * 16 + i16];
        doa[s] = sO[(4 * s + g) * KROW + qs * 16 + i16];
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const float4 a4 = *reinterpret_cast<const float4*>(&sQ[(mt * 16 + i16) * KROW + qs * 16 + 4 * g]);
        const float4 b4 = *reinterpret_cast<const float4*>(&sO[(mt * 16 + i16) * KROW + qs * 16 + 4 * g]);
        qvf[mt][0] = a4.x; qvf[mt][1] = a4.y; qvf[mt][2] = a4.z; qvf[mt][3] = a4.w;
        dovf[mt][0] = b4.x; dovf[mt][1] = b4.y; dovf[mt][2] = b4.z; dovf[mt][3] = b4.w;
      }
      // rows of the accumulators are queries 4g + r of this subtile
      const float4 l4 = *reinterpret_cast<const float4*>(&sL[qs * 16 + 4 * g]);
      const float4 d4 = *reinterpret_cast<const float4*>(&sD[qs * 16 + 4 * g]);
      const float lr[4] = {l4.x, l4.y, l4.z, l4.w};
      const float dr[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
      for (int kt = 0; kt < NK; ++kt) {
        f32x4 S = {0.f, 0.f, 0.f, 0.f}, dP = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          S = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[s], kreg[kt][s], S, 0, 0, 0);
          dP = __builtin_amdgcn_mfma_f32_16x16x4f32(doa[s], vreg[kt][s], dP, 0, 0, 0);
        }
        f32x4 P, dS;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          P[r] = __builtin_amdgcn_exp2f(S[r] - lr[r]);
          dS[r] = P[r] * (dP[r] - dr[r]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            dV[mt][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(dovf[mt][r], P[r], dV[mt][kt], 0, 0, 0);
            dK[mt][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(qvf[mt][r], dS[r], dK[mt][kt], 0, 0, 0);
          }
      }
    }
  }

  float* kout = dqkv + ((size_t)b * 3 * C + (size_t)C + (size_t)head * D) * L;
  float* vout = kout + (size_t)C * L;
#pragma unroll
  for (int kt = 0; kt < NK; ++kt) {
    const int key = kblk0 + kt * 16 + i16;
    if (key < L) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int d = mt * 16 + 4 * g + r;
          if (d < D) {
            kout[(size_t)d * L + key] = dK[mt][kt][r] * inv_sqrt_d;
            vout[(size_t)d * L + key] = dV[mt][kt][r];
          }
        }
    }
  }
}

template <int D>
int launch_bwd(const float* qkv, const float* o, const float* d_o, const float* lse2, float* delta, float* dqkv, int B,
               int C, int heads, int L, hipStream_t stream) {
  const float inv_sqrt_d = 1.0f / sqrtf((float)D);
  const float qscale = 1.4426950408889634f * inv_sqrt_d;
  const int total = B * heads * L;
  (void)hipGetLastError();  // drop any stale error left by another HIP user in this thread
  hipLaunchKernelGGL(mha_delta_kernel, dim3(cdiv(total, 256)), dim3(256), 0, stream, o, d_o, delta, C, D, L, total);
  if (L >= 512) {
    constexpr int N = (D >= 32) ? 2 : 4;
    dim3 grid(cdiv(L, 64 * N), heads, B);
    hipLaunchKernelGGL((mha_bwd_dq_kernel<D, N>), grid, dim3(ATT_THREADS), 0, stream, qkv, d_o, lse2, delta, dqkv, C, L,
                       qscale, inv_sqrt_d);
    hipLaunchKernelGGL((mha_bwd_dkv_kernel<D, N>), grid, dim3(ATT_THREADS), 0, stream, qkv, d_o, lse2, delta, dqkv, C, L,
                       qscale, inv_sqrt_d);
  } else {
    dim3 grid(cdiv(L, 64), heads, B);
    hipLaunchKernelGGL((mha_bwd_dq_kernel<D, 1>), grid, dim3(ATT_THREADS), 0, stream, qkv, d_o, lse2, delta, dqkv, C, L,
                       qscale, inv_sqrt_d);
    hipLaunchKernelGGL((mha_bwd_dkv_kernel<D, 1>), grid, dim3(ATT_THREADS), 0, stream, qkv, d_o, lse2, delta, dqkv, C, L,
                       qscale, inv_sqrt_d);
  }
  HDIFF_CHECK_LAUNCH("mha_bwd kernels");
  return HDIFF_OK;
}

}  // namespace

extern "C" int hdiff_mha_flash_bwd(const float* qkv, const float* o, const float* d_o, const float* lse2, float* delta,
                                   float* dqkv, int B, int C, int heads, int L, hdiff_stream_t stream) {
  HDIFF_CHECK_ARG(qkv && o && d_o && lse2 && delta && dqkv, "mha_flash_bwd: null pointer");
  HDIFF_CHECK_ARG(B > 0 && L > 0 && heads > 0 && C % heads == 0, "mha_flash_bwd: bad sizes B=%d C=%d heads=%d L=%d", B, C,
                  heads, L);
  const int D = C / heads;
  hipStream_t s = (hipStream_t)stream;
  switch (D) {
    case 4: return launch_bwd<4>(qkv, o, d_o, lse2, delta, dqkv, B, C, heads, L, s);
    case 8: return launch_bwd<8>(qkv, o, d_o, lse2, delta, dqkv, B, C, heads, L, s);
    case 16: return launch_bwd<16>(qkv, o, d_o, lse2, delta, dqkv, B, C, heads, L, s);
    case 32: return launch_bwd<32>(qkv, o, d_o, lse2, delta, dqkv, B, C, heads, L, s);
    default: break;
  }
  hdiff::set_error("mha_flash_bwd: head dim %d not in {4, 8, 16, 32}", D);
  return HDIFF_ERR_INVALID;
}
