// Flash-style multi-head self-attention core on the fp32-input MFMA (v_mfma_f32_16x16x4_f32), gfx950.
//
// Replaces softmax(Q K^T / sqrt(d)) V inside nn.MultiheadAttention(C, 8) called as attn(h, h, h)
// (reference: ModelCondition.py:189, 204-208).  The reference materialises the L x L score matrix per head (137 GB per
// sample at 256x256); here it never exists: per 64-key tile the scores live in MFMA accumulators only.
//
// Layout: qkv is [B][3C][L] (the packed in-projection run as a 1x1 conv over the NCHW activation), so every head's
// Q, K, V are "transposed" [d][L] slabs with L contiguous -- coalesced 256-byte rows for the K/V tiles.
//
// MFMA orientation ("keys on rows, queries on lanes"):
//   S^T[key][q]  = sum_d  K[key][d] * Q[q][d]      A = K tile (from LDS), B = Q (registers, pre-scaled by log2e/sqrt(d))
//   O^T[d][q]   += sum_key V[key][d] * P^T[key][q]  A = V tile (from LDS), B = P^T
// With the 16x16x4 shape the S^T accumulator (lane = query, 4 registers = keys 4g..4g+3 of lane group g) is already the
// B operand of the P.V product: MFMA number r of a 16-key subtile consumes register r of every lane, i.e. keys
// {r, 4+r, 8+r, 12+r}.  No cross-lane movement between the two products; softmax needs two lane exchanges per 64 keys.
// Online softmax: running max m and partial row sum l per query (lane-local, the four lane groups hold partial sums).
//
// Schedule: a wave owns NQ query tiles of 16.  Within a key tile the three stages of consecutive query tiles are
// software-pipelined in ONE instruction stream,
//        QK^T(q+1)   ||   softmax(q)   ||   P.V(q-1)
// so the matrix pipe (two MFMA streams, alternated so dependent accumulations are never back to back) always has the
// exp/max/sum VALU work of the middle stage issuing in its shadow (sched_group_barrier pins "1 MFMA : few VALU").
#include <stdlib.h>

#include <type_traits>

#include "common.h"

using namespace hdiff;

namespace {

constexpr int KT = 64;          // keys per tile
constexpr int KROW = KT + 4;    // LDS row stride (floats), 16-byte aligned rows
constexpr int ATT_THREADS = 256;

#define SGB_MFMA 0x008
#define SGB_VALU_TRANS 0x402   // VALU | TRANS (v_exp_f32 is a TRANS instruction)

template <int D, int NQ, int VARIANT>
__global__ __launch_bounds__(ATT_THREADS) void mha_flash_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                                    int C, int L, float qscale) {
  constexpr int KS = D / 4;                 // k-steps of the QK^T product
  constexpr int MT = (D + 15) / 16;         // 16-row M tiles of the PV product
  constexpr int DP = MT * 16;               // padded V rows
  constexpr int NV4 = 2 * D * (KT / 4);     // float4 per K+V tile
  constexpr int NLD = (NV4 + ATT_THREADS - 1) / ATT_THREADS;
  constexpr int QB = 4 * 16 * NQ;           // queries per workgroup
  constexpr int N_QK = 4 * KS;              // MFMAs of one QK^T(q)
  constexpr int N_PV = 16 * MT;             // MFMAs of one P.V(q)

  __shared__ __attribute__((aligned(16))) float sK[2][D * KROW];
  __shared__ __attribute__((aligned(16))) float sV[2][DP * KROW];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  const int head = blockIdx.y, b = blockIdx.z;
  const int qblk0 = blockIdx.x * QB + wave * (16 * NQ);
  const float* qbase = qkv + ((size_t)b * 3 * C + (size_t)head * D) * L;
  const float* kbase = qbase + (size_t)C * L;
  const float* vbase = kbase + (size_t)C * L;
  const bool vec_ok = (L & 3) == 0;
  const int ntiles = (L + KT - 1) / KT;

  // zero the padded V rows once (D < 16)
  if (DP > D) {
    for (int idx = tid; idx < 2 * (DP - D) * KROW; idx += ATT_THREADS) {
      const int bufi = idx / ((DP - D) * KROW), rem = idx - bufi * (DP - D) * KROW;
      sV[bufi][D * KROW + rem] = 0.f;
    }
  }

  // Q fragments: lane (q = i16, g) holds Q[q][d = 4s + g], pre-scaled so that exp2 can be used directly
  float qf[NQ][KS];
#pragma unroll
  for (int qt = 0; qt < NQ; ++qt) {
    const int q = qblk0 + qt * 16 + i16;
#pragma unroll
    for (int s = 0; s < KS; ++s) qf[qt][s] = (q < L) ? qbase[(size_t)(4 * s + g) * L + q] * qscale : 0.f;
  }

  float4 stage[NLD];
  auto stage_load = [&](int t) {
    const int kt0 = t * KT;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = tid + i * ATT_THREADS;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < NV4) {
        const int row = idx >> 4, seg = idx & 15;
        const float* src = (row < D ? kbase + (size_t)row * L : vbase + (size_t)(row - D) * L) + kt0 + seg * 4;
        const int key = kt0 + seg * 4;
        if (vec_ok) {
          if (key < L) v = *reinterpret_cast<const float4*>(src);
        } else {
          if (key + 0 < L) v.x = src[0];
          if (key + 1 < L) v.y = src[1];
          if (key + 2 < L) v.z = src[2];
          if (key + 3 < L) v.w = src[3];
        }
      }
      stage[i] = v;
    }
  };
  auto stage_store = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = tid + i * ATT_THREADS;
      if (idx < NV4) {
        const int row = idx >> 4, seg = idx & 15;
        float* dst = (row < D) ? &sK[buf][row * KROW + seg * 4] : &sV[buf][(row - D) * KROW + seg * 4];
        *reinterpret_cast<float4*>(dst) = stage[i];
      }
    }
  };

  f32x4 O[MT][NQ];
  float m_run[NQ], l_run[NQ];
#pragma unroll
  for (int qt = 0; qt < NQ; ++qt) {
    m_run[qt] = -1e30f;
    l_run[qt] = 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) O[mt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // One key tile.  MASK: the tile holds keys >= L (last tile of a ragged sequence).
  auto do_tile = [&](auto mask_tag, int t, int buf) {
    constexpr bool MASK = decltype(mask_tag)::value;
    // K and V fragments of the whole tile, shared by all NQ query tiles of this wave
    float kf[4][KS];
    float vf[4][MT][4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
      for (int s = 0; s < KS; ++s) kf[ks][s] = sK[buf][(4 * s + g) * KROW + ks * 16 + i16];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const float4 v4 = *reinterpret_cast<const float4*>(&sV[buf][(mt * 16 + i16) * KROW + ks * 16 + 4 * g]);
        vf[ks][mt][0] = v4.x; vf[ks][mt][1] = v4.y; vf[ks][mt][2] = v4.z; vf[ks][mt][3] = v4.w;
      }
    }
    f32x4 S[NQ][4];

    auto qk_mfma = [&](int qt, int i) {       // i-th MFMA of QK^T(qt): k-step s = i / 4, key subtile ks = i % 4
      const int s = i >> 2, ks = i & 3;
      if (s == 0) S[qt][ks] = f32x4{0.f, 0.f, 0.f, 0.f};
      S[qt][ks] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[ks][s], qf[qt][s], S[qt][ks], 0, 0, 0);
    };
    auto pv_mfma = [&](int qt, int i) {       // i-th MFMA of P.V(qt): mt = i % MT, then key subtile / register
      const int mt = i % MT, j = i / MT, ks = j >> 2, r = j & 3;
      O[mt][qt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[ks][mt][r], S[qt][ks][r], O[mt][qt], 0, 0, 0);
    };
    auto softmax = [&](int qt) {
      if (MASK) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (t * KT + ks * 16 + 4 * g + r >= L) S[qt][ks][r] = -1e30f;
      }
      float tm = fmaxf(fmaxf(S[qt][0][0], S[qt][0][1]), S[qt][0][2]);
      tm = fmaxf(fmaxf(tm, S[qt][0][3]), S[qt][1][0]);
      tm = fmaxf(fmaxf(tm, S[qt][1][1]), S[qt][1][2]);
      tm = fmaxf(fmaxf(tm, S[qt][1][3]), S[qt][2][0]);
      tm = fmaxf(fmaxf(tm, S[qt][2][1]), S[qt][2][2]);
      tm = fmaxf(fmaxf(tm, S[qt][2][3]), S[qt][3][0]);
      tm = fmaxf(fmaxf(tm, S[qt][3][1]), S[qt][3][2]);
      tm = fmaxf(tm, S[qt][3][3]);
      tm = fmaxf(tm, __shfl_xor(tm, 16, 64));
      tm = fmaxf(tm, __shfl_xor(tm, 32, 64));
      const float m_new = fmaxf(m_run[qt], tm);
      const float alpha = __builtin_amdgcn_exp2f(m_run[qt] - m_new);
      m_run[qt] = m_new;
      float ls0 = 0.f, ls1 = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pv = __builtin_amdgcn_exp2f(S[qt][ks][r] - m_new);
          S[qt][ks][r] = pv;
          if (r & 1) ls1 += pv; else ls0 += pv;
        }
      l_run[qt] = l_run[qt] * alpha + (ls0 + ls1);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) O[mt][qt] *= alpha;
    };

    if (VARIANT == 0 || VARIANT >= 3) {
      // plain order: all QK^T, all softmax, all P.V (hardware overlap between co-resident waves only)
#pragma unroll
      for (int qt = 0; qt < NQ; ++qt)
#pragma unroll
        for (int i = 0; i < N_QK; ++i) qk_mfma(qt, i);
#pragma unroll
      for (int qt = 0; qt < NQ; ++qt)
        if (VARIANT != 3) softmax(qt);      // 3 = timing-only ablation (wrong results): no softmax
#pragma unroll
      for (int i = 0; i < N_PV; ++i)
#pragma unroll
        for (int qt = 0; qt < NQ; ++qt) pv_mfma(qt, i);
      return;
    }
    // ---- software pipeline over the wave's query tiles
#pragma unroll
    for (int i = 0; i < N_QK; ++i) qk_mfma(0, i);
#pragma unroll
    for (int qt = 0; qt < NQ; ++qt) {
      // MFMA streams of this step, alternated: P.V(qt-1) and QK^T(qt+1)
      constexpr int NMAX = (N_QK > N_PV) ? N_QK : N_PV;
#pragma unroll
      for (int i = 0; i < NMAX; ++i) {
        if (qt > 0 && i < N_PV) pv_mfma(qt - 1, i);
        if (qt + 1 < NQ && i < N_QK) qk_mfma(qt + 1, i);
      }
      softmax(qt);
      // pin the interleave: one MFMA, then a few VALU/TRANS of softmax(qt)
      const int n_mfma = ((qt > 0) ? N_PV : 0) + ((qt + 1 < NQ) ? N_QK : 0);
      if (VARIANT == 2 && n_mfma > 0) {
        const int per = (80 + n_mfma - 1) / n_mfma;
#pragma unroll
        for (int i = 0; i < n_mfma; ++i) {
          __builtin_amdgcn_sched_group_barrier(SGB_MFMA, 1, 0);
          if (per <= 2) __builtin_amdgcn_sched_group_barrier(SGB_VALU_TRANS, 2, 0);
          else if (per == 3) __builtin_amdgcn_sched_group_barrier(SGB_VALU_TRANS, 3, 0);
          else if (per == 4) __builtin_amdgcn_sched_group_barrier(SGB_VALU_TRANS, 4, 0);
          else __builtin_amdgcn_sched_group_barrier(SGB_VALU_TRANS, 5, 0);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < N_PV; ++i) pv_mfma(NQ - 1, i);
  };

  stage_load(0);
  stage_store(0);
  __syncthreads();

  const int nfull = vec_ok ? (L / KT) : 0;   // ragged or unaligned sequences take the masked path for every tile
  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1;
    const int tn = (t + 1 < ntiles) ? t + 1 : t;   // the last iteration re-stages its own tile (branch-free; unused)
    if (VARIANT != 4) stage_load(tn);               // 4 = timing-only ablation (wrong results): no staging, no barrier
    if (t < nfull) do_tile(std::false_type{}, t, VARIANT == 4 ? 0 : buf);
    else do_tile(std::true_type{}, t, buf);
    if (VARIANT != 4) {
      stage_store(buf ^ 1);
      __syncthreads();
    }
  }

  // ---- normalise and store: out[b][head*D + d][q]
  float* obase = out + ((size_t)b * C + (size_t)head * D) * L;
#pragma unroll
  for (int qt = 0; qt < NQ; ++qt) {
    float lt = l_run[qt];
    lt += __shfl_xor(lt, 16, 64);
    lt += __shfl_xor(lt, 32, 64);
    const float inv = 1.0f / lt;
    const int q = qblk0 + qt * 16 + i16;
    if (q < L) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int d = mt * 16 + 4 * g + r;
          if (d < D) obase[(size_t)d * L + q] = O[mt][qt][r] * inv;
        }
    }
  }
}

int att_variant() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("HDIFF_ATT_VARIANT");
    v = e ? atoi(e) : 0;
  }
  return v;
}

template <int D, int NQ, int VARIANT>
void launch_v(const float* qkv, float* o, int B, int C, int heads, int L, float qscale, hipStream_t stream) {
  dim3 grid(cdiv(L, 64 * NQ), heads, B);
  hipLaunchKernelGGL((mha_flash_fwd_kernel<D, NQ, VARIANT>), grid, dim3(ATT_THREADS), 0, stream, qkv, o, C, L, qscale);
}

template <int D>
int launch_d(const float* qkv, float* o, int B, int C, int heads, int L, hipStream_t stream) {
  const float qscale = 1.4426950408889634f / sqrtf((float)D);
  (void)hipGetLastError();  // drop any stale error left by another HIP user in this thread
  if (L >= 512) {
    switch (att_variant()) {
      case 1: launch_v<D, 4, 1>(qkv, o, B, C, heads, L, qscale, stream); break;
      case 2: launch_v<D, 4, 2>(qkv, o, B, C, heads, L, qscale, stream); break;
      case 3: launch_v<D, 4, 3>(qkv, o, B, C, heads, L, qscale, stream); break;
      case 4: launch_v<D, 4, 4>(qkv, o, B, C, heads, L, qscale, stream); break;
      default: launch_v<D, 4, 0>(qkv, o, B, C, heads, L, qscale, stream); break;
    }
  } else {
    launch_v<D, 1, 0>(qkv, o, B, C, heads, L, qscale, stream);
  }
  HDIFF_CHECK_LAUNCH("mha_flash_fwd_kernel");
  return HDIFF_OK;
}

}  // namespace

extern "C" int hdiff_mha_flash_fwd(const float* qkv, float* o, int B, int C, int heads, int L, hdiff_stream_t stream) {
  HDIFF_CHECK_ARG(qkv && o, "mha_flash_fwd: null pointer");
  HDIFF_CHECK_ARG(B > 0 && L > 0 && heads > 0 && C % heads == 0, "mha_flash_fwd: bad sizes B=%d C=%d heads=%d L=%d", B, C,
                  heads, L);
  const int D = C / heads;
  hipStream_t s = (hipStream_t)stream;
  switch (D) {
    case 4: return launch_d<4>(qkv, o, B, C, heads, L, s);
    case 8: return launch_d<8>(qkv, o, B, C, heads, L, s);
    case 16: return launch_d<16>(qkv, o, B, C, heads, L, s);
    case 32: return launch_d<32>(qkv, o, B, C, heads, L, s);
    default: break;
  }
  hdiff::set_error("mha_flash_fwd: head dim %d not in {4, 8, 16, 32}", D);
  return HDIFF_ERR_INVALID;
}
