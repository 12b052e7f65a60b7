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
// Cost model (measured on MI355X, tools/mfma_valu.hip): an fp32 MFMA and an ordinary VALU instruction do NOT overlap on
// a SIMD -- the fp32 MFMA runs at the vector FMA rate and every VALU instruction (about 5 cycles, v_exp_f32 about 11) is
// issue time taken away from the MFMA stream, even from a second wave.  The kernel therefore minimises VALU work per score:
//   * lazy running max: m only moves when some score exceeds it by more than 2^64 (detected from the row sum itself, so
//     the common path has no max at all): p = exp2(s - m) directly; fp32 keeps full relative precision at any scale,
//   * packed fp32 math (v_pk_add_f32) for the subtract and the row sums: two scores per instruction,
//   * MFMA accumulators kept in VGPRs (-amdgpu-mfma-vgpr-form): no v_accvgpr copies between the products and softmax.
// A wave owns NQ query tiles of 16 and walks them one at a time per key tile; P.V(q) and QK^T(q+1) MFMAs are alternated
// so that dependent accumulations are never issued back to back.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

using namespace hdiff;

namespace {

constexpr int KT = 64;          // keys per tile
constexpr int KROW = KT + 4;    // LDS row stride (floats), 16-byte aligned rows
constexpr int ATT_THREADS = 256;
constexpr float RESCALE_LIMIT = 1.8446744e19f;   // 2^64: a row sum at or above it (or inf) triggers the max update

typedef float f32x2 __attribute__((ext_vector_type(2)));

// one v_pk_add_f32 (pure: no memory, no side effects -- the compiler may schedule and CSE it like any VALU instruction)
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) {
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

template <int D, int NQ>
__global__ __launch_bounds__(ATT_THREADS) void mha_flash_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                                    float* __restrict__ lse2, int C, int L, float qscale, int check) {
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
  const TileId tile = xcd_tile();
  const int head = tile.head, b = tile.b;
  const int qblk0 = tile.x * QB + wave * (16 * NQ);
  const float* qbase = qkv + ((size_t)b * 3 * C + (size_t)head * D) * L;
  const float* kbase = qbase + (size_t)C * L;
  const float* vbase = kbase + (size_t)C * L;
  const bool vec_ok = (L & 3) == 0;
  const int ntiles = (L + KT - 1) / KT;

  if (check) {
    // second pass behind mha_flash_fwd_fast_kernel: only query blocks it flagged (NaN in their first output row) are
    // recomputed here with the overflow-proof running max; everything else exits at once
    const int q = tile.x * QB + tid;
    const bool flagged = (tid < QB) && (q < L) && isnan(out[((size_t)b * C + (size_t)head * D) * L + q]);
    if (!__syncthreads_or(flagged)) return;
  }

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

  f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  asm volatile("" : "+v"(zero4));     // opaque: keep ONE live zero tuple instead of re-materialising zeros per MFMA chain
  f32x4 O[MT][NQ];
  float m_run[NQ], l_run[NQ];
#pragma unroll
  for (int qt = 0; qt < NQ; ++qt) {
    m_run[qt] = -1e30f;       // finite: the first tile always takes the max-update path (s - m overflows exp2)
    l_run[qt] = 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) O[mt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // One key tile.  MASK: the tile may hold keys >= L (ragged / unaligned sequences).
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
    f32x4 S[2][4];     // scores of the current and the next query tile
    f32x4 P[4];
    const f32x4 ZERO4 = zero4;

    auto qk_mfma = [&](int qt, int i) {       // i-th MFMA of QK^T(qt): k-step s = i / 4, key subtile ks = i % 4
      const int s = i >> 2, ks = i & 3;
      // the first k-step reads the shared zero tuple as C: no per-chain zeroing of the accumulator registers
      S[qt & 1][ks] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[ks][s], qf[qt][s], s == 0 ? ZERO4 : S[qt & 1][ks], 0, 0, 0);
    };
    auto pv_mfma = [&](int qt, int i) {       // i-th MFMA of P.V(qt): mt = i % MT, then key subtile / register
      const int mt = i % MT, j = i / MT, ks = j >> 2, r = j & 3;
      O[mt][qt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[ks][mt][r], P[ks][r], O[mt][qt], 0, 0, 0);
    };
    auto softmax = [&](int qt) {
      f32x4(&Sq)[4] = S[qt & 1];
      if (MASK) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (t * KT + ks * 16 + 4 * g + r >= L) Sq[ks][r] = -__builtin_inff();
      }
      // common path: no max, two scores per packed instruction
      const f32x2 nm2 = {-m_run[qt], -m_run[qt]};     // added, not subtracted: a v2f32 fadd selects v_pk_add_f32
      f32x2 sum2 = {0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const f32x2 a = __builtin_shufflevector(Sq[ks], Sq[ks], 0, 1) + nm2;
        const f32x2 c = __builtin_shufflevector(Sq[ks], Sq[ks], 2, 3) + nm2;
        const f32x2 pa = {__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
        const f32x2 pc = {__builtin_amdgcn_exp2f(c.x), __builtin_amdgcn_exp2f(c.y)};
        sum2 += pa;
        sum2 += pc;
        P[ks] = f32x4{pa.x, pa.y, pc.x, pc.y};
      }
      float ls = sum2.x + sum2.y;
      if (__any(!(ls < RESCALE_LIMIT))) {
        // rare path (always the first tile): move the running max, rescale what was accumulated at the old one
        float tm = fmaxf(fmaxf(Sq[0][0], Sq[0][1]), fmaxf(Sq[0][2], Sq[0][3]));
#pragma unroll
        for (int ks = 1; ks < 4; ++ks) tm = fmaxf(tm, fmaxf(fmaxf(Sq[ks][0], Sq[ks][1]), fmaxf(Sq[ks][2], Sq[ks][3])));
        tm = fmaxf(tm, __shfl_xor(tm, 16, 64));
        tm = fmaxf(tm, __shfl_xor(tm, 32, 64));
        const float m_new = fmaxf(m_run[qt], tm);
        const float alpha = __builtin_amdgcn_exp2f(m_run[qt] - m_new);
        m_run[qt] = m_new;
        ls = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pv = __builtin_amdgcn_exp2f(Sq[ks][r] - m_new);
            P[ks][r] = pv;
            ls += pv;
          }
        l_run[qt] = l_run[qt] * alpha + ls;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) O[mt][qt] *= alpha;
      } else {
        l_run[qt] += ls;
      }
    };

#pragma unroll
    for (int i = 0; i < N_QK; ++i) qk_mfma(0, i);
#pragma unroll
    for (int qt = 0; qt < NQ; ++qt) {
      softmax(qt);
      // P.V(qt) alternated with QK^T(qt+1): consecutive MFMAs never accumulate into the same registers
      constexpr int NMAX = (N_QK > N_PV) ? N_QK : N_PV;
#pragma unroll
      for (int i = 0; i < NMAX; ++i) {
        if (i < N_PV) pv_mfma(qt, i);
        if (qt + 1 < NQ && i < N_QK) qk_mfma(qt + 1, i);
      }
    }
  };

  stage_load(0);
  stage_store(0);
  __syncthreads();

  const int nfull = vec_ok ? (L / KT) : 0;   // ragged or unaligned sequences take the masked path for every tile
  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1;
    const int tn = (t + 1 < ntiles) ? t + 1 : t;   // the last iteration re-stages its own tile (branch-free; unused)
    stage_load(tn);
    if (t < nfull) do_tile(std::false_type{}, t, buf);
    else do_tile(std::true_type{}, t, buf);
    stage_store(buf ^ 1);
    __syncthreads();
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
    // log2-domain log-sum-exp of the scaled scores (what the backward kernels recompute P from)
    if (lse2 != nullptr && q < L && g == 0) lse2[((size_t)b * gridDim.y + head) * L + q] = m_run[qt] + __builtin_amdgcn_logf(lt);
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


// ---------------------------------------------------------------------------------------------------------------------
// Fast path for full-length tiles (L % 64 == 0): FIXED softmax reference point.
// The running max is taken from the first key tile only and never moved: p = exp2(s - m1).  fp32 keeps full relative
// precision at any scale, so this is exact as long as no later score exceeds m1 by ~2^7 (88 nats) -- and then the
// product overflows to inf, which is detected from the row sum at the end: the wave poisons its outputs with NaN and the
// overflow-proof kernel above, launched right behind with check = 1, recomputes just those query blocks.
// What it buys: the hot loop has no max, no compare, no branch and no subtraction -- -m1 is the initial accumulator of the
// QK^T MFMA chain -- so per 16x16 score tile the VALU work is 4 v_exp_f32 + 2 v_pk_add_f32 per lane and nothing else
// (on gfx950 every VALU instruction is fp32-MFMA issue time).
// ---------------------------------------------------------------------------------------------------------------------
constexpr float FAST_OVERFLOW_LIMIT = 1.2379400e27f;   // 2^90: a row sum at or above it (or NaN) -> recompute safely

template <int D, int NQ>
__global__ __launch_bounds__(ATT_THREADS) void mha_flash_fwd_fast_kernel(const float* __restrict__ qkv,
                                                                         float* __restrict__ out,
                                                                         float* __restrict__ lse2, int C, int L,
                                                                         float qscale) {
  constexpr int KS = D / 4;
  constexpr int MT = (D + 15) / 16;
  constexpr int DP = MT * 16;
  constexpr int NV4 = 2 * D * (KT / 4);
  constexpr int NLD = (NV4 + ATT_THREADS - 1) / ATT_THREADS;
  constexpr int QB = 4 * 16 * NQ;
  constexpr int N_QK = 4 * KS;
  constexpr int N_PV = 16 * MT;

  __shared__ __attribute__((aligned(16))) float sK[2][D * KROW];
  __shared__ __attribute__((aligned(16))) float sV[2][DP * KROW];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  const TileId tile = xcd_tile();
  // the tile coordinates come out of integer divisions the compiler evaluates on the vector ALU: moved to scalar registers, so
  // that everything derived from them (the K / V base pointers of the staging loads) is scalar too
  const int head = __builtin_amdgcn_readfirstlane(tile.head), b = __builtin_amdgcn_readfirstlane(tile.b);
  const int qblk0 = __builtin_amdgcn_readfirstlane(tile.x) * QB + wave * (16 * NQ);
  const float* qbase = qkv + ((size_t)b * 3 * C + (size_t)head * D) * L;
  const float* kbase = qbase + (size_t)C * L;
  const float* vbase = kbase + (size_t)C * L;
  const int ntiles = L / KT;

  if (DP > D) {
    for (int idx = tid; idx < 2 * (DP - D) * KROW; idx += ATT_THREADS) {
      const int bufi = idx / ((DP - D) * KROW), rem = idx - bufi * (DP - D) * KROW;
      sV[bufi][D * KROW + rem] = 0.f;
    }
  }

  float qf[NQ][KS];
#pragma unroll
  for (int qt = 0; qt < NQ; ++qt) {
    const int q = qblk0 + qt * 16 + i16;
#pragma unroll
    for (int s = 0; s < KS; ++s) qf[qt][s] = (q < L) ? qbase[(size_t)(4 * s + g) * L + q] * qscale : 0.f;
  }

  f32x4 stage[NLD];
  // d_head >= 16: a staging slot (256 consecutive 16-byte pieces = 16 rows) lies entirely in K or entirely in V, so it is
  // read with buffer loads: resource = the head's K (or V) rows, vector offset = the lane's fixed byte offset, SCALAR offset =
  // the tile -- no vector address arithmetic in the loop (a 64-bit v_lshl_add_u64 per slot and tile with flat pointers: the
  // compiler folds the lane offset into the pointer and re-adds the tile offset on the vector ALU)
  constexpr bool UNIFORM_SLOTS = (D % 16 == 0);
  const __amdgpu_buffer_rsrc_t rsrcK = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(kbase), 0, D * L * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcV = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(vbase), 0, D * L * 4, 0x00020000);
  unsigned lane_off[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int idx = tid + i * ATT_THREADS, row = idx >> 4, seg = idx & 15;
    lane_off[i] = (unsigned)((row < D ? row : row - D) * L + seg * 4) * 4u;      // bytes (< 2^32: 32 rows of L floats)
  }
  auto stage_load = [&](int t) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = tid + i * ATT_THREADS;
      if (NV4 % ATT_THREADS == 0 || idx < NV4) {
        const int row = idx >> 4, seg = idx & 15;
        if constexpr (UNIFORM_SLOTS) {
          const bool is_k = ((i * ATT_THREADS) >> 4) < D;                         // compile-time choice per slot
          stage[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(is_k ? rsrcK : rsrcV, lane_off[i],
                                                                                     t * (KT * 4), 0));
        } else {
          const float* src = (row < D ? kbase + (size_t)row * L : vbase + (size_t)(row - D) * L) + t * KT + seg * 4;
          stage[i] = *reinterpret_cast<const f32x4*>(src);
        }
      }
    }
  };
  auto stage_store = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = tid + i * ATT_THREADS;
      if (NV4 % ATT_THREADS == 0 || idx < NV4) {
        const int row = idx >> 4, seg = idx & 15;
        float* dst = (row < D) ? &sK[buf][row * KROW + seg * 4] : &sV[buf][(row - D) * KROW + seg * 4];
        *reinterpret_cast<f32x4*>(dst) = stage[i];
      }
    }
  };

  f32x4 O[MT][NQ];
  f32x4 negm4[NQ];       // -m1 of the wave's queries, splat over a 4-register tuple: the C operand that starts each QK^T chain
  f32x2 l_run[NQ];
#pragma unroll
  for (int qt = 0; qt < NQ; ++qt) {
    l_run[qt] = f32x2{0.f, 0.f};
    negm4[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) O[mt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // P of the query tile in flight; its row sums are added one query tile LATE (see softmax), so it outlives do_tile
  f32x4 P[4];
  auto rowsum = [&](int qt) {
    // the tile's row sums as PACKED adds on register-quad halves: v_pk_add_f32 as inline asm because the compiler scalarises
    // these vector adds (34 scalar + 19 packed adds per 64x64 scores in the element-wise form; every VALU instruction is
    // issue time taken from the fp32 MFMA stream).  A tree, not a chain: the compiler puts one wait state between an asm
    // statement and an asm statement reading its result, and the tree has two such adjacent pairs where the chain had seven.
    const f32x2 a = pk_add(__builtin_shufflevector(P[0], P[0], 0, 1), __builtin_shufflevector(P[0], P[0], 2, 3));
    const f32x2 b = pk_add(__builtin_shufflevector(P[1], P[1], 0, 1), __builtin_shufflevector(P[1], P[1], 2, 3));
    const f32x2 c = pk_add(__builtin_shufflevector(P[2], P[2], 0, 1), __builtin_shufflevector(P[2], P[2], 2, 3));
    const f32x2 d = pk_add(__builtin_shufflevector(P[3], P[3], 0, 1), __builtin_shufflevector(P[3], P[3], 2, 3));
    const f32x2 ab = pk_add(a, b), cd = pk_add(c, d);
    l_run[qt] = pk_add(l_run[qt], pk_add(ab, cd));    // packed running sums, folded after the last tile
  };
  auto do_tile = [&](auto first_tag, int buf) {
    constexpr bool FIRST = decltype(first_tag)::value;
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
    f32x4 S[2][4];
    auto qk_mfma = [&](int qt, int i) {
      const int s = i >> 2, ks = i & 3;
      // the chain starts from -m1 (zero on the first tile), so the accumulator already holds s - m1
      S[qt & 1][ks] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[ks][s], qf[qt][s], s == 0 ? negm4[qt] : S[qt & 1][ks], 0, 0, 0);
    };
    auto pv_mfma = [&](int qt, int i) {
      const int mt = i % MT, j = i / MT, ks = j >> 2, r = j & 3;
      O[mt][qt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[ks][mt][r], P[ks][r], O[mt][qt], 0, 0, 0);
    };
    auto softmax = [&](int qt) {
      f32x4(&Sq)[4] = S[qt & 1];
      if (FIRST) {
        float tm = fmaxf(fmaxf(Sq[0][0], Sq[0][1]), fmaxf(Sq[0][2], Sq[0][3]));
#pragma unroll
        for (int ks = 1; ks < 4; ++ks) tm = fmaxf(tm, fmaxf(fmaxf(Sq[ks][0], Sq[ks][1]), fmaxf(Sq[ks][2], Sq[ks][3])));
        tm = fmaxf(tm, __shfl_xor(tm, 16, 64));
        tm = fmaxf(tm, __shfl_xor(tm, 32, 64));
        const float nm = -tm;
        negm4[qt] = f32x4{nm, nm, nm, nm};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) Sq[ks] += negm4[qt];
      }
      // Order per query tile: [all MFMAs issued so far] | row sums of the PREVIOUS query tile's P | the 16 exps | MFMAs.
      // * The MFMAs stay above: left alone, the compiler hoists each key subtile's exps to just behind the last MFMA of that
      //   subtile's chain and pays the MFMA -> VALU result hazard in s_nop (two s_nop 8 and three s_nop 4-5 per query tile).
      // * The previous tile's row sums (its P is still in registers; they read no MFMA result) fill what is left of that
      //   hazard window, so no s_nop is issued at all, and they are nowhere near a v_exp_f32 writing their sources: gfx950
      //   needs one wait state between a transcendental and a VALU instruction reading its result, and the compiler's hazard
      //   recogniser does not look inside asm statements (tests/test_host_cpu.py disassembles the library and checks that).
      if (!FIRST || qt > 0) {
        __builtin_amdgcn_sched_barrier(0);
        rowsum((qt + NQ - 1) % NQ);
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        P[ks] = f32x4{__builtin_amdgcn_exp2f(Sq[ks][0]), __builtin_amdgcn_exp2f(Sq[ks][1]), __builtin_amdgcn_exp2f(Sq[ks][2]),
                      __builtin_amdgcn_exp2f(Sq[ks][3])};
      __builtin_amdgcn_sched_barrier(0);
    };
#pragma unroll
    for (int i = 0; i < N_QK; ++i) qk_mfma(0, i);
#pragma unroll
    for (int qt = 0; qt < NQ; ++qt) {
      softmax(qt);
      // P.V(qt) alternates with QK^T(qt + 1), which leads by LEAD slots, and the last LEAD MFMAs of the group are fenced off
      // as P.V's: LEAD MFMAs plus the 8 row-sum adds then separate the last QK^T MFMA from the first exp that reads a score
      // -- the 11 wait states of the MFMA -> VALU result hazard without a single s_nop.
      constexpr int LEAD = 3;
      static_assert(N_QK <= N_PV, "the QK^T chain is folded into the P.V slots");
      if (qt + 1 < NQ) {
#pragma unroll
        for (int i = 0; i < LEAD; ++i) qk_mfma(qt + 1, i);
      }
#pragma unroll
      for (int i = 0; i < N_PV; ++i) {
        if (i == N_PV - LEAD) __builtin_amdgcn_sched_barrier(0);
        pv_mfma(qt, i);
        if (qt + 1 < NQ && i + LEAD < N_QK) qk_mfma(qt + 1, i + LEAD);
      }
    }
  };

  stage_load(0);
  stage_store(0);
  __syncthreads();
  // the first tile (which fixes the reference point) is peeled, so the steady-state loop body is branch-free
  stage_load(ntiles > 1 ? 1 : 0);
  do_tile(std::true_type{}, 0);
  stage_store(1);
  __syncthreads();
  // two tiles per iteration, so that the LDS buffer index is a compile-time constant: every LDS address of the body is a
  // fixed base register + immediate (with a run-time buffer index the compiler re-bases them on the vector ALU, ten
  // v_add_u32 per tile at d_head 16 -- issue time the fp32 MFMA stream cannot overlap)
  auto step = [&](int t, auto buf_tag) {
    constexpr int buf = decltype(buf_tag)::value;
    stage_load((t + 1 < ntiles) ? t + 1 : t);
    do_tile(std::false_type{}, buf);
    stage_store(buf ^ 1);
    __syncthreads();
  };
  int t = 1;
  for (; t + 1 < ntiles; t += 2) {
    step(t, std::integral_constant<int, 1>{});
    step(t + 1, std::integral_constant<int, 0>{});
  }
  if (t < ntiles) step(t, std::integral_constant<int, 1>{});
  rowsum(NQ - 1);      // the last query tile's sums of the last key tile

  float* obase = out + ((size_t)b * C + (size_t)head * D) * L;
#pragma unroll
  for (int qt = 0; qt < NQ; ++qt) {
    float lt = l_run[qt].x + l_run[qt].y;
    lt += __shfl_xor(lt, 16, 64);
    lt += __shfl_xor(lt, 32, 64);
    const bool bad = !(lt < FAST_OVERFLOW_LIMIT);            // overflow (or NaN): hand this query block to the safe kernel
    const float inv = bad ? __builtin_nanf("") : 1.0f / lt;
    const int q = qblk0 + qt * 16 + i16;
    if (lse2 != nullptr && q < L && g == 0)
      lse2[((size_t)b * gridDim.y + head) * L + q] = bad ? __builtin_nanf("") : __builtin_amdgcn_logf(lt) - negm4[qt][0];
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

template <int D, int NQ>
void launch_v(const float* qkv, float* o, float* lse2, int B, int C, int heads, int L, float qscale, int check,
              hipStream_t stream) {
  dim3 grid(cdiv(L, 64 * NQ), heads, B);
  hipLaunchKernelGGL((mha_flash_fwd_kernel<D, NQ>), grid, dim3(ATT_THREADS), 0, stream, qkv, o, lse2, C, L, qscale, check);
}

template <int D>
int launch_d(const float* qkv, float* o, float* lse2, int B, int C, int heads, int L, void* ws, int64_t ws_bytes,
             hipStream_t stream) {
  const float qscale = 1.4426950408889634f / sqrtf((float)D);
  (void)hipGetLastError();  // drop any stale error left by another HIP user in this thread
  if constexpr (D >= 48) {
    // d_head 48 / 64 (ch_mult containing 3 or 4 at ch = 128: C = 384 / 512): the running-max kernel with one query tile per
    // wave -- the register budget of the wider tiles does not stretch to 12-16 k-steps; not shapes of the default model
    launch_v<D, 1>(qkv, o, lse2, B, C, heads, L, qscale, 0, stream);
  } else {
    int nq = (L >= 512) ? 4 : 1;   // 4 query tiles per wave: 3 waves per SIMD at d_head 16, 2 at d_head 32 (8 tiles measured no faster)
    // d_head 32 on the fixed-reference kernel: 134 TFLOP/s against 120 on the running-max kernel (L = 16 384, batch 16)
    if (nq == 4 && contraction_mode() == HDIFF_CONTRACT_BF16X3 &&
        (launch_mha_fwd_h2(qkv, o, lse2, B, C, heads, L, qscale, ws, ws_bytes, stream) ||       // d_head 16, fp16 pairs (needs the workspace)
         launch_mha_fwd_x3p(qkv, o, lse2, B, C, heads, L, qscale, ws, ws_bytes, stream) ||      // d_head 32, fp16 pairs (needs the workspace)
         launch_mha_fwd_x3(qkv, o, lse2, B, C, heads, L, qscale, stream))) {                    // bf16 triples split in the loop
      // split-bf16 kernel (attention_x3.hip), same fixed-reference protocol: overflow-proof fp32 kernel in check mode behind it
      static const bool skip_check = getenv("HDIFF_NO_CHECK_PASS") != nullptr;      // dev knob: look at the poisoned rows
      if (!skip_check) launch_v<D, 4>(qkv, o, lse2, B, C, heads, L, qscale, 1, stream);
    } else if (nq == 4 && L % KT == 0) {
      // fixed-reference fast kernel, then the overflow-proof kernel in check mode (exits at once unless flagged)
      dim3 grid(cdiv(L, 256), heads, B);
      hipLaunchKernelGGL((mha_flash_fwd_fast_kernel<D, 4>), grid, dim3(ATT_THREADS), 0, stream, qkv, o, lse2, C, L, qscale);
      launch_v<D, 4>(qkv, o, lse2, B, C, heads, L, qscale, 1, stream);
    } else if (nq >= 8) launch_v<D, 8>(qkv, o, lse2, B, C, heads, L, qscale, 0, stream);
    else if (nq >= 4) launch_v<D, 4>(qkv, o, lse2, B, C, heads, L, qscale, 0, stream);
    else launch_v<D, 1>(qkv, o, lse2, B, C, heads, L, qscale, 0, stream);
  }
  HDIFF_CHECK_LAUNCH("mha_flash_fwd_kernel");
  return HDIFF_OK;
}

}  // namespace

static int mha_flash_fwd_any(const float* qkv, float* o, float* lse2, int B, int C, int heads, int L, void* ws,
                             int64_t ws_bytes, hipStream_t s) {
  HDIFF_CHECK_ARG(qkv && o, "mha_flash_fwd: null pointer");
  HDIFF_CHECK_ARG(B > 0 && L > 0 && heads > 0 && C % heads == 0, "mha_flash_fwd: bad sizes B=%d C=%d heads=%d L=%d", B, C,
                  heads, L);
  const int D = C / heads;
  switch (D) {
    case 4: return launch_d<4>(qkv, o, lse2, B, C, heads, L, ws, ws_bytes, s);
    case 8: return launch_d<8>(qkv, o, lse2, B, C, heads, L, ws, ws_bytes, s);
    case 12: return launch_d<12>(qkv, o, lse2, B, C, heads, L, ws, ws_bytes, s);
    case 16: return launch_d<16>(qkv, o, lse2, B, C, heads, L, ws, ws_bytes, s);
    case 24: return launch_d<24>(qkv, o, lse2, B, C, heads, L, ws, ws_bytes, s);
    case 32: return launch_d<32>(qkv, o, lse2, B, C, heads, L, ws, ws_bytes, s);
    case 48: return launch_d<48>(qkv, o, lse2, B, C, heads, L, ws, ws_bytes, s);
    case 64: return launch_d<64>(qkv, o, lse2, B, C, heads, L, ws, ws_bytes, s);
    default: break;
  }
  hdiff::set_error("mha_flash_fwd: head dim %d not in {4, 8, 12, 16, 24, 32, 48, 64}", D);
  return HDIFF_ERR_INVALID;
}

extern "C" int hdiff_mha_flash_fwd(const float* qkv, float* o, float* lse2, int B, int C, int heads, int L,
                                   hdiff_stream_t stream) {
  return mha_flash_fwd_any(qkv, o, lse2, B, C, heads, L, nullptr, 0, (hipStream_t)stream);
}

extern "C" int hdiff_mha_flash_fwd_workspace(int B, int C, int heads, int L, int64_t* bytes_out) {
  HDIFF_CHECK_ARG(bytes_out, "mha_flash_fwd_workspace: null pointer");
  HDIFF_CHECK_ARG(B > 0 && L > 0 && heads > 0 && C % heads == 0, "mha_flash_fwd_workspace: bad sizes B=%d C=%d heads=%d L=%d", B,
                  C, heads, L);
  *bytes_out = mha_fwd_x3p_workspace(B, C, heads, L);      // a function of the shape only, whatever the contraction mode is now
  return HDIFF_OK;
}

extern "C" int hdiff_mha_flash_fwd_ws(const float* qkv, float* o, float* lse2, int B, int C, int heads, int L, void* ws,
                                      int64_t ws_bytes, hdiff_stream_t stream) {
  HDIFF_CHECK_ARG(ws_bytes >= 0 && (ws != nullptr || ws_bytes == 0), "mha_flash_fwd_ws: workspace pointer / size mismatch");
  return mha_flash_fwd_any(qkv, o, lse2, B, C, heads, L, ws, ws_bytes, (hipStream_t)stream);
}
