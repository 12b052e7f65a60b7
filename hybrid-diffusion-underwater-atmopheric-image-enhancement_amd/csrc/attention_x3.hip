// Flash attention forward with fp32 operands carried as three bf16 pieces on the bf16 matrix core (gfx950
// v_mfma_f32_16x16x32_bf16), fp32 accumulation.  Same contract and data layout as attention.hip (reference:
// nn.MultiheadAttention core, ModelCondition.py:189, 204-208).
//
// Why: tools/mfma_valu.hip / tools/mfma_bf16_valu.hip measured that the fp32-input MFMA runs at the vector FMA rate and
// does not overlap with VALU work, while the bf16 MFMA is 16x faster per product AND executes concurrently with the VALU.
// An fp32 number is exactly x = x0 + x1 + x2 with xi = bf16(x - sum of earlier pieces) (3 x 8 significand bits + signs),
// so  a*b = sum_{i+j<=2} ai*bj  up to 3 * 2^-24 relative: six bf16 products, each exact in the fp32 accumulator.  That is
// the same error class as an fp32 FMA chain (checked against float64 in tests/test_gpu_ops.py), for 6/16 of the matrix
// time, and the softmax exp / split work runs underneath it.
//
//   QK^T:  the six (k_i, q_j) terms are laid along the MFMA's 32-wide contraction: d_head 16 -> two terms per MFMA
//          (3 MFMAs per 16x16 score tile), d_head 32 -> one term per MFMA (6).  -m1 is the chain's initial accumulator.
//   P.V :  P = exp2(S) is split in registers (v_cvt_pk_bf16_f32 + v_dot2c_f32_bf16 remainders: 3.5 VALU per score);
//          two 16-key score tiles form the 32 contraction slots of one MFMA; V pieces come pre-split from LDS.
//   K, V:  split once per tile by the staging threads (not per wave), stored as bf16 [piece][key][d] / [piece][d][key].
// Softmax reference point, overflow poisoning and the safe second pass are those of mha_flash_fwd_fast_kernel.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

using namespace hdiff;

namespace {

#ifndef X3_VALU_PER_STAGE
#define X3_VALU_PER_STAGE 80
#endif
constexpr int KT = 64;
constexpr int THREADS = 256;
constexpr float OVERFLOW_LIMIT = 1.2379400e27f;   // 2^90, as in attention.hip

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// (a, b) -> three packed bf16 pairs with a = a0 + a1 + a2 exactly (b likewise): each piece is the top 16 bits of what is
// left (8 significand bits, truncated), the remainders are exact fp32 subtractions.  Only plain VALU instructions
// (v_and, v_sub, v_perm): v_dot2c_f32_bf16 and the packed-fp32 instructions would be fewer, but tools/mfma_bf16_coexec.hip
// shows that those stall against the bf16 MFMA stream instead of running beside it.
__device__ __forceinline__ unsigned pack_hi16(float lo, float hi) {
  return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, hi), __builtin_bit_cast(unsigned, lo), 0x07060302u);
}
__device__ __forceinline__ float top16(float x) {
  return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & 0xffff0000u);
}
__device__ __forceinline__ void split3(float a, float b, unsigned& h0, unsigned& h1, unsigned& h2) {
  h0 = pack_hi16(a, b);
  const float ra = a - top16(a), rb = b - top16(b);
  h1 = pack_hi16(ra, rb);
  const float sa = ra - top16(ra), sb = rb - top16(rb);
  h2 = pack_hi16(sa, sb);
}

// one v_add_f32 the vectoriser cannot pair into v_pk_add_f32 (which stalls against the bf16 MFMA stream like the other
// packed-fp32 forms: tools/valu_rates.hip -- 2.5 cycles for the plain add, 4.6 + a stall for the packed one)
__device__ __forceinline__ float add1(float a, float b) {
  float r;
  asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// the same for operands that come straight out of v_exp_f32: gfx950 forwards a transcendental result to the next vector
// instruction only after one wait state, and the compiler's hazard pass does not look inside inline asm -- without the
// s_nop the add reads garbage (found the hard way: tests/test_gpu_ops.py::test_flash_attention_core)
__device__ __forceinline__ float add1_after_trans(float a, float b) {
  float r;
  asm("s_nop 0\n\tv_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

__device__ __forceinline__ f32x4 mfma_bf16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// split-product terms kept (piece of K or V, piece of Q or P): all i + j <= 2
__device__ constexpr int TERM_A[6] = {0, 1, 0, 2, 1, 0};
__device__ constexpr int TERM_B[6] = {0, 0, 1, 0, 1, 2};

// Round 5: this kernel serves the calls WITHOUT a workspace (hdiff_mha_flash_fwd; the engine always passes one, and then the
// fp16-pair kernels of attention_h2.hip / attention_x3p.hip run): fp32 operands in, split into bf16 triples by the staging threads
// of every workgroup.  Rounds 3 / 4 also fed it pre-split operands from the workspace (PRE; git show 6a5a4b1 has that form).
template <int D, int NQ>
__global__ __launch_bounds__(THREADS, 2) void mha_flash_fwd_x3_kernel(const float* __restrict__ qkv, const __bf16* __restrict__ ws,
                                                                      float* __restrict__ out, float* __restrict__ lse2, int C,
                                                                      int L, float qscale) {
  (void)ws;
  static_assert(D == 16 || D == 32, "head dim");
  constexpr int TPM = 32 / D;              // terms per QK^T MFMA
  constexpr int NQK = 6 / TPM;             // QK^T MFMAs per 16x16 score tile
  constexpr int MT = D / 16;               // 16-row tiles of the output
  constexpr int KROWB = D * 2;             // bytes per key of one K piece
  constexpr int KPART = KT * KROWB;
  constexpr int VROWB = KT * 2 + 8;        // bytes per d row of one V piece (+8: the 16 rows of an operand read spread over banks)
  constexpr int VPART = D * VROWB;
  constexpr int DK = D / 4;                // K floats staged per thread
  constexpr int NVL = D / 16;              // V float4 staged per thread
  constexpr int QB = 64 * NQ;              // queries per workgroup (4 waves x NQ tiles of 16)

  constexpr int VBASE = 3 * KPART;
  constexpr int BUFB = VBASE + 3 * VPART;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2][BUFB];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  const TileId tile = xcd_tile();
  const int head = tile.head, b = tile.b;
  const int qblk0 = tile.x * QB + wave * (16 * NQ);
  const float* qbase = qkv + ((size_t)b * 3 * C + (size_t)head * D) * L;
  const float* kbase = qbase + (size_t)C * L;
  const float* vbase = kbase + (size_t)C * L;
  const int ntiles = L / KT;

  // contraction slots of this lane: 8 consecutive d of one term
  const int doff = (TPM == 2) ? 8 * (g & 1) : 8 * g;
  const bool hi = (TPM == 2) && (g >> 1);

  // Q operands: pre-scaled (exp2 domain), split once
  u32x4 qop[NQ][NQK];
#pragma unroll
  for (int qt = 0; qt < NQ; ++qt) {
    const int q = qblk0 + qt * 16 + i16;
    u32x4 piece[3];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float a = (q < L) ? qbase[(size_t)(doff + 2 * j) * L + q] * qscale : 0.f;
      const float c = (q < L) ? qbase[(size_t)(doff + 2 * j + 1) * L + q] * qscale : 0.f;
      unsigned h0, h1, h2;
      split3(a, c, h0, h1, h2);
      piece[0][j] = h0; piece[1][j] = h1; piece[2][j] = h2;
    }
#pragma unroll
    for (int j = 0; j < NQK; ++j) {
      if (TPM == 2) qop[qt][j] = hi ? piece[TERM_B[2 * j + 1]] : piece[TERM_B[2 * j]];
      else qop[qt][j] = piece[TERM_B[j]];
    }
  }
  // K operand addresses (bytes inside one buffer): piece chosen per lane half at d_head 16
  int kaddr[NQK];
#pragma unroll
  for (int j = 0; j < NQK; ++j) {
    const int piece = (TPM == 2) ? (hi ? TERM_A[2 * j + 1] : TERM_A[2 * j]) : TERM_A[j];
    kaddr[j] = piece * KPART + i16 * KROWB + doff * 2;
  }
  const int vaddr = i16 * VROWB + 8 * g;

  // staging: K thread = (key, group of DK d), V thread = (d, 4 keys), split into bf16 triples while storing
  const int skey = tid & 63, sdg = tid >> 6;
  const int sd = tid >> 4, sseg = tid & 15;
  float kst[DK];
  f32x4 vst[NVL];
  auto stage_load = [&](int t) {
#pragma unroll
    for (int j = 0; j < DK; ++j) kst[j] = kbase[(size_t)(sdg * DK + j) * L + t * KT + skey];
#pragma unroll
    for (int i = 0; i < NVL; ++i)
      vst[i] = *reinterpret_cast<const f32x4*>(vbase + (size_t)(sd + 16 * i) * L + t * KT + 4 * sseg);
  };
  auto stage_store = [&](int buf) {
    unsigned kp[3][DK / 2];
#pragma unroll
    for (int j = 0; j < DK / 2; ++j) split3(kst[2 * j], kst[2 * j + 1], kp[0][j], kp[1][j], kp[2][j]);
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      unsigned char* dst = &smem[buf][p * KPART + skey * KROWB + sdg * DK * 2];
      if (DK == 4) *reinterpret_cast<u32x2*>(dst) = u32x2{kp[p][0], kp[p][1]};
      else *reinterpret_cast<u32x4*>(dst) = u32x4{kp[p][0], kp[p][1], kp[p][2 % (DK / 2)], kp[p][3 % (DK / 2)]};
    }
#pragma unroll
    for (int i = 0; i < NVL; ++i) {
      unsigned a0, a1, a2, b0, b1, b2;
      split3(vst[i][0], vst[i][1], a0, a1, a2);
      split3(vst[i][2], vst[i][3], b0, b1, b2);
      unsigned char* dst = &smem[buf][VBASE + (sd + 16 * i) * VROWB + sseg * 8];
      *reinterpret_cast<u32x2*>(dst) = u32x2{a0, b0};
      *reinterpret_cast<u32x2*>(dst + VPART) = u32x2{a1, b1};
      *reinterpret_cast<u32x2*>(dst + 2 * VPART) = u32x2{a2, b2};
    }
  };

  f32x4 O[MT][NQ];
  f32x4 negm4[NQ];
  f32x2 l_run[NQ];
#pragma unroll
  for (int qt = 0; qt < NQ; ++qt) {
    l_run[qt] = f32x2{0.f, 0.f};
    negm4[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) O[mt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  auto do_tile = [&](auto first_tag, int buf) {
    constexpr bool FIRST = decltype(first_tag)::value;
    const unsigned char* kb = smem[buf];
    const unsigned char* vb = smem[buf] + VBASE;
    // V operands of the tile: [piece][row tile][32-key chunk]
    u32x4 vop[3][MT][2];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const unsigned char* src = vb + p * VPART + mt * 16 * VROWB + 64 * c + vaddr;
          const u32x2 lo = *reinterpret_cast<const u32x2*>(src);
          const u32x2 hi2 = *reinterpret_cast<const u32x2*>(src + 32);
          vop[p][mt][c] = u32x4{lo[0], lo[1], hi2[0], hi2[1]};
        }
    // K operands: [key tile][MFMA]
    u32x4 kop[4][NQK];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int j = 0; j < NQK; ++j) kop[kt][j] = *reinterpret_cast<const u32x4*>(kb + kaddr[j] + kt * 16 * KROWB);

    // Software pipeline over the wave's query tiles: while the VALU turns S(qt) into split P(qt), the matrix core runs
    // QK^T(qt+1) and P.V(qt-1) -- the bf16 MFMA executes concurrently with VALU work, but a wave issues in order, so the
    // independent MFMAs have to sit between the VALU instructions in program order.
    f32x4 S[2][4];
    u32x4 pop[2][3][2];
    auto qk = [&](int qt) {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        f32x4 acc = negm4[qt];     // the chain starts from -m1 (zero on the first tile): the accumulator holds s - m1
#pragma unroll
        for (int j = 0; j < NQK; ++j) acc = mfma_bf16(kop[kt][j], qop[qt][j], acc);
        S[qt & 1][kt] = acc;
      }
    };
    auto pv = [&](int qt) {
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int term = 5; term >= 0; --term)     // small terms first
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            O[mt][qt] = mfma_bf16(vop[TERM_A[term]][mt][c], pop[qt & 1][TERM_B[term]][c], O[mt][qt]);
    };
    auto softmax_split = [&](int qt) {
      f32x4(&Sq)[4] = S[qt & 1];
      if (FIRST) {
        float tm = fmaxf(fmaxf(Sq[0][0], Sq[0][1]), fmaxf(Sq[0][2], Sq[0][3]));
#pragma unroll
        for (int kt = 1; kt < 4; ++kt) tm = fmaxf(tm, fmaxf(fmaxf(Sq[kt][0], Sq[kt][1]), fmaxf(Sq[kt][2], Sq[kt][3])));
        tm = fmaxf(tm, __shfl_xor(tm, 16, 64));
        tm = fmaxf(tm, __shfl_xor(tm, 32, 64));
        const float nm = -tm;
        negm4[qt] = f32x4{nm, nm, nm, nm};
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) Sq[kt] += negm4[qt];
      }
      float sum0 = 0.f, sum1 = 0.f;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        const f32x2 pa = {__builtin_amdgcn_exp2f(Sq[kt][0]), __builtin_amdgcn_exp2f(Sq[kt][1])};
        const f32x2 pc = {__builtin_amdgcn_exp2f(Sq[kt][2]), __builtin_amdgcn_exp2f(Sq[kt][3])};
        sum0 = add1(sum0, add1_after_trans(pa[0], pc[0]));      // scalar adds on purpose (see add1)
        sum1 = add1(sum1, add1_after_trans(pa[1], pc[1]));
        const int c = kt >> 1, o = (kt & 1) * 2;
        unsigned a0, a1, a2, c0, c1, c2;
        split3(pa[0], pa[1], a0, a1, a2);
        split3(pc[0], pc[1], c0, c1, c2);
        pop[qt & 1][0][c][o] = a0; pop[qt & 1][1][c][o] = a1; pop[qt & 1][2][c][o] = a2;
        pop[qt & 1][0][c][o + 1] = c0; pop[qt & 1][1][c][o + 1] = c1; pop[qt & 1][2][c][o + 1] = c2;
      }
      l_run[qt][0] = add1(l_run[qt][0], sum0);
      l_run[qt][1] = add1(l_run[qt][1], sum1);
    };
    qk(0);
    auto stage = [&](auto qt_tag) {
      constexpr int qt = decltype(qt_tag)::value;
      if (qt + 1 < NQ) qk(qt + 1);
      if (qt > 0) pv(qt - 1);
      softmax_split(qt);
      if (!FIRST) {
        // pin the interleave: one MFMA, then its share of the stage's VALU instructions (the scheduler otherwise clumps
        // the MFMAs, which costs about 9% -- measured)
        constexpr int nm = ((qt + 1 < NQ) ? 4 * NQK : 0) + ((qt > 0) ? 12 * MT : 0);
        constexpr int per = X3_VALU_PER_STAGE / (nm > 0 ? nm : 1);
#pragma unroll
        for (int i = 0; i < nm; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, per, 0);
        }
      }
    };
    stage(std::integral_constant<int, 0>{});
    stage(std::integral_constant<int, 1>{});
    if constexpr (NQ > 2) {
      stage(std::integral_constant<int, 2>{});
      stage(std::integral_constant<int, 3>{});
    }
    pv(NQ - 1);
  };

  stage_load(0);
  stage_store(0);
  __syncthreads();
  stage_load(ntiles > 1 ? 1 : 0);
  do_tile(std::true_type{}, 0);
  stage_store(1);
  __syncthreads();
  for (int t = 1; t < ntiles; ++t) {
    const int buf = t & 1;
    stage_load((t + 1 < ntiles) ? t + 1 : t);
    do_tile(std::false_type{}, buf);
    stage_store(buf ^ 1);
    __syncthreads();
  }

  float* obase = out + ((size_t)b * C + (size_t)head * D) * L;
#pragma unroll
  for (int qt = 0; qt < NQ; ++qt) {
    float lt = l_run[qt][0] + l_run[qt][1];
    lt += __shfl_xor(lt, 16, 64);
    lt += __shfl_xor(lt, 32, 64);
    const bool bad = !(lt < OVERFLOW_LIMIT);            // overflow (or NaN): hand this query block to the safe kernel
    const float inv = bad ? __builtin_nanf("") : 1.0f / lt;
    const int q = qblk0 + qt * 16 + i16;
    if (lse2 != nullptr && q < L && g == 0)
      lse2[((size_t)b * gridDim.y + head) * L + q] = bad ? __builtin_nanf("") : __builtin_amdgcn_logf(lt) - negm4[qt][0];
    if (q < L) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) obase[(size_t)(mt * 16 + 4 * g + r) * L + q] = O[mt][qt][r] * inv;
    }
  }
}

}  // namespace

namespace hdiff {

// Launches the split-bf16 kernel for (d_head, L) it supports; returns false if this shape is not covered.
bool launch_mha_fwd_x3(const float* qkv, float* o, float* lse2, int B, int C, int heads, int L, float qscale, hipStream_t stream) {
  const int D = C / heads;
  if (L % KT != 0 || L < 512) return false;
  if (D == 16) {
    hipLaunchKernelGGL((mha_flash_fwd_x3_kernel<16, 4>), dim3(cdiv(L, 256), heads, B), dim3(THREADS), 0, stream, qkv, nullptr, o, lse2, C, L, qscale);
    return true;
  }
  if (D == 32) {
    hipLaunchKernelGGL((mha_flash_fwd_x3_kernel<32, 2>), dim3(cdiv(L, 128), heads, B), dim3(THREADS), 0, stream, qkv, nullptr, o, lse2, C, L, qscale);
    return true;
  }
  return false;
}

}  // namespace hdiff
