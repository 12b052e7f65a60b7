// Flash attention forward on PRE-SPLIT operands: every fp32 value of Q, K, V is written once, by a small streaming kernel,
// as three bf16 pieces (x = x0 + x1 + x2 exactly, see attention_x3.hip), and the attention kernel contracts the pieces on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  Same contract as attention.hip / attention_x3.hip (reference:
// nn.MultiheadAttention core, ModelCondition.py:189, 204-208); same fixed softmax reference point, overflow poisoning and
// check pass.
//
// What the pre-split buys (measured at L = 65 536 / 16 384, tools/x3_check.py, split pass included): the K / V / Q splits
// leave the loop -- 256 workgroups of a (head, sample) pair used to repeat them -- and with them about 10 % of its vector
// instructions, which is what bounds this formulation: per score and lane 2 exp slots + 5.5 split / pack instructions + the
// row sum, about 34 VALU cycles, against 24 cycles of bf16 MFMA, and on gfx950 the two overlap only by a few vector
// instructions per MFMA (tools/mfma_bf16_coexec32.hip: beside one 32-cycle MFMA about five plain VALU instructions are
// free, each further one costs its full 4-5 cycles, whatever the MFMA shape and however many waves share the SIMD).
// Two kernels read the workspace:
//   d_head 32: the 32x32x16 kernel below -- M = 32 = d, every MFMA row useful, a quarter fewer MFMA issue slots than the
//              16x16x32 form: 217-220 TFLOP/s fp32-equivalent against 197 for the kernel that splits in its loop;
//   d_head 16: the 16x16x32 kernel of attention_x3.hip with PRE = true (185 against 177).  This file's kernel also runs
//              d_head 16 (HDIFF_X3P=16), where a 32-row tile carries TWO V pieces; it then spends 8 MFMA slots on 6
//              products and measured 162 (straight order) / 174 (QK of the next unit and PV of the previous one software-
//              pipelined beside the exp / split stream, one basic block per tile: removed again, it did not pay).
//
//   workspace (bf16), per (sample, head):  Qs[3][L][D] (pre-scaled into the exp2 domain), Ks[3][L][D], Vs[3][D][L]
//   S^T = K Q^T : M = 32 keys, N = 32 queries, K = 16 of d; the six piece products accumulate in one chain that starts
//                 from -m (the fixed reference point), so the accumulator IS s - m.
//   O^T += V^T P: P = exp2(S^T) is split in registers; its accumulator layout (keys 8j + 4h + i on registers, queries on
//                 lanes) is the B-operand layout of the next MFMA up to a permutation of the contraction slots, which the
//                 V operand reads follow -- no cross-lane traffic.  d_head 16 fills only half of M = 32, so the two halves
//                 carry two different V pieces ([v0; v1] with p0 and with p1, [v2; 0] with p0, [v0; 0] with p2: 4 MFMAs
//                 per 16 keys, the (v1, p1) term comes for free) and are added once at the end; d_head 32 uses 6.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

using namespace hdiff;

namespace {

constexpr int KT = 64;
constexpr int THREADS = 256;
constexpr float OVERFLOW_LIMIT = 1.2379400e27f;   // 2^90, as in attention.hip

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pack_hi16(float lo, float hi) {
  return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, hi), __builtin_bit_cast(unsigned, lo), 0x07060302u);
}
__device__ __forceinline__ float top16(float x) {
  return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & 0xffff0000u);
}
// (a, b) -> three packed bf16 pairs, a = a0 + a1 + a2 exactly (truncation split; plain VALU only, see attention_x3.hip)
__device__ __forceinline__ void split3(float a, float b, unsigned& h0, unsigned& h1, unsigned& h2) {
  h0 = pack_hi16(a, b);
  const float ra = a - top16(a), rb = b - top16(b);
  h1 = pack_hi16(ra, rb);
  const float sa = ra - top16(ra), sb = rb - top16(rb);
  h2 = pack_hi16(sa, sb);
}

__device__ __forceinline__ f32x16 mfma32(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32h(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// (a, b) -> two packed fp16 pairs, a = h0.lo + h1.lo up to 2^-23 |a| (or 2^-25 absolute); attention_h2.hip.  `one` is 1.0f in a
// register the compiler cannot see through (the residual must stay an fma: v_fma_mixlo / mixhi_f16).
__device__ __forceinline__ void split2(float a, float b, float one, unsigned& h0, unsigned& h1) {
  const f16x2 p = {(_Float16)a, (_Float16)b};
  unsigned u = __builtin_bit_cast(unsigned, p);
  asm("" : "+v"(u));
  const f16x2 q = __builtin_bit_cast(f16x2, u);
  const f16x2 r = {(_Float16)__builtin_fmaf(a, one, -(float)q[0]), (_Float16)__builtin_fmaf(b, one, -(float)q[1])};
  h0 = u;
  h1 = __builtin_bit_cast(unsigned, r);
}
constexpr float P_SHIFT = 8.0f;       // PVH: the reference point enters as P = 2^8 ...
constexpr float P_TRIP = 32768.0f;    // ... and moves when a lane's 16 P values of one block sum to 2^15 (attention_h2.hip)

// split-product terms kept (piece of K or V, piece of Q or P): all i + j <= 2, small terms last in the table
__device__ constexpr int TERM_A[6] = {0, 1, 0, 2, 1, 0};
__device__ constexpr int TERM_B[6] = {0, 0, 1, 0, 1, 2};

// ---------------------------------------------------------------------------------------------------------------------
// fp32 qkv [B][3C][L]  ->  bf16 pieces in the workspace layout above.  Streaming: reads 4 bytes, writes 6 per element.
// grid (L / 256, 3 * heads, B), 256 threads.  Q and K: thread = one position, all D channels (reads coalesced over the
// threads, writes D * 2 contiguous bytes per thread and piece).  V: thread = two neighbouring positions of each channel.
// ---------------------------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(THREADS) void qkv_split3_kernel(const float* __restrict__ qkv, __bf16* __restrict__ ws, int C, int L,
                                                             float qscale) {
  const int heads = C / D;
  const int which = blockIdx.y / heads, head = blockIdx.y - which * heads, b = blockIdx.z;
  const float* src = qkv + ((size_t)b * 3 * C + (size_t)which * C + (size_t)head * D) * L;
  __bf16* dst = ws + ((size_t)b * heads + head) * 9 * (size_t)L * D + (size_t)which * 3 * L * D;
  const size_t piece = (size_t)L * D;
  if (which < 2) {
    const int l = blockIdx.x * THREADS + threadIdx.x;
    if (l >= L) return;
    const float sc = which == 0 ? qscale : 1.0f;
    unsigned h[3][D / 2];
#pragma unroll
    for (int j = 0; j < D / 2; ++j) {
      const float a = src[(size_t)(2 * j) * L + l] * sc, c = src[(size_t)(2 * j + 1) * L + l] * sc;
      split3(a, c, h[0][j], h[1][j], h[2][j]);
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      u32x4* o = reinterpret_cast<u32x4*>(dst + p * piece + (size_t)l * D);
#pragma unroll
      for (int j = 0; j < D / 8; ++j) o[j] = u32x4{h[p][4 * j], h[p][4 * j + 1], h[p][4 * j + 2], h[p][4 * j + 3]};
    }
  } else {
    const int l2 = blockIdx.x * THREADS + threadIdx.x;          // pair index: positions 2*l2, 2*l2 + 1
    if (2 * l2 >= L) return;
#pragma unroll 4
    for (int d = 0; d < D; ++d) {
      const f32x2 v = *reinterpret_cast<const f32x2*>(src + (size_t)d * L + 2 * l2);
      unsigned h0, h1, h2;
      split3(v[0], v[1], h0, h1, h2);
      unsigned* o = reinterpret_cast<unsigned*>(dst + (size_t)d * L) + l2;
      o[0] = h0;
      o[piece / 2] = h1;
      o[piece] = h2;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// PVH (round 4, d_head 32): P.V on fp16 PAIRS, as attention_h2.hip does at d_head 16 -- P = h0 + h1 by v_cvt_pk_f16_f32 +
// v_fma_mixlo / mixhi_f16 (1.5 vector instructions per value instead of 5.5), V as two fp16 pieces of V 2^s with s per channel
// row (launch_v_split_h2; O times 2^-s at the end), three products (v1 p0, v0 p1, v0 p0) instead of six; S = Q K^T stays on the
// bf16 triples.  fp16 ends at 65 504, so the softmax reference MOVES: it starts at (first block's maximum - 8) and whenever a
// lane's 16 P values of a block sum to 2^15 or more the wave recomputes that block from its S accumulator under a reference
// that puts the row's maximum over the block at 2^8, after scaling O and l of that query by the exact power of two.  Rows that
// do not need it get delta = 0 and the same bits.  Each of a lane's two queries has its OWN reference here (the bf16 form shares
// one: fp32's exponent range does not care, but a pair is precise only within 2^22 of fp16's top, and a query that inherits
// the reference of a neighbour with larger scores would carry its whole row as fp16 subnormals -- measured: 4e-4 instead of
// 8e-6); the reference enters each score chain as a freshly splat accumulator.  Plain control flow: this kernel's straight
// (query group, block) order has no hand-placed slots to protect.
// PIPE (PVH only): both query groups' score chains are issued before the first group's exp / split stream, so that a wave's own
// matrix work (the second group's 12 MFMAs, then the first group's P.V) runs beside its vector work instead of only the other
// wave's; the references of the two groups are independent (one per query), so the order is free.
// Round 5 (PVH): the scores on fp16 pairs with a balance per product term as well (attention_h2.hip has the scheme and its
// error analysis): K pieces k0, k0 2^-8, k1 2^8 (, k1), Q pieces q0, q1 2^8 from the workspace and q0 2^-8 (, q1) made in
// registers; NQT product terms (piece i of K against piece i of Q), each two k-steps of v_mfma_f32_32x32x16_f16.  NQT = 4 is
// the d_head 16 kernel's choice (error of S 1.02x the fp32 chain's); NQT = 3 drops k1 q1 (1.14x) and needs no more registers
// than the bf16 triples did.
#ifndef X3P_NQT
#define X3P_NQT 3
#endif
template <int D, bool PVH = false, bool PIPE = false>
__global__ __launch_bounds__(THREADS, 2) void mha_flash_fwd_x3p_kernel(const __bf16* __restrict__ ws, float* __restrict__ out,
                                                                      float* __restrict__ lse2, int C, int L, float one) {
  static_assert(D == 16 || D == 32, "head dim");
  static_assert(!PVH || D == 32, "the fp16-pair P.V form of this kernel is the d_head 32 one");
  constexpr int NPC = PVH ? 2 : 3;             // pieces of V and of P
  constexpr int NKP = PVH ? X3P_NQT : 3;       // K pieces staged (PVH: = score product terms)
  constexpr int NQT = PVH ? X3P_NQT : 6;       // score product terms
  constexpr int KS = D / 16;                   // k-steps of the QK^T product
  constexpr int NPV = (D == 16) ? 4 : 6;       // P.V MFMAs per 16 keys
  constexpr int KROWB = D * 2 + 16;            // bytes per key of one K piece in LDS (+16: conflict-free ds_read_b128)
  constexpr int KPART = KT * KROWB;
  constexpr int VROWB = KT * 2 + 8;            // bytes per d row of one V piece (+8: rows spread over the banks)
  constexpr int VPART = D * VROWB;
  constexpr int NKC = NKP * KT * D / 8;        // 16-byte chunks of a K tile (all pieces)
  constexpr int NVC = NPC * D * 8;             // 16-byte chunks of a V tile
  constexpr int NLD = (NKC + NVC) / THREADS;   // chunks per thread: 3 (d 16), 6 (d 32)
  static_assert((NKC + NVC) % THREADS == 0, "staging geometry");
  constexpr int QB = 256;                      // queries per workgroup: 4 waves x 2 groups of 32

  constexpr int VBASE = NKP * KPART;                                       // one buffer = K pieces, V pieces, one row of zeros
  constexpr int BUFB = (VBASE + NPC * VPART + VROWB + 15) / 16 * 16;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2][BUFB];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const TileId tile = xcd_tile();
  const int head = tile.head, b = tile.b;
  const int heads = gridDim.y;
  const int qblk0 = tile.x * QB + wave * 64;
  const size_t piece = (size_t)L * D;
  const __bf16* qs = ws + ((size_t)b * heads + head) * 9 * piece;
  const __bf16* ks = qs + (PVH ? 2 : 3) * piece;      // PVH: Q pieces 0, 1, K pieces 2 .. 5 (launch_qk_split_h2)
  const __bf16* vs = qs + 6 * piece;
  const int ntiles = L / KT;

  // zero row of both V buffers (read by the upper half of the one-piece V operands at d_head 16)
  if (tid < 2 * (VROWB / 4)) {
    const int bufi = tid / (VROWB / 4), w = tid - bufi * (VROWB / 4);
    *reinterpret_cast<unsigned*>(&smem[bufi][VBASE + NPC * VPART + 4 * w]) = 0u;
  }

  // Q operands (B of S^T = K Q^T): lane (query l31, half h) holds d = 16 ks + 8 h .. + 7 of each piece
  constexpr int NQP = PVH ? NQT : 3;           // Q pieces held
  u32x4 qop[2][NQP][KS];
#pragma unroll
  for (int G = 0; G < 2; ++G) {
    const int q = qblk0 + 32 * G + l31;
#pragma unroll
    for (int p = 0; p < (PVH ? 2 : 3); ++p)
#pragma unroll
      for (int s = 0; s < KS; ++s)
        qop[G][p][s] = *reinterpret_cast<const u32x4*>(qs + p * piece + (size_t)q * D + 16 * s + 8 * h);
    if constexpr (PVH) {
      // q0 2^-8 (and q1 = q1 2^8 2^-8): packed fp16 multiplications by 2^-8 (asm: see attention_h2.hip)
      const unsigned dn2 = 0x1c001c00u;
#pragma unroll
      for (int p = 2; p < NQP; ++p)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          unsigned w4[4];
#pragma unroll
          for (int w = 0; w < 4; ++w) asm("v_pk_mul_f16 %0, %1, %2" : "=v"(w4[w]) : "v"(qop[G][p - 2][s][w]), "v"(dn2));
          qop[G][p][s] = u32x4{w4[0], w4[1], w4[2], w4[3]};
        }
    }
  }

  // staging: chunk c = i * 256 + tid of the tile's 16-byte chunks (K pieces first, then V pieces)
  const unsigned char* gsrc[NLD];
  int lds_off[NLD];
  int gstep[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int c = i * THREADS + tid;
    if (c < NKC) {
      const int p = c / (KT * D / 8), rem = c - p * (KT * D / 8);
      const int key = rem / (D / 8), part = rem - key * (D / 8);
      gsrc[i] = reinterpret_cast<const unsigned char*>(ks + p * piece) + (size_t)rem * 16;
      lds_off[i] = p * KPART + key * KROWB + part * 16;
      gstep[i] = KT * D * 2;
    } else {
      const int cv = c - NKC;
      const int p = cv / (D * 8), rem = cv - p * (D * 8);
      const int d = rem >> 3, seg = rem & 7;
      gsrc[i] = reinterpret_cast<const unsigned char*>(vs + p * piece + (size_t)d * L) + seg * 16;
      lds_off[i] = VBASE + p * VPART + d * VROWB + seg * 16;
      gstep[i] = KT * 2;
    }
  }
  u32x4 stage[NLD];
  auto stage_load = [&](int t) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) stage[i] = *reinterpret_cast<const u32x4*>(gsrc[i] + (size_t)t * gstep[i]);
  };
  // two 8-byte stores per chunk for K and V alike (V rows are 8-byte aligned): no per-thread branch, so a whole tile of
  // the main loop stays ONE basic block and the scheduler can keep every unit's vector work beside its MFMAs
  auto stage_store = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      unsigned char* dst = &smem[buf][lds_off[i]];
      *reinterpret_cast<u32x2*>(dst) = u32x2{stage[i][0], stage[i][1]};
      *reinterpret_cast<u32x2*>(dst + 8) = u32x2{stage[i][2], stage[i][3]};
    }
  };

  // operand addresses inside a buffer
  const int kaddr = l31 * KROWB + 16 * h;                                  // + piece * KPART + key block * 32 * KROWB + ks * 32
  constexpr int NVK = (D == 16) ? 2 : NPC;                                 // V operand kinds (d_head 16) / pieces (d_head 32)
  int vaddr[NVK];
  if constexpr (D == 16) {
    const int d = l31 & 15;
    const bool up = l31 >= 16;
    vaddr[0] = (up ? VPART : 0) + d * VROWB + 8 * h;                       // [v0; v1]
    vaddr[1] = up ? 3 * VPART : 2 * VPART + d * VROWB + 8 * h;             // [v2; 0]
  } else {
#pragma unroll
    for (int p = 0; p < NPC; ++p) vaddr[p] = p * VPART + l31 * VROWB + 8 * h;
  }

  // negm16: the softmax reference point, negated and splat over an accumulator tuple = the C operand that starts every
  // QK^T chain.  ONE value per lane serves both of its queries (group 0 and group 1): the larger of their two first-block
  // maxima.  Any reference works as long as exp2 neither overflows nor flushes the whole row; both accidents end in a
  // NaN output and the check pass (attention.hip) recomputes that query block.
  f32x16 O[2], negm16;
  f32x2 l_run[2];
  float negm2[2] = {0.f, 0.f};          // PVH: -m of the lane's two queries
#pragma unroll
  for (int G = 0; G < 2; ++G) {
    l_run[G] = f32x2{0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 16; ++r) O[G][r] = 0.f;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) negm16[r] = 0.f;

  auto load_k = [&](int buf, int kb, u32x4 (&kop)[NKP][KS]) {
    const unsigned char* kbuf = smem[buf] + kb * 32 * KROWB + kaddr;
#pragma unroll
    for (int p = 0; p < NKP; ++p)
#pragma unroll
      for (int s = 0; s < KS; ++s) kop[p][s] = *reinterpret_cast<const u32x4*>(kbuf + p * KPART + 32 * s);
  };
  // V operands of the two 16-key halves of a 32-key block: contraction slot 8 h + 4 jj + i  <->  key 16 ab + 8 jj + 4 h + i
  auto load_v = [&](int buf, int kb, u32x4 (&vop)[2][NVK]) {
    const unsigned char* vbuf = smem[buf] + VBASE + kb * 64;               // 32 keys = 64 bytes along a V row
#pragma unroll
    for (int ab = 0; ab < 2; ++ab)
#pragma unroll
      for (int kind = 0; kind < NVK; ++kind) {
        const unsigned char* src = vbuf + vaddr[kind] + 32 * ab;
        const u32x2 lo = *reinterpret_cast<const u32x2*>(src);
        const u32x2 hi2 = *reinterpret_cast<const u32x2*>(src + 16);
        vop[ab][kind] = u32x4{lo[0], lo[1], hi2[0], hi2[1]};
      }
  };
  auto qk = [&](const u32x4 (&kop)[NKP][KS], int G, f32x16 c) {
    f32x16 S = c;                                  // the chain starts from -m: the accumulator holds s - m
    if constexpr (PVH) {                           // fp16 pairs: piece i of K against piece i of Q, large term first
#pragma unroll
      for (int term = 0; term < NQT; ++term)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          u32x4 qb = qop[G][term][s];
          if ((HDIFF_MUTANT & 2) && term == 1)       // (mutation test: the low five bits of q1 2^8 dropped: 2^-17 of q)
#pragma unroll
            for (int w = 0; w < 4; ++w) qb[w] &= 0xffe0ffe0u;
          S = mfma32h(kop[term][s], qb, S);
        }
      return S;
    }
#pragma unroll
    for (int term = 0; term < 6; ++term)
#pragma unroll
      for (int s = 0; s < KS; ++s)
        if (!((HDIFF_MUTANT & 2) && TERM_A[term] == 0 && TERM_B[term] == 2)) S = mfma32(kop[TERM_A[term]][s], qop[G][TERM_B[term]][s], S);
    return S;
  };
  auto softmax_split = [&](const f32x16& S, int G, u32x4 (&pop)[2][3]) {
    float sum0 = 0.f, sum1 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float p0 = __builtin_amdgcn_exp2f(S[4 * j]), p1 = __builtin_amdgcn_exp2f(S[4 * j + 1]);
      const float p2 = __builtin_amdgcn_exp2f(S[4 * j + 2]), p3 = __builtin_amdgcn_exp2f(S[4 * j + 3]);
      sum0 += p0 + p2;
      sum1 += p1 + p3;
      unsigned a0, a1, a2, c0, c1, c2;
      split3(p0, p1, a0, a1, a2);
      split3(p2, p3, c0, c1, c2);
      const int ab = j >> 1, o = (j & 1) * 2;
      pop[ab][0][o] = a0; pop[ab][1][o] = a1; pop[ab][2][o] = a2;
      pop[ab][0][o + 1] = c0; pop[ab][1][o + 1] = c1; pop[ab][2][o + 1] = c2;
    }
    l_run[G][0] += sum0;
    l_run[G][1] += sum1;
  };
  // PVH: P = exp2(S) of one (block, query group) as fp16 pairs + the lane's row sums; the reference moves when it must
  auto exp_pairs = [&](const f32x16& S, u32x4 (&pop)[2][3], float& sum0, float& sum1) {
    sum0 = 0.f; sum1 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float p0 = __builtin_amdgcn_exp2f(S[4 * j]), p1 = __builtin_amdgcn_exp2f(S[4 * j + 1]);
      const float p2 = __builtin_amdgcn_exp2f(S[4 * j + 2]), p3 = __builtin_amdgcn_exp2f(S[4 * j + 3]);
      sum0 += p0 + p2;
      sum1 += p1 + p3;
      unsigned a0, a1, c0, c1;
      split2(p0, p1, one, a0, a1);
      split2(p2, p3, one, c0, c1);
      const int ab = j >> 1, o = (j & 1) * 2;
      pop[ab][0][o] = a0; pop[ab][1][o] = a1;
      pop[ab][0][o + 1] = c0; pop[ab][1][o + 1] = c1;
    }
  };
  auto softmax_pairs = [&](f32x16& S, int G, u32x4 (&pop)[2][3]) {
    float sum0, sum1;
    exp_pairs(S, pop, sum0, sum1);
    // any lane whose 16 values sum to 2^15 or more: some P of this block may not fit fp16 (they are >= 0)
    if (__builtin_amdgcn_ballot_w64(sum0 + sum1 >= P_TRIP) != 0ull) {
      float mx = fmaxf(fmaxf(S[0], S[1]), fmaxf(S[2], S[3]));
#pragma unroll
      for (int j = 1; j < 4; ++j) mx = fmaxf(mx, fmaxf(fmaxf(S[4 * j], S[4 * j + 1]), fmaxf(S[4 * j + 2], S[4 * j + 3])));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));                       // the two lanes that share the query
      const float delta = (mx > P_SHIFT + 1.0f) ? __builtin_ceilf(mx - P_SHIFT) : 0.f;
      const int e = -(int)delta;
#pragma unroll
      for (int r = 0; r < 16; ++r) O[G][r] = __builtin_ldexpf(O[G][r], e);
      l_run[G][0] = __builtin_ldexpf(l_run[G][0], e);
      l_run[G][1] = __builtin_ldexpf(l_run[G][1], e);
      negm2[G] -= delta;
#pragma unroll
      for (int r = 0; r < 16; ++r) S[r] -= delta;
      exp_pairs(S, pop, sum0, sum1);
    }
    l_run[G][0] += sum0;
    l_run[G][1] += sum1;
  };
  auto pv = [&](const u32x4 (&vop)[2][NVK], const u32x4 (&pop)[2][3], int G) {
#pragma unroll
    for (int ab = 0; ab < 2; ++ab) {
      if constexpr (PVH) {                               // small terms first
        O[G] = mfma32h(vop[ab][1], pop[ab][0], O[G]);    // v1 p0
        O[G] = mfma32h(vop[ab][0], pop[ab][1], O[G]);    // v0 p1
        O[G] = mfma32h(vop[ab][0], pop[ab][0], O[G]);    // v0 p0
      } else if constexpr (D == 16) {
        O[G] = mfma32(vop[ab][0], pop[ab][2], O[G]);     // v0 p2 (and v1 p2: a term beyond the six, for free)   small terms first
        O[G] = mfma32(vop[ab][1], pop[ab][0], O[G]);     // v2 p0
        O[G] = mfma32(vop[ab][0], pop[ab][1], O[G]);     // v0 p1, v1 p1
        O[G] = mfma32(vop[ab][0], pop[ab][0], O[G]);     // v0 p0, v1 p0
      } else {
#pragma unroll
        for (int term = 5; term >= 0; --term) O[G] = mfma32(vop[ab][TERM_A[term]], pop[ab][TERM_B[term]], O[G]);
      }
    }
  };
  // ---- straight order: per 32-key block and query group  QK^T -> exp / split -> P.V  (the two waves of a SIMD overlap
  // each other's phases)
  stage_load(0);
  stage_store(0);
  __syncthreads();
  stage_load(ntiles > 1 ? 1 : 0);
  {
    u32x4 K0[NKP][KS];
    load_k(0, 0, K0);
    const f32x16 s0 = qk(K0, 0, negm16), s1 = qk(K0, 1, negm16);       // C = 0 here
    float tm = fmaxf(s0[0], s1[0]);
#pragma unroll
    for (int r = 1; r < 16; ++r) tm = fmaxf(tm, fmaxf(s0[r], s1[r]));
    tm = fmaxf(tm, __shfl_xor(tm, 32, 64));
    if constexpr (PVH) {                 // per query: the maximum over its own first block
      float t0 = s0[0], t1 = s1[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) { t0 = fmaxf(t0, s0[r]); t1 = fmaxf(t1, s1[r]); }
      negm2[0] = P_SHIFT - fmaxf(t0, __shfl_xor(t0, 32, 64));
      negm2[1] = P_SHIFT - fmaxf(t1, __shfl_xor(t1, 32, 64));
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) negm16[r] = -tm;
  }
  auto block = [&](int buf, int kb) {
    u32x4 kop[NKP][KS], vop[2][NVK];
    load_k(buf, kb, kop);
    load_v(buf, kb, vop);
    if constexpr (PVH && PIPE) {
      f32x16 S2[2];
#pragma unroll
      for (int G = 0; G < 2; ++G) {
        float nm = negm2[G];
        asm volatile("" : "+v"(nm));
#pragma unroll
        for (int r = 0; r < 16; ++r) negm16[r] = nm;
        S2[G] = qk(kop, G, negm16);
      }
#pragma unroll
      for (int G = 0; G < 2; ++G) {
        u32x4 pop[2][3];
        softmax_pairs(S2[G], G, pop);
        pv(vop, pop, G);
      }
      return;
    }
#pragma unroll
    for (int G = 0; G < 2; ++G) {
      if constexpr (PVH) {
        float nm = negm2[G];
        asm volatile("" : "+v"(nm));               // a fresh splat per chain: one tuple of registers, not one per query
#pragma unroll
        for (int r = 0; r < 16; ++r) negm16[r] = nm;
      }
      f32x16 S = qk(kop, G, negm16);
      u32x4 pop[2][3];
      if constexpr (PVH) softmax_pairs(S, G, pop);
      else softmax_split(S, G, pop);
      pv(vop, pop, G);
    }
  };
  block(0, 0);
  block(0, 1);
  stage_store(1);
  __syncthreads();
  for (int t = 1; t < ntiles; ++t) {
    const int buf = t & 1;
    stage_load((t + 1 < ntiles) ? t + 1 : t);
    block(buf, 0);
    block(buf, 1);
    stage_store(buf ^ 1);
    __syncthreads();
  }


  float* obase = out + ((size_t)b * C + (size_t)head * D) * L;
#pragma unroll
  for (int G = 0; G < 2; ++G) {
    float lt = l_run[G][0] + l_run[G][1];
    lt += __shfl_xor(lt, 32, 64);
    const bool bad = !(lt < OVERFLOW_LIMIT);            // overflow (or NaN): hand this query block to the safe kernel
    const float inv = bad ? __builtin_nanf("") : 1.0f / lt;
    const int q = qblk0 + 32 * G + l31;
    if (lse2 != nullptr && h == 0)
      lse2[((size_t)b * heads + head) * L + q] = bad ? __builtin_nanf("") : __builtin_amdgcn_logf(lt) - (PVH ? negm2[G] : negm16[0]);
    // accumulator register r holds row 8 (r / 4) + 4 h + (r % 4) of O^T for query l31
    if (D == 16) {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          obase[(size_t)(8 * jj + 4 * h + i) * L + q] = (O[G][4 * jj + i] + O[G][4 * (jj + 2) + i]) * inv;
    } else if constexpr (PVH) {
      const float* vinv = reinterpret_cast<const float*>(qs + 8 * piece);      // 2^-s per channel of this head (launch_v_split_h2)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int d = 8 * (r >> 2) + 4 * h + (r & 3);
        obase[(size_t)d * L + q] = (O[G][r] * inv) * vinv[d];
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) obase[(size_t)(8 * (r >> 2) + 4 * h + (r & 3)) * L + q] = O[G][r] * inv;
    }
  }
}

}  // namespace

namespace hdiff {

// Bytes of workspace the pre-split path needs for this shape; 0 when the shape is not covered (the caller then runs the
// kernels that split in the loop, or the fp32 ones).
int64_t mha_fwd_x3p_workspace(int B, int C, int heads, int L) {
  const int D = C / heads;
  if ((D != 16 && D != 32) || L % 256 != 0 || L < 512) return 0;
  return (int64_t)B * 3 * C * L * 6 + mha_fwd_h2_tail_bytes(B, C);      // nine 2-byte pieces per element + the d_head 16 kernel's row maxima
}

// Q (pre-scaled) and K alone as bf16 triples into the workspace (for attention_h2.hip, which writes its own V pieces)
void launch_qk_split3(const float* qkv, void* ws, int B, int C, int heads, int L, float qscale, hipStream_t stream) {
  const int D = C / heads;
  dim3 sgrid(cdiv(L, 256), 2 * heads, B);          // blockIdx.y / heads = 0 (Q), 1 (K)
  if (D == 16) hipLaunchKernelGGL((qkv_split3_kernel<16>), sgrid, dim3(THREADS), 0, stream, qkv, (__bf16*)ws, C, L, qscale);
  else hipLaunchKernelGGL((qkv_split3_kernel<32>), sgrid, dim3(THREADS), 0, stream, qkv, (__bf16*)ws, C, L, qscale);
}

// Which kernel consumes the pre-split operands (both are exact to the same class; this is speed only, measured at
// L = 65 536 / 16 384, tools/x3_check.py): d_head 32 -> the 32x32x16 kernel here (220 vs 197 TFLOP/s-equivalent: M = 32 = d,
// no idle MFMA rows); d_head 16 -> the 16x16x32 kernel of attention_x3.hip reading the same workspace (at d = 16 the 32-row
// tiles carry two V pieces and spend 8 MFMA slots on 6 products).  Dev knob HDIFF_X3P: 0 = never pre-split, 16 / 32 = the
// 32x32x16 kernel for that head width only, 1 = for both.
bool launch_mha_fwd_x3p(const float* qkv, float* o, float* lse2, int B, int C, int heads, int L, float qscale, void* ws,
                        int64_t ws_bytes, hipStream_t stream) {
  const int64_t need = mha_fwd_x3p_workspace(B, C, heads, L);
  if (need == 0 || ws == nullptr || ws_bytes < need) return false;
  const int D = C / heads;
  static const char* e = getenv("HDIFF_X3P");
  const int sel = e ? atoi(e) : 32;
  if (sel == 0) return false;
  const bool wide = (sel == D || sel == 1);
  dim3 sgrid(cdiv(L, 256), 3 * heads, B), grid(L / 256, heads, B);
  if (D == 16) {
    hipLaunchKernelGGL((qkv_split3_kernel<16>), sgrid, dim3(THREADS), 0, stream, qkv, (__bf16*)ws, C, L, qscale);
    if (!wide) return launch_mha_fwd_x3(qkv, ws, o, lse2, B, C, heads, L, qscale, stream);
    hipLaunchKernelGGL((mha_flash_fwd_x3p_kernel<16>), grid, dim3(THREADS), 0, stream, (const __bf16*)ws, o, lse2, C, L, 1.0f);
  } else if (wide && mha_fwd_h2_enabled()) {
    // P.V on fp16 pairs: Q, K as bf16 triples, V as fp16 pairs with per-row powers of two (the V region of the same workspace)
    launch_qk_split_h2(qkv, ws, B, C, heads, L, qscale, stream);      // scores on fp16 pairs too (round 5)
    launch_v_split_h2(qkv, ws, B, C, heads, L, stream);
    static const char* pe = getenv("HDIFF_X3P_PIPE");       // dev knob (A/B): 0 = the straight (group, block) order (2.129 vs 2.092 ms at L = 16384, B = 2)
    if (!(pe && atoi(pe) == 0))
      hipLaunchKernelGGL((mha_flash_fwd_x3p_kernel<32, true, true>), grid, dim3(THREADS), 0, stream, (const __bf16*)ws, o, lse2, C, L, 1.0f);
    else
      hipLaunchKernelGGL((mha_flash_fwd_x3p_kernel<32, true>), grid, dim3(THREADS), 0, stream, (const __bf16*)ws, o, lse2, C, L, 1.0f);
  } else {
    hipLaunchKernelGGL((qkv_split3_kernel<32>), sgrid, dim3(THREADS), 0, stream, qkv, (__bf16*)ws, C, L, qscale);
    if (!wide) return launch_mha_fwd_x3(qkv, ws, o, lse2, B, C, heads, L, qscale, stream);
    hipLaunchKernelGGL((mha_flash_fwd_x3p_kernel<32>), grid, dim3(THREADS), 0, stream, (const __bf16*)ws, o, lse2, C, L, 1.0f);
  }
  return true;
}

}  // namespace hdiff
