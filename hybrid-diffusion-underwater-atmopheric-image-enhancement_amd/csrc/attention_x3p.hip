// Flash attention forward at d_head 32 in the split-operand mode: every operand as fp16 PAIRS on v_mfma_f32_32x32x16_f16 with fp32
// accumulation, operands split ONCE per tensor into a workspace by small streaming kernels (attention_h2.hip: launch_qk_split_h2,
// launch_v_split_h2).  Same contract as attention.hip (reference: nn.MultiheadAttention core, ModelCondition.py:189, 204-208).
// The seven L <= 16 384 launches of a 256x256 forward run here; d_head 16 runs attention_h2.hip.
//
// History (git has the code): round 3 built this kernel on bf16 triples (six products for the scores and for P V; also at
// d_head 16, where its 32-row tiles carried two V pieces); round 4 moved P V to fp16 pairs; round 5 moved the scores too and
// retired the bf16-triple forms (the in-loop-split kernel of attention_x3.hip remains for calls without a workspace).
//
//   workspace, per (sample, head), pieces of L * D 2-byte elements:  0: q0, 1: q1 2^8 [L][D];  2 .. 5: k0, k0 2^-8, k1 2^8, k1 [L][D];
//              6, 7: the two fp16 pieces of V 2^s [D][L] (s per channel row);  8: the D factors 2^-s
//   S^T = K Q^T : M = 32 keys, N = 32 queries, K = 16 of d (two k-steps).  Scores on fp16 pairs with a balance per product term
//                 (attention_h2.hip has the scheme and its error analysis): piece i of K against piece i of Q -- k0 q0,
//                 (k0 2^-8)(q1 2^8), (k1 2^8)(q0 2^-8) -- THREE terms: dropping k1 q1 leaves the error of S at 1.14x the fp32
//                 chain's (tools/h2_sim_qk_terms.py) and the output's error against float64 at 0.67-0.84x the fp32-MFMA kernel's
//                 (profiles/r05_attention_error_ratio.txt: the same as with four), for six MFMAs per 32 x 32 scores instead of
//                 twelve.  q0 2^-8 is made in registers.  The chain starts from -m and adds the large term first.
//   O^T += V^T P: P = exp2(S^T) = h0 + h1 by v_cvt_pk_f16_f32 + v_fma_mixlo / mixhi_f16 (1.5 vector instructions per value); its
//                 accumulator layout (keys 8 j + 4 h + i on registers, queries on lanes) IS the B operand of the next MFMA up to a
//                 permutation of the contraction slots, which the V operand reads follow -- no cross-lane traffic.  Three products
//                 (v1 p0, v0 p1, v0 p0) per 16 keys; M = 32 = d, every MFMA row useful.
//   reference   : fp16 ends at 65 504, so the softmax reference MOVES: it starts at (first block's maximum - 8) and whenever a
//                 lane's 16 P values of a block sum to 2^15 or more the wave recomputes that block from its S accumulator under a
//                 reference that puts the row's maximum over the block at 2^8, after scaling O and l of that query by the exact
//                 power of two.  Rows that do not need it get delta = 0 and the same bits.  Each of a lane's two queries has its
//                 OWN reference (a pair is precise only within 2^22 of fp16's top: a query that inherited the reference of a
//                 neighbour with larger scores would carry its whole row as fp16 subnormals -- measured 4e-4 instead of 8e-6);
//                 the reference enters each score chain as a freshly splat accumulator.  Plain control flow: this kernel's
//                 (query group, block) order has no hand-placed slots to protect.
//   order       : both query groups' score chains are issued before the first group's exp / split stream, so that a wave's own
//                 matrix work runs beside its vector work instead of only the other wave's (+2 % over the straight order).
#include <stdlib.h>

#include "common.h"

using namespace hdiff;

namespace {

constexpr int D = 32;
constexpr int KT = 64;
constexpr int THREADS = 256;
constexpr int NQT = 3;                            // score product terms = K pieces staged = Q pieces held
constexpr float OVERFLOW_LIMIT = 1.2379400e27f;   // 2^90, as in attention.hip
constexpr float P_SHIFT = 8.0f;                   // the reference point enters as P = 2^8 ...
constexpr float P_TRIP = 32768.0f;                // ... and moves when a lane's 16 P values of one block sum to 2^15 (attention_h2.hip)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

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

__global__ __launch_bounds__(THREADS, 2) void mha_flash_fwd_x3p_kernel(const __bf16* __restrict__ ws, float* __restrict__ out,
                                                                      float* __restrict__ lse2, int C, int L, float one) {
  constexpr int KS = D / 16;                   // k-steps of the QK^T product
  constexpr int KROWB = D * 2 + 16;            // bytes per key of one K piece in LDS (+16: conflict-free ds_read_b128)
  constexpr int KPART = KT * KROWB;
  constexpr int VROWB = KT * 2 + 8;            // bytes per d row of one V piece (+8: rows spread over the banks)
  constexpr int VPART = D * VROWB;
  constexpr int NKC = NQT * KT * D / 8;        // 16-byte chunks of a K tile (all pieces)
  constexpr int NVC = 2 * D * 8;               // 16-byte chunks of a V tile
  constexpr int NLD = (NKC + NVC) / THREADS;   // chunks per thread
  static_assert((NKC + NVC) % THREADS == 0, "staging geometry");
  constexpr int QB = 256;                      // queries per workgroup: 4 waves x 2 groups of 32
  constexpr int VBASE = NQT * KPART;
  constexpr int BUFB = (VBASE + 2 * VPART + 15) / 16 * 16;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2][BUFB];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const TileId tile = xcd_tile();
  const int head = tile.head, b = tile.b;
  const int heads = gridDim.y;
  const int qblk0 = tile.x * QB + wave * 64;
  const size_t piece = (size_t)L * D;
  const __bf16* qs = ws + ((size_t)b * heads + head) * 9 * piece;
  const __bf16* ks = qs + 2 * piece;
  const __bf16* vs = qs + 6 * piece;
  const int ntiles = L / KT;

  // Q operands (B of S^T = K Q^T): lane (query l31, half h) holds d = 16 s + 8 h .. + 7 of each piece; piece 2 = q0 2^-8 is a
  // packed fp16 multiplication of piece 0 (asm: attention_h2.hip has the reason)
  u32x4 qop[2][NQT][KS];
#pragma unroll
  for (int G = 0; G < 2; ++G) {
    const int q = qblk0 + 32 * G + l31;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int s = 0; s < KS; ++s)
        qop[G][p][s] = *reinterpret_cast<const u32x4*>(qs + p * piece + (size_t)q * D + 16 * s + 8 * h);
    const unsigned dn2 = 0x1c001c00u;          // (2^-8, 2^-8) as packed fp16
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      unsigned w4[4];
#pragma unroll
      for (int w = 0; w < 4; ++w) asm("v_pk_mul_f16 %0, %1, %2" : "=v"(w4[w]) : "v"(qop[G][0][s][w]), "v"(dn2));
      qop[G][2][s] = u32x4{w4[0], w4[1], w4[2], w4[3]};
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
  // two 8-byte stores per chunk for K and V alike (V rows are 8-byte aligned): no per-thread branch
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
  int vaddr[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) vaddr[p] = p * VPART + l31 * VROWB + 8 * h;

  f32x16 O[2], negm16;
  f32x2 l_run[2];
  float negm2[2] = {0.f, 0.f};          // -m of the lane's two queries
#pragma unroll
  for (int G = 0; G < 2; ++G) {
    l_run[G] = f32x2{0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 16; ++r) O[G][r] = 0.f;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) negm16[r] = 0.f;

  auto load_k = [&](int buf, int kb, u32x4 (&kop)[NQT][KS]) {
    const unsigned char* kbuf = smem[buf] + kb * 32 * KROWB + kaddr;
#pragma unroll
    for (int p = 0; p < NQT; ++p)
#pragma unroll
      for (int s = 0; s < KS; ++s) kop[p][s] = *reinterpret_cast<const u32x4*>(kbuf + p * KPART + 32 * s);
  };
  // V operands of the two 16-key halves of a 32-key block: contraction slot 8 h + 4 jj + i  <->  key 16 ab + 8 jj + 4 h + i
  auto load_v = [&](int buf, int kb, u32x4 (&vop)[2][2]) {
    const unsigned char* vbuf = smem[buf] + VBASE + kb * 64;               // 32 keys = 64 bytes along a V row
#pragma unroll
    for (int ab = 0; ab < 2; ++ab)
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const unsigned char* src = vbuf + vaddr[p] + 32 * ab;
        const u32x2 lo = *reinterpret_cast<const u32x2*>(src);
        const u32x2 hi2 = *reinterpret_cast<const u32x2*>(src + 16);
        vop[ab][p] = u32x4{lo[0], lo[1], hi2[0], hi2[1]};
      }
  };
  auto qk = [&](const u32x4 (&kop)[NQT][KS], int G, f32x16 c) {
    f32x16 S = c;                                  // the chain starts from -m: the accumulator holds s - m
#pragma unroll
    for (int term = 0; term < NQT; ++term)         // piece i of K against piece i of Q, large term first
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        u32x4 qb = qop[G][term][s];
        if ((HDIFF_MUTANT & 2) && term == 1)         // (mutation test: the low five bits of q1 2^8 dropped: 2^-17 of q)
#pragma unroll
          for (int w = 0; w < 4; ++w) qb[w] &= 0xffe0ffe0u;
        S = mfma32h(kop[term][s], qb, S);
      }
    return S;
  };
  // P = exp2(S) of one (block, query group) as fp16 pairs + the lane's row sums; the reference moves when it must
  auto exp_pairs = [&](const f32x16& S, u32x4 (&pop)[2][2], float& sum0, float& sum1) {
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
  auto softmax_pairs = [&](f32x16& S, int G, u32x4 (&pop)[2][2]) {
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
  auto pv = [&](const u32x4 (&vop)[2][2], const u32x4 (&pop)[2][2], int G) {
#pragma unroll
    for (int ab = 0; ab < 2; ++ab) {                   // small terms first
      O[G] = mfma32h(vop[ab][1], pop[ab][0], O[G]);    // v1 p0
      O[G] = mfma32h(vop[ab][0], pop[ab][1], O[G]);    // v0 p1
      O[G] = mfma32h(vop[ab][0], pop[ab][0], O[G]);    // v0 p0
    }
  };

  stage_load(0);
  stage_store(0);
  __syncthreads();
  stage_load(ntiles > 1 ? 1 : 0);
  {
    // the reference points: per query, the maximum over its own first block (scores with C = 0)
    u32x4 K0[NQT][KS];
    load_k(0, 0, K0);
    const f32x16 s0 = qk(K0, 0, negm16), s1 = qk(K0, 1, negm16);
    float t0 = s0[0], t1 = s1[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) { t0 = fmaxf(t0, s0[r]); t1 = fmaxf(t1, s1[r]); }
    negm2[0] = P_SHIFT - fmaxf(t0, __shfl_xor(t0, 32, 64));
    negm2[1] = P_SHIFT - fmaxf(t1, __shfl_xor(t1, 32, 64));
  }
  auto block = [&](int buf, int kb) {
    u32x4 kop[NQT][KS], vop[2][2];
    load_k(buf, kb, kop);
    load_v(buf, kb, vop);
    f32x16 S2[2];
#pragma unroll
    for (int G = 0; G < 2; ++G) {
      float nm = negm2[G];
      asm volatile("" : "+v"(nm));               // a fresh splat per chain: one tuple of registers, not one per query
#pragma unroll
      for (int r = 0; r < 16; ++r) negm16[r] = nm;
      S2[G] = qk(kop, G, negm16);
    }
#pragma unroll
    for (int G = 0; G < 2; ++G) {
      u32x4 pop[2][2];
      softmax_pairs(S2[G], G, pop);
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
  const float* vinv = reinterpret_cast<const float*>(qs + 8 * piece);      // 2^-s per channel of this head (launch_v_split_h2)
#pragma unroll
  for (int G = 0; G < 2; ++G) {
    float lt = l_run[G][0] + l_run[G][1];
    lt += __shfl_xor(lt, 32, 64);
    const bool bad = !(lt < OVERFLOW_LIMIT);            // NaN / inf inputs: hand this query block to the fp32 kernel's check pass
    const float inv = bad ? __builtin_nanf("") : 1.0f / lt;
    const int q = qblk0 + 32 * G + l31;
    if (lse2 != nullptr && h == 0)
      lse2[((size_t)b * heads + head) * L + q] = bad ? __builtin_nanf("") : __builtin_amdgcn_logf(lt) - negm2[G];
    // accumulator register r holds row 8 (r / 4) + 4 h + (r % 4) of O^T for query l31
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int d = 8 * (r >> 2) + 4 * h + (r & 3);
      obase[(size_t)d * L + q] = (O[G][r] * inv) * vinv[d];
    }
  }
}

}  // namespace

namespace hdiff {

// Bytes of workspace the pre-split kernels need for this shape (nine 2-byte pieces per element + the Q / K row maxima behind the
// pairs); 0 when the shape is not covered (the caller then runs the kernel that splits in its loop, or the fp32 ones).
int64_t mha_fwd_x3p_workspace(int B, int C, int heads, int L) {
  const int Dh = C / heads;
  if ((Dh != 16 && Dh != 32) || L % 256 != 0 || L < 512) return 0;
  return (int64_t)B * 3 * C * L * 6 + mha_fwd_h2_tail_bytes(B, C);
}

// The d_head 32 forward on the pre-split fp16 operands.  Returns false when the shape is not covered or the workspace is missing.
bool launch_mha_fwd_x3p(const float* qkv, float* o, float* lse2, int B, int C, int heads, int L, float qscale, void* ws,
                        int64_t ws_bytes, hipStream_t stream) {
  const int64_t need = mha_fwd_x3p_workspace(B, C, heads, L);
  if (need == 0 || ws == nullptr || ws_bytes < need || C / heads != 32) return false;
  launch_qk_split_h2(qkv, ws, B, C, heads, L, qscale, stream);
  launch_v_split_h2(qkv, ws, B, C, heads, L, stream);
  hipLaunchKernelGGL(mha_flash_fwd_x3p_kernel, dim3(L / 256, heads, B), dim3(THREADS), 0, stream, (const __bf16*)ws, o, lse2, C, L, 1.0f);
  return true;
}

}  // namespace hdiff
