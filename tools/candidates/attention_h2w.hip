// Flash attention forward at d_head 16, round 5: the fp16-pair formulation of attention_h2.hip on v_mfma_f32_32x32x16 tiles.
// Same contract as attention.hip (reference: nn.MultiheadAttention core, ModelCondition.py:189, 204-208).
//
// Why (profiles/r04_pmc_summary.txt, tools/h2w_stage_probe.hip, MI355X_MICROARCH.md 'vector-instruction ISSUE cost'): the
// 16x16x32 kernel is bound by vector ISSUE, and every MFMA holds the SIMD's vector issue for 8 cycles whatever its shape --
// its 18 MFMAs per 1024 scores hold 144 cycles beside 56 vector instructions.  Here the same 1024 scores (32 queries x 32
// keys of one wave, 16 per lane) take 10 MFMAs:
//   S^T = K Q^T : M = 32 keys, N = 32 queries, K = 16 = d_head -- ONE bf16-triple term per MFMA, the six terms with
//                 i + j <= 2 in one chain that starts from -m (fp32-class scores, as before);
//   O^T += V^T P: the two fp16 pieces of V are STACKED on the 32 M rows ([v0; v1], 16 channels each), so p0 yields
//                 v0 p0 and v1 p0 in one MFMA and p1 yields v0 p1 (and v1 p1, a term beyond the three, for free): 2 MFMAs
//                 per 16 keys.  The two halves are added once, in the epilogue.  P = exp2(S^T)'s accumulator layout (keys
//                 8 j + 4 h + i on registers, queries on lanes) IS the B operand up to a permutation of the contraction
//                 slots, which the V image follows -- no cross-lane traffic.
// The vector work is unchanged (16 v_exp, 8 v_cvt_pk_f16_f32, 16 v_fma_mix, 16 adds per stage) but it is PLACED: the
// exps are spread two per MFMA gap and every gap carries about 28 cycles of vector issue (tools/h2w_sched.py; the probe
// measures 462 cycles per stage against 480 for the natural order and 522 for the 16x16x32 stage).
//
// Operands arrive in MFMA-operand order: a streaming pass (h2w_qk_split_kernel / h2w_v_split_kernel) writes, per (sample,
// head) and per block of 32 keys, 5 KiB = three K pieces [half h][key][8 d] (bf16) + two 16-key steps of the stacked V
// pieces [half h][row m][8 slots] (fp16, slot 4 jj + i of half h = key 16 step + 8 jj + 4 h + i), so that EVERY operand of
// the loop is "16 bytes at lane * 16" of a 1 KiB piece: the tile is copied verbatim global -> LDS by LDS-DMA
// (global_load_lds_dwordx4: no staging registers, no ds_write, no address arithmetic) and read back with conflict-free
// ds_read_b128.  Three LDS buffers of 128 keys: the DMA of tile t + 2 is issued right behind the ONE barrier of tile t and
// has a whole tile to land.
//
// Softmax reference: as in attention_h2.hip -- fp16 ends at 65 504, so m starts at (first block's maximum - 8) per query
// and MOVES when a lane's 16 P values of a stage sum to 2^15: the wave recomputes that stage from its S accumulator under the
// new reference after scaling O, l by the exact power of two (out-of-line asm pieces entered by a scalar branch; rows that
// do not move get the same bits).  The reference is one value per lane here (a lane owns one query of each group) and enters
// each score chain as a persistent splat accumulator.
#include <stdlib.h>

#include <type_traits>

#include "attention_h2w_sched.h"
#include "common.h"

using namespace hdiff;

namespace {

#ifndef H2W_DIAG
#define H2W_DIAG 0     // diagnostic build (tools/h2w_clock.py): every workgroup stamps s_memtime / s_memrealtime around its tile loop into
#endif                 // a part of the workspace nothing else reads (MI355X_MICROARCH.md, DVFS item 6); never set in the product
#ifndef H2W_ABL
#define H2W_ABL 0      // timing ablations (wrong results by construction): 1 no barrier, 2 no rolling K / V operand reloads,
#endif                 // 4 no LDS-DMA, 8 no reference check
constexpr int THREADS = 256;
constexpr int BLKB = 5120;                        // bytes of one 32-key block: K 3 x 1 KiB, V 2 x 1 KiB
constexpr int VOFF = 3072;                        // V steps inside a block
constexpr float OVERFLOW_LIMIT = 1.2379400e27f;   // 2^90: only NaN / inf inputs get here
constexpr float P_SHIFT = 8.0f;                   // the reference point enters as P = 2^8
constexpr float P_TRIP = 32768.0f;                // per-lane sum of one stage's 16 P values that moves the reference

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) unsigned char lds_byte;

__device__ __forceinline__ f32x16 mfma32(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32h(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// First MFMA of a score chain: D = A B + C with C (the splat -m of the query group) in a tuple of its own.  Written as asm
// because the compiler only has the tied form (C = D) for 16-register accumulators in VGPRs and would copy the 16 registers of
// -m into S in front of every chain.  Nothing reads D before the chain's next MFMA (same opcode, back to back: no wait states).
__device__ __forceinline__ f32x16 mfma32_start(u32x4 a, u32x4 b, const f32x16& c) {
  f32x16 d;
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

// bf16 triples by truncation (x = x0 + x1 + x2 exactly; attention_x3.hip) and fp16 pairs by rounding (attention_h2.hip)
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
__device__ __forceinline__ void split2(float a, float b, float one, unsigned& h0, unsigned& h1) {
  const f16x2 p = {(_Float16)a, (_Float16)b};                 // v_cvt_pk_f16_f32: round to nearest even
  unsigned u = __builtin_bit_cast(unsigned, p);
  asm("" : "+v"(u));
  const f16x2 q = __builtin_bit_cast(f16x2, u);
  const f16x2 r = {(_Float16)__builtin_fmaf(a, one, -(float)q[0]), (_Float16)__builtin_fmaf(b, one, -(float)q[1])};
  h0 = u;
  h1 = __builtin_bit_cast(unsigned, r);
}

// bytes of one (sample, head) pair in the workspace: Q pieces [3][L][16] bf16, the key blocks, 16 factors 2^-s
__host__ __device__ constexpr size_t pair_bytes(int L) { return (size_t)288 * L; }      // = the stride of attention_x3p.hip's layout
__host__ __device__ constexpr size_t kv_offset(int L) { return (size_t)96 * L; }
__host__ __device__ constexpr size_t vinv_offset(int L) { return (size_t)256 * L; }

// ---------------------------------------------------------------------------------------------------------------------
// fp32 qkv [B][3C][L] -> Q as bf16 triples [3][L][16] (pre-scaled into the exp2 domain) and K as bf16 triples in block order.
// grid (L / 256, 2 * heads, B); thread = one position, all 16 channels (reads coalesced over the threads; writes 16-byte
// chunks that are contiguous over the threads of a wave).
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(THREADS) void h2w_qk_split_kernel(const float* __restrict__ qkv, unsigned char* __restrict__ ws, int C,
                                                              int L, float qscale) {
  constexpr int D = 16;
  const int heads = C / D;
  const int which = blockIdx.y / heads, head = blockIdx.y - which * heads, b = blockIdx.z;
  const int l = blockIdx.x * THREADS + threadIdx.x;
  if (l >= L) return;
  const float* src = qkv + ((size_t)b * 3 * C + (size_t)which * C + (size_t)head * D) * L;
  unsigned char* pair = ws + ((size_t)b * heads + head) * pair_bytes(L);
  const float sc = which == 0 ? qscale : 1.0f;
  unsigned h[3][D / 2];
#pragma unroll
  for (int j = 0; j < D / 2; ++j) {
    const float a = src[(size_t)(2 * j) * L + l] * sc, c = src[(size_t)(2 * j + 1) * L + l] * sc;
    split3(a, c, h[0][j], h[1][j], h[2][j]);
  }
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      unsigned char* dst = which == 0 ? pair + ((size_t)p * L + l) * 32 + 16 * hh
                                      : pair + kv_offset(L) + (size_t)(l >> 5) * BLKB + p * 1024 + hh * 512 + (l & 31) * 16;
      *reinterpret_cast<u32x4*>(dst) = u32x4{h[p][4 * hh], h[p][4 * hh + 1], h[p][4 * hh + 2], h[p][4 * hh + 3]};
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// V of qkv -> two fp16 pieces of V * 2^s (s per channel row, max |V 2^s| in [2^14, 2^15)) in block order, the pieces stacked
// as rows m = 16 piece + channel of the P.V A operand; the 16 factors 2^-s behind the blocks.
// One workgroup per (sample, channel) row: a maximum pass, then the split pass (the row comes back from L2).
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(THREADS) void h2w_v_split_kernel(const float* __restrict__ qkv, unsigned char* __restrict__ ws, int C,
                                                             int L, float one) {
  constexpr int D = 16;
  const int heads = C / D;
  const int row = blockIdx.x, head = row / D, d = row - head * D, b = blockIdx.y;
  const int tid = threadIdx.x;
  const float* src = qkv + ((size_t)b * 3 * C + 2 * (size_t)C + row) * L;
  unsigned char* pair = ws + ((size_t)b * heads + head) * pair_bytes(L);
  __shared__ float red[THREADS / 64];

  float amax = 0.f;
  for (int i = tid; i < L / 4; i += THREADS) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + 4 * (size_t)i);
    amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
  if ((tid & 63) == 0) red[tid >> 6] = amax;
  __syncthreads();
  amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  // exponent clamped so that both 2^s and 2^-s are normal numbers (attention_h2.hip: inf / NaN elements stay inf / NaN in fp16)
  int e = (int)((__builtin_bit_cast(unsigned, amax) >> 23) & 0xffu) - 127;
  e = e < -100 ? -100 : (e > 127 ? 127 : e);
  const float scale = __builtin_bit_cast(float, (unsigned)(14 - e + 127) << 23);
  if (tid == 0) reinterpret_cast<float*>(pair + vinv_offset(L))[d] = __builtin_bit_cast(float, (unsigned)(e - 14 + 127) << 23);
  unsigned char* kv = pair + kv_offset(L) + VOFF + d * 16;
  for (int i = tid; i < L / 4; i += THREADS) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + 4 * (size_t)i);
    unsigned a0, a1, c0, c1;
    split2(v[0] * scale, v[1] * scale, one, a0, a1);
    split2(v[2] * scale, v[3] * scale, one, c0, c1);
    // keys 4 i .. 4 i + 3: block i / 8, 16-key step (i / 4) & 1, quad kq = i & 3 -> half h = kq & 1, slots 4 (kq >> 1) ..
    unsigned char* dst = kv + (size_t)(i >> 3) * BLKB + ((i >> 2) & 1) * 1024 + (i & 1) * 512 + ((i >> 1) & 1) * 8;
    *reinterpret_cast<u32x2*>(dst) = u32x2{a0, c0};
    *reinterpret_cast<u32x2*>(dst + 256) = u32x2{a1, c1};            // piece 1: rows m = 16 + d
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The vector work of one stage (16 scores per lane) as 56 numbered steps, op = 14 kt + r for the score quad kt:
//   r 0-3 v_exp_f32 | 4, 5 v_cvt_pk_f16_f32 | 6, 7 p0 + p2, p1 + p3 | 8, 9 v_fma_mixlo_f16 | 10, 11 v_fma_mixhi_f16 | 12, 13 sums
// placed into the ten MFMA gaps by tools/h2w_sched.py (attention_h2w_sched.h: v_exp 8 cycles, the others 4; the exps spread over the
// gaps; two instructions between a transcendental or a v_fma_mixlo and its consumer, one between any other pair: the hazard
// recogniser then has nothing to pad)
// ---------------------------------------------------------------------------------------------------------------------
#ifndef H2W_SCHED
#define H2W_SCHED 0
#endif
__device__ constexpr H2wSched SCHED = H2W_SCHEDS[H2W_SCHED];

// MFMA slots of a stage: Q K^T of the NEXT stage (q: 6 terms) and P V of the PREVIOUS one (p: 2 k-steps x 2 pieces).  The
// score chain ends at slot 7 so that its accumulator has settled when the next stage's first exp reads it.
__device__ constexpr int SLOT_IS_PV[10] = {0, 0, 1, 0, 0, 1, 0, 0, 1, 1};
// score terms (piece of K, piece of Q), LARGE ones first: the chain starts from -m, and -m + k0 q0 cancels to the magnitude of
// s - m before the small terms arrive (small-first rounds each of them at the magnitude of m: measured 1.45x the fp32 kernel's
// error on peaked rows).  K piece 0 is free after term 3, piece 1 after term 4 -- and the next chain opens with piece 0.
__device__ constexpr int TERM_K[6] = {0, 0, 1, 0, 1, 2};
__device__ constexpr int TERM_Q[6] = {0, 1, 0, 2, 1, 0};

// ---------------------------------------------------------------------------------------------------------------------
// Moving the softmax reference (attention_h2.hip has the why and the form): asm pieces that branch on a saved scalar
// condition to code kept out of line and come back; the results of the rare path land in the registers of the common one.
// ---------------------------------------------------------------------------------------------------------------------
#define H2W_RARE_BEGIN "s_cmp_lg_u64 %[cond], 0\n\ts_cbranch_scc1 .Lh2wr_%=\n.Lh2wb_%=:\n\t.subsection 1\n.Lh2wr_%=:\n\t"
#define H2W_RARE_END "s_branch .Lh2wb_%=\n\t.subsection 0"

// the query's maximum of S (= s - m) over this stage's 32 keys and the move: delta = mx > 9 ? ceil(mx - 8) : 0
__device__ __forceinline__ float h2w_rare_delta(unsigned long long cond, const f32x16& S, int bp32) {
  float delta, t, u;
  asm volatile(H2W_RARE_BEGIN
               "v_max3_f32 %[t], %[s0], %[s1], %[s2]\n\t"
               "v_max3_f32 %[u], %[s3], %[s4], %[s5]\n\t"
               "v_max3_f32 %[t], %[t], %[s6], %[s7]\n\t"
               "v_max3_f32 %[u], %[u], %[s8], %[s9]\n\t"
               "v_max3_f32 %[t], %[t], %[s10], %[s11]\n\t"
               "v_max3_f32 %[u], %[u], %[s12], %[s13]\n\t"
               "v_max3_f32 %[t], %[t], %[s14], %[s15]\n\t"
               "v_max_f32 %[t], %[t], %[u]\n\t"
               "s_nop 1\n\t"
               "ds_bpermute_b32 %[u], %[bp32], %[t]\n\t"
               "s_waitcnt lgkmcnt(0)\n\t"
               "v_max_f32 %[t], %[t], %[u]\n\t"              // over the two lanes that share the query
               "v_subrev_f32 %[u], 8.0, %[t]\n\t"
               "v_ceil_f32 %[u], %[u]\n\t"
               "v_cmp_lt_f32 vcc, 0x41100000, %[t]\n\t"      // 9.0 < mx
               "v_cndmask_b32 %[d], 0, %[u], vcc\n\t"
               H2W_RARE_END
               : [d] "=&v"(delta), [t] "=&v"(t), [u] "=&v"(u)
               : [cond] "s"(cond), [bp32] "v"(bp32), [s0] "v"(S[0]), [s1] "v"(S[1]), [s2] "v"(S[2]), [s3] "v"(S[3]), [s4] "v"(S[4]),
                 [s5] "v"(S[5]), [s6] "v"(S[6]), [s7] "v"(S[7]), [s8] "v"(S[8]), [s9] "v"(S[9]), [s10] "v"(S[10]), [s11] "v"(S[11]),
                 [s12] "v"(S[12]), [s13] "v"(S[13]), [s14] "v"(S[14]), [s15] "v"(S[15])
               : "vcc", "scc");
  return delta;      // defined only on the rare path; only the rare path reads it
}

// O[0..15] *= 2^-delta (O of the query group)
__device__ __forceinline__ void h2w_rare_scale_o(unsigned long long cond, float delta, f32x16& X) {
  float x0 = X[0], x1 = X[1], x2 = X[2], x3 = X[3], x4 = X[4], x5 = X[5], x6 = X[6], x7 = X[7], x8 = X[8], x9 = X[9], x10 = X[10],
        x11 = X[11], x12 = X[12], x13 = X[13], x14 = X[14], x15 = X[15], w;
#define H2W_OP(n) "v_ldexp_f32 %[x" #n "], %[x" #n "], %[w]\n\t"
  asm volatile(H2W_RARE_BEGIN "v_cvt_i32_f32 %[w], %[d]\n\tv_sub_u32 %[w], 0, %[w]\n\t" H2W_OP(0) H2W_OP(1) H2W_OP(2) H2W_OP(3) H2W_OP(4)
                   H2W_OP(5) H2W_OP(6) H2W_OP(7) H2W_OP(8) H2W_OP(9) H2W_OP(10) H2W_OP(11) H2W_OP(12) H2W_OP(13) H2W_OP(14) H2W_OP(15)
                       H2W_RARE_END
               : [x0] "+v"(x0), [x1] "+v"(x1), [x2] "+v"(x2), [x3] "+v"(x3), [x4] "+v"(x4), [x5] "+v"(x5), [x6] "+v"(x6), [x7] "+v"(x7),
                 [x8] "+v"(x8), [x9] "+v"(x9), [x10] "+v"(x10), [x11] "+v"(x11), [x12] "+v"(x12), [x13] "+v"(x13), [x14] "+v"(x14),
                 [x15] "+v"(x15), [w] "=&v"(w)
               : [cond] "s"(cond), [d] "v"(delta)
               : "scc");
#undef H2W_OP
  X = f32x16{x0, x1, x2, x3, x4, x5, x6, x7, x8, x9, x10, x11, x12, x13, x14, x15};
}

// The splat -m of the query group -= delta, l *= 2^-delta, the lane's row sums restart from zero.  The 16 registers of -m are
// the C operand of an asm MFMA (mfma32_start) and must stay ONE tuple, so they are updated as one: -m += A B on the fp32 MFMA
// with A = (1 in contraction slot 0, i.e. in the lanes of half 0) and B = -delta of the lane's query -- exact, any delta.
__device__ __forceinline__ void h2w_rare_ref(unsigned long long cond, float delta, float a_slot0, f32x16& negm, float& l, float& sum0,
                                             float& sum1) {
  float w, t;
  asm volatile(H2W_RARE_BEGIN
               "v_xor_b32 %[t], 0x80000000, %[d]\n\t"
               "v_cvt_i32_f32 %[w], %[d]\n\t"
               "v_sub_u32 %[w], 0, %[w]\n\t"
               "v_mfma_f32_32x32x2_f32 %[n], %[a], %[t], %[n]\n\t"
               "v_ldexp_f32 %[l], %[l], %[w]\n\t"
               "v_mov_b32 %[q0], 0\n\t"
               "v_mov_b32 %[q1], 0\n\t"
               H2W_RARE_END
               : [n] "+v"(negm), [l] "+v"(l), [q0] "+v"(sum0), [q1] "+v"(sum1), [w] "=&v"(w), [t] "=&v"(t)
               : [cond] "s"(cond), [d] "v"(delta), [a] "v"(a_slot0)
               : "scc");
}

// eight scores (two quads) again: P = exp2(S - delta), its fp16 pieces and the row sums -- the common path's instructions
// in the common path's order per value, so a row with delta = 0 gets its bits back
__device__ __forceinline__ void h2w_rare_exp_split(unsigned long long cond, float delta, float one, float s0, float s1, float s2, float s3,
                                                   float s4, float s5, float s6, float s7, unsigned& a0, unsigned& a1, unsigned& r0,
                                                   unsigned& r1, unsigned& a2, unsigned& a3, unsigned& r2, unsigned& r3, float& sum0,
                                                   float& sum1) {
  float p0, p1, p2, p3, t0, t1;
#define H2W_QUAD(S0, S1, S2, S3, A0, A1, R0, R1)                                                   \
  "v_sub_f32 %[p0], %[" #S0 "], %[d]\n\t"                                                          \
  "v_sub_f32 %[p1], %[" #S1 "], %[d]\n\t"                                                          \
  "v_sub_f32 %[p2], %[" #S2 "], %[d]\n\t"                                                          \
  "v_sub_f32 %[p3], %[" #S3 "], %[d]\n\t"                                                          \
  "v_exp_f32 %[p0], %[p0]\n\t"                                                                    \
  "v_exp_f32 %[p1], %[p1]\n\t"                                                                    \
  "v_exp_f32 %[p2], %[p2]\n\t"                                                                    \
  "v_exp_f32 %[p3], %[p3]\n\t"                                                                    \
  "s_nop 1\n\t"                                                                                   \
  "v_cvt_pk_f16_f32 %[" #A0 "], %[p0], %[p1]\n\t"                                                 \
  "v_cvt_pk_f16_f32 %[" #A1 "], %[p2], %[p3]\n\t"                                                 \
  "v_add_f32 %[t0], %[p0], %[p2]\n\t"                                                             \
  "v_add_f32 %[t1], %[p1], %[p3]\n\t"                                                             \
  "v_fma_mixlo_f16 %[" #R0 "], %[p0], %[one], -%[" #A0 "] op_sel_hi:[0,0,1]\n\t"                   \
  "v_fma_mixlo_f16 %[" #R1 "], %[p2], %[one], -%[" #A1 "] op_sel_hi:[0,0,1]\n\t"                   \
  "v_add_f32 %[q0], %[q0], %[t0]\n\t"                                                             \
  "v_add_f32 %[q1], %[q1], %[t1]\n\t"                                                             \
  "v_fma_mixhi_f16 %[" #R0 "], %[p1], %[one], -%[" #A0 "] op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"    \
  "v_fma_mixhi_f16 %[" #R1 "], %[p3], %[one], -%[" #A1 "] op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"    \
  "s_nop 1\n\t"
  asm volatile(H2W_RARE_BEGIN H2W_QUAD(s0, s1, s2, s3, a0, a1, r0, r1) H2W_QUAD(s4, s5, s6, s7, a2, a3, r2, r3) H2W_RARE_END
               : [a0] "+v"(a0), [a1] "+v"(a1), [r0] "+v"(r0), [r1] "+v"(r1), [a2] "+v"(a2), [a3] "+v"(a3), [r2] "+v"(r2), [r3] "+v"(r3),
                 [q0] "+v"(sum0), [q1] "+v"(sum1), [p0] "=&v"(p0), [p1] "=&v"(p1), [p2] "=&v"(p2), [p3] "=&v"(p3), [t0] "=&v"(t0),
                 [t1] "=&v"(t1)
               : [cond] "s"(cond), [d] "v"(delta), [one] "s"(one), [s0] "v"(s0), [s1] "v"(s1), [s2] "v"(s2), [s3] "v"(s3), [s4] "v"(s4),
                 [s5] "v"(s5), [s6] "v"(s6), [s7] "v"(s7)
               : "scc");
#undef H2W_QUAD
}

// one 1 KiB piece global -> LDS: lane i's 16 bytes at src + voff land at lds_dst + 16 i.  M0 is written in the statement that
// uses it and restored (cdna_hip_programming.md, 'What hipcc does not do'); the compiler does not count this load: the
// kernel waits for it with its own s_waitcnt vmcnt(0) in front of the barrier that publishes the tile.
__device__ __forceinline__ void dma_piece(const unsigned char* src, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(src), "s"(lds_dst)
               : "memory");
}

// ---------------------------------------------------------------------------------------------------------------------
template <int TB>
__global__ __launch_bounds__(THREADS, 2) void mha_flash_fwd_h2w_kernel(const unsigned char* __restrict__ ws, float* __restrict__ out,
                                                                       float* __restrict__ lse2, int C, int L, float one) {
  constexpr int D = 16;
  constexpr int TILEB = TB * BLKB;             // bytes of one tile (TB blocks of 32 keys)
  constexpr int NPIECE = TILEB / 1024;         // 1 KiB DMA pieces per tile
  constexpr int NS = 2 * TB;                   // stages per tile
  __shared__ __attribute__((aligned(1024))) unsigned char smem[3 * TILEB];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const TileId tile = xcd_tile();
  const int head = tile.head, b = tile.b, heads = gridDim.y;
  const int qblk0 = tile.x * 256 + wave * 64;
  const int ntiles = L / (32 * TB);
  const unsigned char* pair = ws + ((size_t)b * heads + head) * pair_bytes(L);
  const unsigned char* kv = pair + kv_offset(L);

  // Q operands (B of S^T = K Q^T): lane (query l31, half h) holds d = 8 h .. + 7 of each piece
  u32x4 qop[2][3];
#pragma unroll
  for (int G = 0; G < 2; ++G)
#pragma unroll
    for (int p = 0; p < 3; ++p)
      qop[G][p] = *reinterpret_cast<const u32x4*>(pair + ((size_t)p * L + (qblk0 + 32 * G + l31)) * 32 + 16 * h);

  const unsigned lds0 = (unsigned)(size_t)(lds_byte*)smem;
  const unsigned lane16 = lane * 16;
  // tile t -> buffer t % 3; this wave copies pieces wave, wave + 4, ...
  auto dma_tile = [&](int t, int bufi) __attribute__((always_inline)) {
    if (H2W_ABL & 4) return;
    const unsigned char* src = kv + (size_t)t * TILEB;
#pragma unroll
    for (int n = 0; n < (NPIECE + 3) / 4; ++n) {
      const int piece = wave + 4 * n;
      if (NPIECE % 4 == 0 || piece < NPIECE) dma_piece(src + piece * 1024, lane16, lds0 + bufi * TILEB + piece * 1024);
    }
  };

  f32x16 O[2], negm[2], S[2];
  float l_run[2] = {0.f, 0.f};
  u32x4 kop[3], vop[2];
  u32x4 pop[2][2][2];           // [query group][piece][k-step]
  const int bp32 = (lane ^ 32) * 4;
  const float a_slot0 = h ? 0.f : 1.f;      // A of the reference update (h2w_rare_ref)
#pragma unroll
  for (int G = 0; G < 2; ++G)
#pragma unroll
    for (int r = 0; r < 16; ++r) { O[G][r] = 0.f; negm[G][r] = 0.f; }

  // operand reads: 16 bytes at lane * 16 of a 1 KiB piece
  auto lds_read = [&](int base, int off) __attribute__((always_inline)) { return *reinterpret_cast<const u32x4*>(smem + base + off + lane16); };

  // MFMA slot i of the stage (kb, G): the score chain of the next stage into S[G ^ 1], P V of the previous stage from
  // pop[G ^ 1] into O[G ^ 1].  G = 0: the next stage is (kb, 1), same keys -- each K piece is reloaded for block kb + 1 behind
  // its last term -- and the previous one is (kb - 1, 1): V(kb) follows V(kb - 1) through vop behind each k-step's second MFMA.
  auto mfma_slot = [&](int i, int G, bool pend, int knext_base, int knext_off, int vcur_base, int vcur_off) __attribute__((always_inline)) {
    int nq = 0, np = 0;
    for (int k = 0; k < i; ++k) (SLOT_IS_PV[k] ? np : nq)++;
    if (SLOT_IS_PV[i]) {
      if (!pend) return;
      const int ks = np >> 1, piece = 1 - (np & 1);                // small term first: [v0; v1] p1, then [v0; v1] p0
      u32x4 pb = pop[G ^ 1][piece][ks];
      if ((HDIFF_MUTANT & 8) && piece == 1)                        // (mutation test: the low five bits of every second piece of P dropped)
#pragma unroll
        for (int w = 0; w < 4; ++w) pb[w] &= 0xffe0ffe0u;
      O[G ^ 1] = mfma32h(vop[ks], pb, O[G ^ 1]);
      if (G == 0 && piece == 0 && !(H2W_ABL & 2)) vop[ks] = lds_read(vcur_base, vcur_off + VOFF + ks * 1024);
    } else {
      const int kp = TERM_K[nq], qp = TERM_Q[nq];
      if (!((HDIFF_MUTANT & 4) && kp == 0 && qp == 2))             // (mutation test: the k0 q2 term dropped)
        S[G ^ 1] = nq == 0 ? mfma32_start(kop[kp], qop[G ^ 1][qp], negm[G ^ 1]) : mfma32(kop[kp], qop[G ^ 1][qp], S[G ^ 1]);      // the chain starts from -m
      bool last = true;
      for (int k = nq + 1; k < 6; ++k) last = last && TERM_K[k] != kp;
      if (G == 0 && last && !(H2W_ABL & 2)) kop[kp] = lds_read(knext_base, knext_off + kp * 1024);
    }
  };

  // One stage = the vector work of (block kb, query group G): S[G] -> P pieces in pop[G], hand-interleaved with the MFMAs of
  // the neighbouring stages, fenced slot by slot.
  auto stage_fn = [&](auto g_tag, bool pend, int knext_base, int knext_off, int vcur_base, int vcur_off) __attribute__((always_inline)) {
    constexpr int G = decltype(g_tag)::value;
    float pe[16], ad[8], sum0 = 0.f, sum1 = 0.f;
    unsigned u[8], r2[8];
    _Float16 rl[8];
    // every step ends in an (empty) volatile asm on its result: volatile statements keep their order, which pins the step
    // into its gap (left alone, instruction selection emits these chain-less operations in an order of its own)
    auto vstep = [&](int op) __attribute__((always_inline)) {
      const int kt = op / 14, r = op - kt * 14;
      if (r < 4) {
        pe[4 * kt + r] = __builtin_amdgcn_exp2f(S[G][4 * kt + r]);
        asm volatile("" : "+v"(pe[4 * kt + r]));
      } else if (r < 6) {
        const f16x2 p = {(_Float16)pe[4 * kt + 2 * (r - 4)], (_Float16)pe[4 * kt + 2 * (r - 4) + 1]};
        unsigned w = __builtin_bit_cast(unsigned, p);
        asm volatile("" : "+v"(w));
        u[2 * kt + r - 4] = w;
      } else if (r < 8) {
        ad[2 * kt + r - 6] = pe[4 * kt + r - 6] + pe[4 * kt + r - 4];
        asm volatile("" : "+v"(ad[2 * kt + r - 6]));
      } else if (r < 10) {
        rl[2 * kt + r - 8] = (_Float16)__builtin_fmaf(pe[4 * kt + 2 * (r - 8)], one, -(float)__builtin_bit_cast(f16x2, u[2 * kt + r - 8])[0]);
        asm volatile("" : "+v"(rl[2 * kt + r - 8]));
      } else if (r < 12) {
        const f16x2 pr = {rl[2 * kt + r - 10], (_Float16)__builtin_fmaf(pe[4 * kt + 2 * (r - 10) + 1], one,
                                                                         -(float)__builtin_bit_cast(f16x2, u[2 * kt + r - 10])[1])};
        unsigned w = __builtin_bit_cast(unsigned, pr);
        asm volatile("" : "+v"(w));              // pins v_fma_mixhi_f16 to this step
        r2[2 * kt + r - 10] = w;
      } else if (r == 12) {
        sum0 = (kt == 0) ? ad[2 * kt] : sum0 + ad[2 * kt];
        asm volatile("" : "+v"(sum0));
      } else {
        sum1 = (kt == 0) ? ad[2 * kt + 1] : sum1 + ad[2 * kt + 1];
        asm volatile("" : "+v"(sum1));
      }
    };
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      mfma_slot(i, G, pend, knext_base, knext_off, vcur_base, vcur_off);
      __builtin_amdgcn_sched_barrier(0);         // the MFMA leads its gap
#pragma unroll
      for (int n = (i ? SCHED.gap_end[i - 1] : 0); n < SCHED.gap_end[i]; ++n) vstep(SCHED.order[n]);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (!(H2W_ABL & 8)) {
      // any lane with sum >= P_TRIP: some P of this stage may not fit fp16 (they are >= 0)
      const unsigned long long cond = __builtin_amdgcn_ballot_w64(sum0 + sum1 >= P_TRIP);
      const float delta = h2w_rare_delta(cond, S[G], bp32);
      h2w_rare_scale_o(cond, delta, O[G]);
      h2w_rare_ref(cond, delta, a_slot0, negm[G], l_run[G], sum0, sum1);
#pragma unroll
      for (int kq = 0; kq < 2; ++kq)
        h2w_rare_exp_split(cond, delta, one, S[G][8 * kq], S[G][8 * kq + 1], S[G][8 * kq + 2], S[G][8 * kq + 3], S[G][8 * kq + 4],
                           S[G][8 * kq + 5], S[G][8 * kq + 6], S[G][8 * kq + 7], u[4 * kq], u[4 * kq + 1], r2[4 * kq], r2[4 * kq + 1],
                           u[4 * kq + 2], u[4 * kq + 3], r2[4 * kq + 2], r2[4 * kq + 3], sum0, sum1);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int w = 0; w < 4; ++w) { pop[G][0][ks][w] = u[4 * ks + w]; pop[G][1][ks][w] = r2[4 * ks + w]; }
    l_run[G] += sum0 + sum1;
  };

  // the very first stage of a query group (block 0 of tile 0): fixes the reference point; compiler-scheduled
  auto first_stage = [&](auto g_tag) __attribute__((always_inline)) {
    constexpr int G = decltype(g_tag)::value;
    if constexpr (G == 0) {
      // score chain of (0, 1) on K(0); K(1) is fetched afterwards; V(0) for the first P V
#pragma unroll
      for (int nq = 0; nq < 6; ++nq)
        if (!((HDIFF_MUTANT & 4) && TERM_K[nq] == 0 && TERM_Q[nq] == 2)) S[1] = mfma32(kop[TERM_K[nq]], qop[1][TERM_Q[nq]], nq == 0 ? negm[1] : S[1]);
#pragma unroll
      for (int p = 0; p < 3; ++p) kop[p] = lds_read(0, BLKB + p * 1024);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) vop[ks] = lds_read(0, VOFF + ks * 1024);
    } else {
      // score chain of (1, 0) on K(1) (its reference is final: group 0's first stage has run) and P V of (0, 0)
#pragma unroll
      for (int nq = 0; nq < 6; ++nq)
        if (!((HDIFF_MUTANT & 4) && TERM_K[nq] == 0 && TERM_Q[nq] == 2)) S[0] = mfma32(kop[TERM_K[nq]], qop[0][TERM_Q[nq]], nq == 0 ? negm[0] : S[0]);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        O[0] = mfma32h(vop[ks], pop[0][1][ks], O[0]);
        O[0] = mfma32h(vop[ks], pop[0][0][ks], O[0]);
      }
    }
    float mx = S[G][0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, S[G][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float nm = P_SHIFT - mx;
#pragma unroll
    for (int r = 0; r < 16; ++r) { negm[G][r] = nm; S[G][r] += nm; }
    float sum0 = 0.f, sum1 = 0.f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      const float p0 = __builtin_amdgcn_exp2f(S[G][4 * kt]), p1 = __builtin_amdgcn_exp2f(S[G][4 * kt + 1]);
      const float p2 = __builtin_amdgcn_exp2f(S[G][4 * kt + 2]), p3 = __builtin_amdgcn_exp2f(S[G][4 * kt + 3]);
      sum0 += p0 + p2;
      sum1 += p1 + p3;
      unsigned a0, a1, c0, c1;
      split2(p0, p1, one, a0, a1);
      split2(p2, p3, one, c0, c1);
      const int ks = kt >> 1, o = (kt & 1) * 2;
      pop[G][0][ks][o] = a0; pop[G][1][ks][o] = a1;
      pop[G][0][ks][o + 1] = c0; pop[G][1][ks][o + 1] = c1;
    }
    l_run[G] = sum0 + sum1;
  };

  // ---- prologue: tiles 0 and 1 in flight, K(0) -> score chain of (0, 0)
  dma_tile(0, 0);
  if (ntiles > 1) dma_tile(1, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int p = 0; p < 3; ++p) kop[p] = lds_read(0, p * 1024);
#pragma unroll
  for (int nq = 0; nq < 6; ++nq)
    if (!((HDIFF_MUTANT & 4) && TERM_K[nq] == 0 && TERM_Q[nq] == 2)) S[0] = mfma32(kop[TERM_K[nq]], qop[0][TERM_Q[nq]], nq == 0 ? negm[0] : S[0]);

  // Tile t lives in buffer t % 3 (byte offset cur; nxt = tile t + 1's).  Stage s = 2 kb + G of tile t; block kb + 1 of the last
  // block is block 0 of tile t + 1.  The barrier sits in front of the first read of tile t + 1 (the K reload of stage
  // 2 TB - 2): every wave has waited for its own pieces of that tile by then, and behind it nobody reads tile t - 1's buffer
  // any more, which takes tile t + 2.
  auto tile_fn = [&](auto first_tag, int t, int cur, int nxt) __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(first_tag)::value;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int kb = s >> 1;
      if (s == NS - 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!(H2W_ABL & 1)) __builtin_amdgcn_s_barrier();
        if (t + 2 < ntiles) dma_tile(t + 2, (t + 2) % 3);
      }
      const bool lastb = kb == TB - 1;
      const int knext_base = lastb ? nxt : cur, knext_off = lastb ? 0 : (kb + 1) * BLKB;
      if (FIRST && s == 0) first_stage(std::integral_constant<int, 0>{});
      else if (FIRST && s == 1) first_stage(std::integral_constant<int, 1>{});
      else if ((s & 1) == 0) stage_fn(std::integral_constant<int, 0>{}, true, knext_base, knext_off, cur, kb * BLKB);
      else stage_fn(std::integral_constant<int, 1>{}, true, knext_base, knext_off, cur, kb * BLKB);
    }
  };
#if H2W_DIAG
  const unsigned long long diag_t0 = __builtin_amdgcn_s_memtime(), diag_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  tile_fn(std::true_type{}, 0, 0, TILEB);
  int cur = TILEB;
  for (int t = 1; t < ntiles; ++t) {
    const int nxt = (cur == 2 * TILEB) ? 0 : cur + TILEB;
    tile_fn(std::false_type{}, t, cur, nxt);
    cur = nxt;
  }
#if H2W_DIAG
  if (tid == 0) {
    unsigned long long* dg = reinterpret_cast<unsigned long long*>(const_cast<unsigned char*>(pair) + vinv_offset(L) + 64) + 2 * tile.x;
    dg[0] = __builtin_amdgcn_s_memtime() - diag_t0;
    dg[1] = __builtin_amdgcn_s_memrealtime() - diag_r0;
  }
#endif
  // P V of the last stage (group 1 of the last block; vop still holds that block's V)
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    O[1] = mfma32h(vop[ks], pop[1][1][ks], O[1]);
    O[1] = mfma32h(vop[ks], pop[1][0][ks], O[1]);
  }

  float* obase = out + ((size_t)b * C + (size_t)head * D) * L;
  const float* vinv = reinterpret_cast<const float*>(pair + vinv_offset(L));      // 2^-s per channel of this head
#pragma unroll
  for (int G = 0; G < 2; ++G) {
    float lt = l_run[G];
    lt += __shfl_xor(lt, 32, 64);
    const bool bad = !(lt < OVERFLOW_LIMIT);            // NaN / inf inputs: hand this query block to the fp32 kernel's check pass
    const float inv = bad ? __builtin_nanf("") : 1.0f / lt;
    const int q = qblk0 + 32 * G + l31;
    if (lse2 != nullptr && h == 0)
      lse2[((size_t)b * heads + head) * L + q] = bad ? __builtin_nanf("") : __builtin_amdgcn_logf(lt) - negm[G][0];
    // accumulator register 4 j + i holds row 8 j + 4 h + i of [v0; v1] P: rows 0-15 the v0 half, rows 16-31 the v1 half
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int d = 8 * jj + 4 * h + i;
        obase[(size_t)d * L + q] = ((O[G][4 * jj + i] + O[G][4 * (jj + 2) + i]) * inv) * vinv[d];
      }
  }
}

}  // namespace

namespace hdiff {

// The d_head 16 forward on 32x32x16 tiles; its own operand layout in the workspace of hdiff_mha_flash_fwd_workspace (same size).
// Returns false when the shape is not covered or the workspace is missing.
bool launch_mha_fwd_h2w(const float* qkv, float* o, float* lse2, int B, int C, int heads, int L, float qscale, void* ws,
                        int64_t ws_bytes, hipStream_t stream) {
  const int64_t need = mha_fwd_x3p_workspace(B, C, heads, L);
  if (!mha_fwd_h2_enabled() || need == 0 || ws == nullptr || ws_bytes < need) return false;
  if (C / heads != 16 || L % 256 != 0) return false;
  static const char* e = getenv("HDIFF_H2W");       // dev knob (A/B inside one gpurun call) while the kernel is being tuned: 1 = this kernel
  if (!(e && atoi(e) == 1)) return false;
  hipLaunchKernelGGL(h2w_qk_split_kernel, dim3(cdiv(L, 256), 2 * heads, B), dim3(THREADS), 0, stream, qkv, (unsigned char*)ws, C, L, qscale);
  hipLaunchKernelGGL(h2w_v_split_kernel, dim3(C, B), dim3(THREADS), 0, stream, qkv, (unsigned char*)ws, C, L, 1.0f);
  hipLaunchKernelGGL((mha_flash_fwd_h2w_kernel<4>), dim3(L / 256, heads, B), dim3(THREADS), 0, stream, (const unsigned char*)ws, o, lse2,
                     C, L, 1.0f);
  return true;
}

}  // namespace hdiff
