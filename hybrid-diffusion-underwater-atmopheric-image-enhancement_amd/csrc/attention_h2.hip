// Flash attention forward, split-operand form, round 4: the P.V product on fp16 PAIRS instead of bf16 triples.
// Same contract and data layout as attention.hip / attention_x3.hip (reference: nn.MultiheadAttention core,
// ModelCondition.py:189, 204-208); reads the pre-split workspace of attention_x3p.hip for Q and K.
//
// Why (profiles/r03_pmc_summary.txt, tools/h2_probe.hip): the bf16-triple kernel is bound by the SUM of its vector and matrix
// issue time, and 5.5 of its 8 vector instructions per score split P = exp2(S) into three bf16 pieces.  fp16 carries 11
// significand bits, so TWO pieces hold 22-23 of P's 24 bits, and gfx950 has the instructions to make them in 1.5 per value:
//     h0 = v_cvt_pk_f16_f32(p_a, p_b)                      round-to-nearest-even pair (produces fp16 subnormals)
//     h1.lo = v_fma_mixlo_f16(p_a, 1.0, -h0.lo)            fp32 fma (the residual is exact), ONE rounding to fp16
//     h1.hi = v_fma_mixhi_f16(p_b, 1.0, -h0.hi)
// (the compiler emits exactly these from the plain C below when its SLP vectoriser is off -- with it the residuals become a
// v_pk_fma_f32, which stalls against the MFMA stream; all three issue like v_perm / v_and beside the MFMA, h2_probe part B).
//
//   S  = Q K^T : round 5 -- fp16 pairs with a balance PER TERM, four products on TWO MFMAs per 16x16 tile (bf16 triples: six on
//                three).  Round 4 had rejected pairs for the scores: S is an exponent, it must come out unscaled, and one scale
//                pair q 2^a, k 2^-a cannot keep both SECOND pieces above fp16's 2^-3 full-precision floor.  But every product
//                term is its own set of contraction slots and carries its own balance:
//                    k = k0 + k1, q = q0 + q1 (fp16 roundings of k 2^a, q 2^-a; a per (sample, head) from the maxima)
//                    S = k0 q0 + (k0 2^-8)(q1 2^8)  |  (k1 2^8)(q0 2^-8) + k1 q1
//                the second pieces are stored scaled UP by 2^8 (normal numbers whenever the value is), their partners scaled
//                DOWN, where fp16's absolute floor 2^-25 meets a factor 2^-3 |x| -- harmless, and only the ABSOLUTE error of an
//                exponent matters.  tools/h2_sim_qk_terms.py: error of S against float64 1.02x the fp32 chain's (Gaussian,
//                peaked, spiked, ramped keys, |q| 2^-10 / |k| 2^10, channels nine binades apart); three products: 1.14x.
//                The chain starts from -m and adds the large MFMA first.  Why it pays although this kernel is bound by vector
//                issue: it is ALSO at the board's power limit (profiles/r05_attention_h2w_32x32.txt) -- a third fewer score
//                products are joules, 32 cycles less held issue per stage, and 16 registers of Q operands.
//   O += P V   : P = h0 + h1 with |P - h0 - h1| <= 2^-23 P (one fp32 ulp: the second piece keeps 11 of the residual's 12
//                bits) or <= 2^-25 absolute (fp16 subnormal spacing 2^-24); V likewise as two fp16 pieces of V * 2^s with
//                the power of two s chosen PER CHANNEL ROW so that max |V 2^s| lies in [2^14, 2^15) (v_split_h2_kernel; O is
//                multiplied by 2^-s at the end, exactly).  Products kept: h0 v0, h0 v1, h1 v0 (each exact in the fp32
//                accumulator); dropped: h1 v1 <= 2^-22 |P V|.  So every product P_k V_k enters with a relative error of a
//                few 2^-23 -- random in sign (round to nearest) -- where the fp32-MFMA kernel rounds its RUNNING SUM to
//                2^-24 at every k-step: over L >= 512 keys the chain's error is the larger one.  tools/h2_sim.py emulates
//                both against float64; tests/test_gpu_ops.py holds this kernel to the same gate as the bf16 triples (error
//                against float64 <= 1.25x rms / 2x max of the fp32-MFMA kernel's), also for peaked rows, rows whose
//                maximum moves by 2^40 along the keys, and V channels spanning 2^30.
//   range     : fp16 ends at 65504, so the fixed softmax reference of the fp32 / bf16 kernels (first key tile's maximum,
//                never moved, overflow -> NaN -> check pass) is not enough.  Here m starts as (first tile's maximum - 8),
//                i.e. that maximum enters as P = 2^8, and MOVES when needed: every stage compares the sum of the 16 P values
//                a lane has just made against 2^15 (they are >= 0, so a smaller sum means every one of them is below 2^15);
//                a wave in which any lane trips recomputes that stage from its S accumulators (still in registers) under
//                the new reference max - 8, after scaling O and l by the exact power of two.  One add, one compare and one
//                scalar branch per 16 scores; the rare path costs a few hundred cycles per moved maximum.
//                l >= 2^8 at the end (the element that set the last reference contributes 2^8), so the absolute 2^-25
//                floor of small P values is <= 2^-33 l per element.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

using namespace hdiff;

namespace {

#ifndef H2_ABL
#define H2_ABL 0      // timing ablations (wrong results by construction; tools/README.md): 1 no workgroup barrier, 2 no rolling
#endif                // K / Q reloads, 4 no global -> LDS staging in the loop, 8 no reference check, 16 no V reload
#ifndef H2_DIAG
#define H2_DIAG 0       // diagnostic build: s_memtime / s_memrealtime around the tile loop (tools/h2w_clock.py); never set in the product
#endif
#ifndef H2_T4
#define H2_T4 1       // dev: 0 = the fourth score term k1 q1 (2^-24 of a score) multiplied by ZEROS -- same MFMAs, a quarter of the score product's
#endif                // multipliers idle (A/B for the energy of the kernel at the board's power limit; tools/scripts/r6_fwd_t4.sh)
#ifndef H2_VALU_PER_STAGE
#define H2_VALU_PER_STAGE 54
#endif
constexpr int KT = 64;
constexpr int THREADS = 256;
constexpr float OVERFLOW_LIMIT = 1.2379400e27f;   // 2^90: only NaN / inf inputs get here (the reference moves before fp16 overflows)
constexpr float P_SHIFT = 8.0f;                   // the reference point enters as P = 2^8
constexpr float P_TRIP = 32768.0f;                // per-lane sum of one stage's 16 P values that moves the reference

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma_f16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// (a, b) -> two packed fp16 pairs with a = h0.lo + h1.lo up to 2^-23 |a| (or 2^-25 absolute), b likewise in the high halves.
// `one` is 1.0f in a register the compiler cannot see through: fma(a, 1, -h) must stay an fma (v_fma_mixlo_f16), a - h
// would be a conversion and a subtraction.
__device__ __forceinline__ void split2(float a, float b, float one, unsigned& h0, unsigned& h1) {
  const f16x2 p = {(_Float16)a, (_Float16)b};                 // v_cvt_pk_f16_f32: round to nearest even
  unsigned u = __builtin_bit_cast(unsigned, p);
  asm("" : "+v"(u));                                          // the halves are read back out of the packed register
  const f16x2 q = __builtin_bit_cast(f16x2, u);
  const f16x2 r = {(_Float16)__builtin_fmaf(a, one, -(float)q[0]), (_Float16)__builtin_fmaf(b, one, -(float)q[1])};
  h0 = u;
  h1 = __builtin_bit_cast(unsigned, r);
}

// ---------------------------------------------------------------------------------------------------------------------
// Moving the softmax reference, exactly, in the stage that needs it.  A stage whose lane sum of 16 P values reached P_TRIP
// (so some P may not fit fp16) is made again from its S accumulators, which are still in registers, under a reference
// that puts the row's maximum over this key tile at 2^8, after O, l and -m of that query tile have been rescaled by the
// exact power of two.  The whole wave takes the path when one lane trips; rows that do not need it get delta = 0 and
// the same bits as before.
// Form: a handful of asm statements, each "branch on a saved scalar condition to code kept out of line (.subsection 1),
// come back".  Every compiler-visible form of this branch -- if, if + tied moves, a loop -- was measured: the backend
// structurises uniform branches too and the merge of the two paths cannot share registers across its flow blocks, which put
// 18 register copies per stage on the common path (173 ms against 150 ms without any check at L = 65 536, batch 16).  Here the
// common path costs one add, one compare and six scalar branches that are not taken, and the results of the rare path land
// in the registers the common path uses (tied operands).  The hazard recogniser does not look inside asm statements: the
// s_nop / operand order below provide the wait states themselves (transcendental -> use, cvt_pk / mixlo partial writes ->
// use, VALU -> ds_bpermute).
// ---------------------------------------------------------------------------------------------------------------------
#define H2_RARE_BEGIN "s_cmp_lg_u64 %[cond], 0\n\ts_cbranch_scc1 .Lh2r_%=\n.Lh2b_%=:\n\t.subsection 1\n.Lh2r_%=:\n\t"
#define H2_RARE_END "s_branch .Lh2b_%=\n\t.subsection 0"

// the row's maximum of S (= s - m) over this stage's 64 keys and the move: delta = mx > 9 ? ceil(mx - 8) : 0
__device__ __forceinline__ float h2_rare_delta(unsigned long long cond, const f32x4 (&S)[4], int bp16, int bp32) {
  float delta, t, u;
  asm volatile(H2_RARE_BEGIN
               "v_max3_f32 %[t], %[s0], %[s1], %[s2]\n\t"
               "v_max3_f32 %[u], %[s3], %[s4], %[s5]\n\t"
               "v_max3_f32 %[t], %[t], %[s6], %[s7]\n\t"
               "v_max3_f32 %[u], %[u], %[s8], %[s9]\n\t"
               "v_max3_f32 %[t], %[t], %[s10], %[s11]\n\t"
               "v_max3_f32 %[u], %[u], %[s12], %[s13]\n\t"
               "v_max3_f32 %[t], %[t], %[s14], %[s15]\n\t"
               "v_max_f32 %[t], %[t], %[u]\n\t"
               "s_nop 1\n\t"
               "ds_bpermute_b32 %[u], %[bp16], %[t]\n\t"
               "s_waitcnt lgkmcnt(0)\n\t"
               "v_max_f32 %[t], %[t], %[u]\n\t"
               "s_nop 1\n\t"
               "ds_bpermute_b32 %[u], %[bp32], %[t]\n\t"
               "s_waitcnt lgkmcnt(0)\n\t"
               "v_max_f32 %[t], %[t], %[u]\n\t"              // over the four lanes that share the query
               "v_subrev_f32 %[u], 8.0, %[t]\n\t"
               "v_ceil_f32 %[u], %[u]\n\t"
               "v_cmp_lt_f32 vcc, 0x41100000, %[t]\n\t"      // 9.0 < mx
               "v_cndmask_b32 %[d], 0, %[u], vcc\n\t"
               H2_RARE_END
               : [d] "=&v"(delta), [t] "=&v"(t), [u] "=&v"(u)
               : [cond] "s"(cond), [bp16] "v"(bp16), [bp32] "v"(bp32), [s0] "v"(S[0][0]), [s1] "v"(S[0][1]), [s2] "v"(S[0][2]),
                 [s3] "v"(S[0][3]), [s4] "v"(S[1][0]), [s5] "v"(S[1][1]), [s6] "v"(S[1][2]), [s7] "v"(S[1][3]), [s8] "v"(S[2][0]),
                 [s9] "v"(S[2][1]), [s10] "v"(S[2][2]), [s11] "v"(S[2][3]), [s12] "v"(S[3][0]), [s13] "v"(S[3][1]),
                 [s14] "v"(S[3][2]), [s15] "v"(S[3][3])
               : "vcc", "scc");
  return delta;      // defined only on the rare path; only the rare path reads it
}

// O, l *= 2^-delta; -m -= delta; the lane's row sums restart from zero
__device__ __forceinline__ void h2_rare_rescale(unsigned long long cond, float delta, f32x4& O, f32x4& negm, float& l, float& sum0,
                                                float& sum1) {
  float o0 = O[0], o1 = O[1], o2 = O[2], o3 = O[3], n0 = negm[0], n1 = negm[1], n2 = negm[2], n3 = negm[3], w;
  asm volatile(H2_RARE_BEGIN
               "v_cvt_i32_f32 %[w], %[d]\n\t"
               "v_sub_u32 %[w], 0, %[w]\n\t"
               "v_ldexp_f32 %[o0], %[o0], %[w]\n\t"
               "v_ldexp_f32 %[o1], %[o1], %[w]\n\t"
               "v_ldexp_f32 %[o2], %[o2], %[w]\n\t"
               "v_ldexp_f32 %[o3], %[o3], %[w]\n\t"
               "v_ldexp_f32 %[l], %[l], %[w]\n\t"
               "v_sub_f32 %[n0], %[n0], %[d]\n\t"
               "v_sub_f32 %[n1], %[n1], %[d]\n\t"
               "v_sub_f32 %[n2], %[n2], %[d]\n\t"
               "v_sub_f32 %[n3], %[n3], %[d]\n\t"
               "v_mov_b32 %[q0], 0\n\t"
               "v_mov_b32 %[q1], 0\n\t"
               H2_RARE_END
               : [o0] "+v"(o0), [o1] "+v"(o1), [o2] "+v"(o2), [o3] "+v"(o3), [n0] "+v"(n0), [n1] "+v"(n1), [n2] "+v"(n2),
                 [n3] "+v"(n3), [l] "+v"(l), [q0] "+v"(sum0), [q1] "+v"(sum1), [w] "=&v"(w)
               : [cond] "s"(cond), [d] "v"(delta)
               : "scc");
  O = f32x4{o0, o1, o2, o3};
  negm = f32x4{n0, n1, n2, n3};
}

// four scores of one 16-key tile again: P = exp2(S - delta), its fp16 pieces and the row sums -- the common path's
// instructions in the common path's order, so a row with delta = 0 gets its bits back
__device__ __forceinline__ void h2_rare_exp_split(unsigned long long cond, float delta, float one, f32x4 S, unsigned& a0, unsigned& a1,
                                                  unsigned& r0, unsigned& r1, float& sum0, float& sum1) {
  float p0, p1, p2, p3, t0, t1;
  asm volatile(H2_RARE_BEGIN
               "v_sub_f32 %[p0], %[s0], %[d]\n\t"
               "v_sub_f32 %[p1], %[s1], %[d]\n\t"
               "v_sub_f32 %[p2], %[s2], %[d]\n\t"
               "v_sub_f32 %[p3], %[s3], %[d]\n\t"
               "v_exp_f32 %[p0], %[p0]\n\t"
               "v_exp_f32 %[p1], %[p1]\n\t"
               "v_exp_f32 %[p2], %[p2]\n\t"
               "v_exp_f32 %[p3], %[p3]\n\t"
               "s_nop 1\n\t"
               "v_cvt_pk_f16_f32 %[a0], %[p0], %[p1]\n\t"
               "v_cvt_pk_f16_f32 %[a1], %[p2], %[p3]\n\t"
               "v_add_f32 %[t0], %[p0], %[p2]\n\t"
               "v_add_f32 %[t1], %[p1], %[p3]\n\t"
               "v_fma_mixlo_f16 %[r0], %[p0], %[one], -%[a0] op_sel_hi:[0,0,1]\n\t"
               "v_fma_mixlo_f16 %[r1], %[p2], %[one], -%[a1] op_sel_hi:[0,0,1]\n\t"
               "v_add_f32 %[q0], %[q0], %[t0]\n\t"
               "v_add_f32 %[q1], %[q1], %[t1]\n\t"
               "v_fma_mixhi_f16 %[r0], %[p1], %[one], -%[a0] op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
               "v_fma_mixhi_f16 %[r1], %[p3], %[one], -%[a1] op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
               "s_nop 1\n\t"
               H2_RARE_END
               : [a0] "+v"(a0), [a1] "+v"(a1), [r0] "+v"(r0), [r1] "+v"(r1), [q0] "+v"(sum0), [q1] "+v"(sum1), [p0] "=&v"(p0),
                 [p1] "=&v"(p1), [p2] "=&v"(p2), [p3] "=&v"(p3), [t0] "=&v"(t0), [t1] "=&v"(t1)
               : [cond] "s"(cond), [d] "v"(delta), [one] "s"(one), [s0] "v"(S[0]), [s1] "v"(S[1]), [s2] "v"(S[2]), [s3] "v"(S[3])
               : "scc");
}

// Score operands.  K pieces in the workspace / LDS: 0 = k0, 1 = k0 2^-8, 2 = k1 2^8, 3 = k1; Q pieces in registers: q0, q1 2^8 read
// from the workspace, q0 2^-8 and q1 made from them.  MFMA j contracts K operand set j (pieces 2 j | 2 j + 1 on the low | high 16
// slots) with Q operand set j: MFMA 0 = k0 q0 + (k0 2^-8)(q1 2^8), MFMA 1 = (k1 2^8)(q0 2^-8) + k1 q1.
constexpr int QK_SHIFT = 8;

// ---------------------------------------------------------------------------------------------------------------------
// Q and K of qkv [B][3C][L] (fp32) -> fp16 score operands (see QK_SHIFT) in the workspace, per (sample, head):
//   piece 0: q0 [L][D], 1: q1 2^8 [L][D], 2..5: k0, k0 2^-8, k1 2^8, k1 [L][D]   (q = q_in qscale 2^-a, k = k_in 2^a)
// Pass 1 (qk_rowmax_kernel, grid (2C, B)): max |x| of every Q / K channel row into rowmax[B][2C] behind the pairs.
// Pass 2 (qk_split_h2_kernel, grid (L / 256, 2 heads, B)): thread = one position, all D channels; the balance a puts the two
// maxima of the head in the same binade (both then sit ~2^1..2^4 for O(1) scores: 2^11 below fp16's top, 2^15 above its
// normal floor); clamped so that every power of two stays a normal float.  An infinite or NaN input stays one (the row's
// output is then NaN and the caller's check pass takes the query block, as for every kernel of this family).
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(THREADS) void qk_rowmax_kernel(const float* __restrict__ qkv, float* __restrict__ rowmax, int C, int L) {
  const int row = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const float* src = qkv + ((size_t)b * 3 * C + row) * L;
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
  if (tid == 0) rowmax[(size_t)b * 2 * C + row] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

template <int D>
__global__ __launch_bounds__(THREADS) void qk_split_h2_kernel(const float* __restrict__ qkv, const float* __restrict__ rowmax,
                                                              __bf16* __restrict__ ws, int C, int L, float qscale, float one) {
  const int heads = C / D;
  const int which = blockIdx.y / heads, head = blockIdx.y - which * heads, b = blockIdx.z;
  const int l = blockIdx.x * THREADS + threadIdx.x;
  // the head's balance: exponents of max |q| qscale and max |k| (every thread reads the 2 D row maxima: 128 bytes, L2-resident)
  float mq = 0.f, mk = 0.f;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    mq = fmaxf(mq, rowmax[(size_t)b * 2 * C + head * D + d]);
    mk = fmaxf(mk, rowmax[(size_t)b * 2 * C + C + head * D + d]);
  }
  mq *= qscale;
  const int eq = (int)((__builtin_bit_cast(unsigned, mq) >> 23) & 0xffu), ek = (int)((__builtin_bit_cast(unsigned, mk) >> 23) & 0xffu);
  int a = (eq == 0 || ek == 0 || eq == 255 || ek == 255) ? 0 : (eq - ek) / 2;      // k 2^a, q 2^-a (zero / inf / NaN rows: no balance)
  a = a < -60 ? -60 : (a > 60 ? 60 : a);
  const float sc = which == 0 ? qscale * __builtin_bit_cast(float, (unsigned)(127 - a) << 23) : __builtin_bit_cast(float, (unsigned)(127 + a) << 23);
  if (l >= L) return;
  const float* src = qkv + ((size_t)b * 3 * C + (size_t)which * C + (size_t)head * D) * L;
  const size_t piece = (size_t)L * D;
  __bf16* pair = ws + ((size_t)b * heads + head) * 9 * piece;
  const float up = (float)(1 << QK_SHIFT) * one;
  typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
  const f16x2v dn = {(_Float16)(1.0f / (1 << QK_SHIFT)), (_Float16)(1.0f / (1 << QK_SHIFT))};
  unsigned h[4][D / 2];          // x0, x0 2^-8, x1 2^8, x1 as packed fp16 pairs (channels 2 j, 2 j + 1)
#pragma unroll
  for (int j = 0; j < D / 2; ++j) {
    float xa = src[(size_t)(2 * j) * L + l] * sc, xb = src[(size_t)(2 * j + 1) * L + l] * sc;
    // the fp32 products and the packed first pieces are made opaque: left alone the compiler rounds x0 twice -- once from the
    // fp32 product for the stored piece, once from the EXACT product (v_fma_mixlo_f16) for the residual -- and where the two
    // differ by an ulp the stored pieces no longer add up (found by tools/h2_qk_debug.py: 4e-4 instead of 5e-7)
    asm("" : "+v"(xa), "+v"(xb));
    unsigned u0 = __builtin_bit_cast(unsigned, f16x2v{(_Float16)xa, (_Float16)xb});
    asm("" : "+v"(u0));
    const f16x2v x0 = __builtin_bit_cast(f16x2v, u0);
    const float ra = xa - (float)x0[0], rb = xb - (float)x0[1];            // exact
    const f16x2v x1s = {(_Float16)(ra * up), (_Float16)(rb * up)}, x1 = {(_Float16)ra, (_Float16)rb};
    h[0][j] = __builtin_bit_cast(unsigned, x0);
    h[1][j] = __builtin_bit_cast(unsigned, x0 * dn);
    h[2][j] = __builtin_bit_cast(unsigned, x1s);
    h[3][j] = __builtin_bit_cast(unsigned, x1);
  }
  // Q: pieces 0 (q0) and 1 (q1 2^8); K: pieces 2 .. 5
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    if (which == 0 && (p == 1 || p == 3)) continue;
    const int slot = which == 0 ? (p == 0 ? 0 : 1) : 2 + p;
    u32x4* o = reinterpret_cast<u32x4*>(pair + slot * piece + (size_t)l * D);
#pragma unroll
    for (int j = 0; j < D / 8; ++j) o[j] = u32x4{h[p][4 * j], h[p][4 * j + 1], h[p][4 * j + 2], h[p][4 * j + 3]};
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// V of qkv [B][3C][L] (fp32)  ->  two fp16 pieces of V * 2^s, s per channel row, in the V region of the pre-split workspace:
// per (sample, head)  Vh[2][D][L] (fp16) at piece slot 6, and the D factors 2^-s (fp32) at piece slot 8.
// One workgroup per (sample, channel) row: a maximum pass, then the split pass (the row comes back from L2).
// ---------------------------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(THREADS) void v_split_h2_kernel(const float* __restrict__ qkv, __bf16* __restrict__ ws, int C, int L,
                                                             float one) {
  const int heads = C / D;
  const int row = blockIdx.x, head = row / D, d = row - head * D, b = blockIdx.y;
  const int tid = threadIdx.x;
  const float* src = qkv + ((size_t)b * 3 * C + 2 * (size_t)C + row) * L;
  const size_t piece = (size_t)L * D;
  __bf16* pair = ws + ((size_t)b * heads + head) * 9 * piece;
  _Float16* dst = reinterpret_cast<_Float16*>(pair + 6 * piece) + (size_t)d * L;
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
  // 2^s with max |v| 2^s in [2^14, 2^15); exponent clamped so that both 2^s and 2^-s are normal numbers (an all-zero or
  // denormal row is scaled by 2^114 at most, an infinite one by 2^-113: inf / NaN elements stay inf / NaN in fp16)
  int e = (int)((__builtin_bit_cast(unsigned, amax) >> 23) & 0xffu) - 127;
  e = e < -100 ? -100 : (e > 127 ? 127 : e);
  const float scale = __builtin_bit_cast(float, (unsigned)(14 - e + 127) << 23);
  if (tid == 0) reinterpret_cast<float*>(pair + 8 * piece)[d] = __builtin_bit_cast(float, (unsigned)(e - 14 + 127) << 23);
  for (int i = tid; i < L / 4; i += THREADS) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + 4 * (size_t)i);
    unsigned a0, a1, c0, c1;
    split2(v[0] * scale, v[1] * scale, one, a0, a1);
    split2(v[2] * scale, v[3] * scale, one, c0, c1);
    *reinterpret_cast<u32x2*>(dst + 4 * (size_t)i) = u32x2{a0, c0};
    *reinterpret_cast<u32x2*>(dst + piece + 4 * (size_t)i) = u32x2{a1, c1};
  }
}

// ---------------------------------------------------------------------------------------------------------------------
#ifndef H2_DMA
#define H2_DMA 1             // K tiles global -> LDS by LDS-DMA (0: through registers like V; dev A/B)
#endif
typedef __attribute__((address_space(3))) unsigned char lds_byte;
// One 1 KiB run global -> LDS without staging registers: lane i's 16 bytes at src + voff land at lds_dst + 16 i (global_load_lds_dwordx4).
// M0 is written in the statement that uses it and restored (cdna_hip_programming.md, 'What hipcc does not do').  The compiler does not
// count this load: the kernel waits with its own s_waitcnt vmcnt(0) in front of the barrier that publishes the tile.
__device__ __forceinline__ void dma_1k(const unsigned char* src, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(src), "s"(lds_dst)
               : "memory");
}

template <int D, int NQ>
__global__ __launch_bounds__(THREADS, 2) void mha_flash_fwd_h2_kernel(const __bf16* __restrict__ ws, float* __restrict__ out,
                                                                      float* __restrict__ lse2, int C, int L, float one) {
  static_assert(D == 16, "head dim: two terms share one MFMA's 32 contraction slots");
  static_assert(NQ == 4, "the stage pipeline is written for four query tiles per wave");
  constexpr int NKP = 4;                   // K pieces (see QK_SHIFT)
  constexpr int NKS = 2;                   // K operand sets per 16 keys = QK^T MFMAs per 16x16 score tile
  constexpr int NQK = NKS;
  constexpr int MT = D / 16;               // 16-row tiles of the output
  constexpr int KROWB = D * 2;             // bytes per key of one K piece
  constexpr int KPART = KT * KROWB;
  constexpr int VROWB = KT * 2 + 8;        // bytes per d row of one V piece (+8: the 16 rows of an operand read spread over banks)
  constexpr int VPART = D * VROWB;
  constexpr int QB = 64 * NQ;              // queries per workgroup (4 waves x NQ tiles of 16)
  constexpr int VBASE = NKP * KPART;
  constexpr int BUFB = VBASE + 2 * VPART;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2][BUFB];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  const TileId tile = xcd_tile();
  const int head = tile.head, b = tile.b;
  const int qblk0 = tile.x * QB + wave * (16 * NQ);
  const int ntiles = L / KT;

  // contraction slots of this lane: 8 consecutive d of one term
  const int doff = 8 * (g & 1);
  const bool hi = g >> 1;

  const size_t piece_n = (size_t)L * D;
  const __bf16* wsq = ws + ((size_t)b * gridDim.y + head) * 9 * piece_n;
  // Q operands of the wave's four query tiles, in registers for the whole kernel: [query tile][MFMA].  The workspace holds q0 and
  // q1 2^8; q0 2^-8 and q1 are their multiples (v_pk_mul_f16 by 2^-8: exact, or rounded into fp16's subnormals like the split
  // pass would have).
  u32x4 qop[NQ][NQK];
  {
    static_assert(QK_SHIFT == 8, "packed fp16 constant below");
    const unsigned dn2 = 0x1c001c00u;                // (2^-8, 2^-8) as packed fp16
#pragma unroll
    for (int qt = 0; qt < NQ; ++qt) {
      const int q = qblk0 + qt * 16 + i16;
      const u32x4 q0 = *reinterpret_cast<const u32x4*>(wsq + (size_t)q * D + doff);
      const u32x4 q1s = *reinterpret_cast<const u32x4*>(wsq + piece_n + (size_t)q * D + doff);
      const u32x4 sel = hi ? q1s : q0;  // this lane's half of the pairing: (q0, q0 2^-8) on the low slots, (q1 2^8, q1) on the high ones
      // (asm: written as four v2f16 multiplications in C, hipcc 7.2 -O3 emits ONE v_pk_mul_f16 and broadcasts word 0 over the tuple)
      unsigned dnw[4];
#pragma unroll
      for (int w = 0; w < 4; ++w) asm("v_pk_mul_f16 %0, %1, %2" : "=v"(dnw[w]) : "v"(sel[w]), "v"((!H2_T4 && hi) ? 0u : dn2));
      qop[qt][0] = sel;                                      // against K set 0 = (k0 | k0 2^-8)
      qop[qt][1] = u32x4{dnw[0], dnw[1], dnw[2], dnw[3]};    // against K set 1 = (k1 2^8 | k1)
    }
  }
  int kaddr[NKS];          // K operand set s: the A operand of MFMA s
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) kaddr[ks] = (2 * ks + (hi ? 1 : 0)) * KPART + i16 * KROWB + doff * 2;
  const int vaddr = i16 * VROWB + 8 * g;

  // staging: chunk c = i * 256 + tid of the tile's 16-byte chunks (four K pieces, then two V pieces), copied as they are
  // (H2_DMA) the four K pieces of a tile are eight contiguous 1 KiB runs in the workspace AND in LDS ([piece][key][32 bytes], no padding):
  // each wave copies two of them by LDS-DMA -- no staging registers, no ds_write, no per-thread addresses; only V (padded rows) goes
  // through registers
  constexpr bool DMA = (H2_DMA != 0);
  constexpr int NKC = DMA ? 0 : NKP * KT * D / 8, NVC = 2 * D * 8, NCH = NKC + NVC, NLD = (NCH + THREADS - 1) / THREADS;
  static_assert(NLD * THREADS - NCH <= NVC, "staging geometry");
  static_assert(!DMA || (KPART == 2048 && THREADS == 256), "LDS-DMA geometry: eight 1 KiB runs, two per wave");
  const unsigned char* gsrc[NLD];
  int lds_off[NLD], gstep[NLD];
  u32x4 stage[NLD];
  {
    const __bf16* ksp = wsq + 2 * piece_n;
    const __bf16* vsp = wsq + 6 * piece_n;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      int c = i * THREADS + tid;
      if (c >= NCH) c -= NLD * THREADS - NCH;
      if (c < NKC) {
        const int p = c / (KT * D / 8), rem = c - p * (KT * D / 8);
        gsrc[i] = reinterpret_cast<const unsigned char*>(ksp + p * piece_n) + (size_t)rem * 16;
        lds_off[i] = p * KPART + rem * 16;
        gstep[i] = KT * D * 2;
      } else {
        const int cv = c - NKC;
        const int p = cv / (D * 8), rem = cv - p * (D * 8);
        const int d = rem >> 3, seg = rem & 7;
        gsrc[i] = reinterpret_cast<const unsigned char*>(vsp + p * piece_n + (size_t)d * L) + seg * 16;
        lds_off[i] = VBASE + p * VPART + d * VROWB + seg * 16;
        gstep[i] = KT * 2;
      }
    }
  }
  const unsigned lds0 = (unsigned)(size_t)(lds_byte*)&smem[0][0];
  const unsigned char* kdma = reinterpret_cast<const unsigned char*>(wsq + 2 * piece_n);      // + piece * piece_n * 2 + tile * 2048 + half * 1024
  // wave w copies runs 2 w, 2 w + 1 of the tile: piece w, its two 32-key halves
  auto dma_k = [&](int t, int buf) {
    if (!DMA) return;
    const int ws_ = __builtin_amdgcn_readfirstlane(wave);      // the asm operands must be scalar registers
    const unsigned char* src = kdma + (size_t)ws_ * (piece_n * 2) + (size_t)t * KPART;
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + buf * BUFB + ws_ * KPART);
    dma_1k(src, lane * 16, dst);
    dma_1k(src + 1024, lane * 16, dst + 1024);
  };
  auto stage_load = [&](int t) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      // the per-tile step is a compile-time constant for the rounds that hold only K or only V chunks
      const int step = ((i + 1) * THREADS <= NKC) ? KT * D * 2 : (i * THREADS >= NKC ? KT * 2 : gstep[i]);
      stage[i] = *reinterpret_cast<const u32x4*>(gsrc[i] + (size_t)t * step);
    }
  };
  auto stage_store = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {       // two 8-byte stores for K and V chunks alike: no per-thread branch
      unsigned char* dst = &smem[buf][lds_off[i]];
      *reinterpret_cast<u32x2*>(dst) = u32x2{stage[i][0], stage[i][1]};
      *reinterpret_cast<u32x2*>(dst + 8) = u32x2{stage[i][2], stage[i][3]};
    }
  };

  f32x4 O[MT][NQ];
  f32x4 negm4[NQ];
  float l_run[NQ];
  const int bp16 = (lane ^ 16) * 4, bp32 = (lane ^ 32) * 4;
#pragma unroll
  for (int qt = 0; qt < NQ; ++qt) {
    l_run[qt] = 0.f;
    negm4[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) O[mt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // ---- operands and pipeline state held in registers across tiles
  u32x4 vop[2][MT][2];         // V of the current tile: [piece][row tile][32-key chunk]
  u32x4 kop[4][NKS];           // K of the tile whose scores are being made: [key tile][operand set]
  f32x4 S[2][4];               // scores of two consecutive stages
  u32x4 pop[2][2][2];          // P of two consecutive stages: [stage parity][piece][32-key chunk]
  auto load_v = [&](int buf) {
    const unsigned char* vb = smem[buf] + VBASE;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const unsigned char* src = vb + p * VPART + mt * 16 * VROWB + 64 * c + vaddr;
          const u32x2 lo = *reinterpret_cast<const u32x2*>(src);
          const u32x2 hi2 = *reinterpret_cast<const u32x2*>(src + 32);
          vop[p][mt][c] = u32x4{lo[0], lo[1], hi2[0], hi2[1]};
        }
  };
  auto load_k = [&](int buf) {
    const unsigned char* kb = smem[buf];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) kop[kt][ks] = *reinterpret_cast<const u32x4*>(kb + kaddr[ks] + kt * 16 * KROWB);
  };
  // MFMA n of Q K^T for query tile qt into S[par]: the four key tiles' chains round robin (a dependent pair is 4 apart)
  // rollk >= 0: this is the last query tile of the key tile -- K operand set j is free once MFMA j has been issued and is fetched
  // for the next key tile (buffer rollk) right behind it
  auto qk_mfma = [&](int qt, int par, int n, int rollk = -1) {
    const int j = n >> 2, kt = n & 3;
    u32x4 qb = qop[qt][j];
    if ((HDIFF_MUTANT & 4) && j == 1)            // (mutation test: the low five bits of the small Q pieces dropped: 2^-17 of q)
#pragma unroll
      for (int w = 0; w < 4; ++w) qb[w] &= 0xffe0ffe0u;
    S[par][kt] = mfma_f16(kop[kt][j], qb, j == 0 ? negm4[qt] : S[par][kt]);      // the chain starts from -m, large terms first
    if (H2_ABL & 2) return;
    if (rollk >= 0) kop[kt][j] = *reinterpret_cast<const u32x4*>(smem[rollk] + kaddr[j] + kt * 16 * KROWB);
  };
  constexpr int NPV = 6 * MT;
  // MFMA n of O[qt] += P V with P from pop[par]: per 32-key chunk and row tile the small terms first
  auto pv_mfma = [&](int qt, int par, int n) {
    const int c = n / (3 * MT), r = n - c * 3 * MT, mt = r / 3, term = r - mt * 3;
    const int pv_v = term == 0 ? 1 : 0, pv_p = term == 1 ? 1 : 0;                      // (v1, p0), (v0, p1), (v0, p0)
    u32x4 pb = pop[par][pv_p][c];
    if ((HDIFF_MUTANT & 8) && pv_p == 1)         // (mutation test: the low five bits of every second piece of P dropped: 2^-17 of P)
#pragma unroll
      for (int w = 0; w < 4; ++w) pb[w] &= 0xffe0ffe0u;
    O[mt][qt] = mfma_f16(vop[pv_v][mt][c], pb, O[mt][qt]);
  };
  // P = exp2(S) of one stage (16 queries x 64 keys of this wave; 16 values per lane), its two fp16 pieces packed as the
  // B operands of P.V, and the lane's two partial row sums
  auto exp_split = [&](const f32x4 (&Sq)[4], u32x4 (&pp)[2][2], float& sum0, float& sum1) {
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      const float p0 = __builtin_amdgcn_exp2f(Sq[kt][0]), p1 = __builtin_amdgcn_exp2f(Sq[kt][1]);
      const float p2 = __builtin_amdgcn_exp2f(Sq[kt][2]), p3 = __builtin_amdgcn_exp2f(Sq[kt][3]);
      sum0 = (kt == 0) ? p0 + p2 : sum0 + (p0 + p2);
      sum1 = (kt == 0) ? p1 + p3 : sum1 + (p1 + p3);
      const int c = kt >> 1, o = (kt & 1) * 2;
      unsigned a0, a1, c0, c1;
      split2(p0, p1, one, a0, a1);
      split2(p2, p3, one, c0, c1);
      pp[0][c][o] = a0; pp[1][c][o] = a1;
      pp[0][c][o + 1] = c0; pp[1][c][o + 1] = c1;
    }
  };
  auto stage_max = [&](int par) {
    float mx = fmaxf(fmaxf(S[par][0][0], S[par][0][1]), fmaxf(S[par][0][2], S[par][0][3]));
#pragma unroll
    for (int kt = 1; kt < 4; ++kt) mx = fmaxf(mx, fmaxf(fmaxf(S[par][kt][0], S[par][kt][1]), fmaxf(S[par][kt][2], S[par][kt][3])));
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    return fmaxf(mx, __shfl_xor(mx, 32, 64));
  };
  // One stage = the vector work of (tile, query tile QT): S[par] -> P pieces in pop[par] -- hand-interleaved with the
  // MFMAs of the neighbouring stages: Q K^T for the NEXT stage (into S[par ^ 1]; for QT = 3 that is query tile 0 of the
  // next key tile, whose K operands are in kop by then) and P V of the PREVIOUS one (pop[par ^ 1]).  The order is written
  // out and fenced slot by slot (one MFMA, about three vector instructions): left to the scheduler the MFMAs of a stage
  // clump, and a vector instruction only runs beside an MFMA, not instead of waiting for one.
  auto stage_fn = [&](auto qt_tag, auto first_tag, auto pend_tag, int rollk = -1) {
    constexpr int QT = decltype(qt_tag)::value;
    constexpr bool FIRST = decltype(first_tag)::value;       // first key tile: fixes the reference point
    constexpr bool PEND = decltype(pend_tag)::value;         // P V of the previous stage is pending
    constexpr int par = QT & 1;
    constexpr int qk_q = (QT + 1) % NQ, pv_q = (QT + NQ - 1) % NQ;
    constexpr int NQKM = 4 * NQK, NM = NQKM + (PEND ? NPV : 0);
    float sum0, sum1;
    if constexpr (FIRST) {
#pragma unroll
      for (int n = 0; n < NQKM; ++n) qk_mfma(qk_q, par ^ 1, n, rollk);
      if constexpr (PEND)
#pragma unroll
        for (int n = 0; n < NPV; ++n) pv_mfma(pv_q, par ^ 1, n);
      const float nm = P_SHIFT - stage_max(par);
      negm4[QT] = f32x4{nm, nm, nm, nm};
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) S[par][kt] += negm4[QT];
      exp_split(S[par], pop[par], sum0, sum1);
    } else {
      // the stage's vector instructions as 56 numbered steps (14 per key tile: 4 exp, 2 cvt_pk, 2 adds, 2 mixlo, 2 mixhi, 2 adds)
      float pe[4][4], ad[4][2];
      unsigned u[4][2];
      _Float16 rl[4][2];
      unsigned r2[4][2];
      auto vstep = [&](int n) {
        const int kt = n / 14, r = n - kt * 14;
        if (r < 4) pe[kt][r] = __builtin_amdgcn_exp2f(S[par][kt][r]);
        else if (r < 6) {
          const f16x2 p = {(_Float16)pe[kt][2 * (r - 4)], (_Float16)pe[kt][2 * (r - 4) + 1]};
          unsigned w = __builtin_bit_cast(unsigned, p);
          asm("" : "+v"(w));
          u[kt][r - 4] = w;
        } else if (r < 8) ad[kt][r - 6] = pe[kt][r - 6] + pe[kt][r - 6 + 2];
        else if (r < 10) rl[kt][r - 8] = (_Float16)__builtin_fmaf(pe[kt][2 * (r - 8)], one, -(float)__builtin_bit_cast(f16x2, u[kt][r - 8])[0]);
        else if (r < 12) {
          const f16x2 pr = {rl[kt][r - 10],
                            (_Float16)__builtin_fmaf(pe[kt][2 * (r - 10) + 1], one, -(float)__builtin_bit_cast(f16x2, u[kt][r - 10])[1])};
          unsigned w = __builtin_bit_cast(unsigned, pr);
          asm("" : "+v"(w));              // pins v_fma_mixhi_f16 to this step
          r2[kt][r - 10] = w;
        } else if (r == 12) sum0 = (kt == 0) ? ad[kt][0] : sum0 + ad[kt][0];
        else sum1 = (kt == 0) ? ad[kt][1] : sum1 + ad[kt][1];
      };
      constexpr int NV = 56;
#pragma unroll
      for (int i = 0; i < NM; ++i) {
        if constexpr (PEND) {          // the P V MFMAs spread evenly between the Q K^T ones: q p q q p q p q q p ...
          const int npv = (i + 1) * NPV / NM, ppv = i * NPV / NM;
          if (npv != ppv) pv_mfma(pv_q, par ^ 1, ppv);
          else qk_mfma(qk_q, par ^ 1, i - ppv, rollk);
        } else qk_mfma(qk_q, par ^ 1, i, rollk);
#pragma unroll
        for (int n = NV * i / NM; n < NV * (i + 1) / NM; ++n) vstep(n);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (!(H2_ABL & 8)) {
        // any lane with sum >= P_TRIP: some P of this stage may not fit fp16 (see above)
        const unsigned long long cond = __builtin_amdgcn_ballot_w64(sum0 + sum1 >= P_TRIP);
        const float delta = h2_rare_delta(cond, S[par], bp16, bp32);
        h2_rare_rescale(cond, delta, O[0][QT], negm4[QT], l_run[QT], sum0, sum1);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
          h2_rare_exp_split(cond, delta, one, S[par][kt], u[kt][0], u[kt][1], r2[kt][0], r2[kt][1], sum0, sum1);
      }
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        const int c = kt >> 1, o = (kt & 1) * 2;
        pop[par][0][c][o] = u[kt][0]; pop[par][1][c][o] = r2[kt][0];
        pop[par][0][c][o + 1] = u[kt][1]; pop[par][1][c][o + 1] = r2[kt][1];
      }
    }
    l_run[QT] += sum0 + sum1;
  };

  // Key tile t (buffer t & 1).  On entry: kop = K(t), S[0] = scores of (t, query tile 0), global loads of tile t + 1 in
  // flight; unless FIRST, vop = V(t - 1) and pop[1] = P(t - 1, 3) with its P V pending.
  auto tile_fn = [&](auto first_tag, int t) {
    constexpr bool FIRST = decltype(first_tag)::value;
    using Pend = std::integral_constant<bool, !FIRST>;
    const int buf = t & 1;
    if constexpr (FIRST) load_v(buf);
    stage_fn(std::integral_constant<int, 0>{}, first_tag, Pend{});
    if constexpr (!FIRST) if (!(H2_ABL & 16)) load_v(buf);
    stage_fn(std::integral_constant<int, 1>{}, first_tag, std::true_type{});
    if (!(H2_ABL & 4)) stage_store(buf ^ 1);                 // tile t + 1: its buffer was last read before the previous tile's barrier
    if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's K runs of tile t + 1 have landed in buffer buf ^ 1
    if (!(H2_ABL & 1)) __syncthreads();
    if (!(H2_ABL & 4)) stage_load((t + 2 < ntiles) ? t + 2 : ntiles - 1);
    if (!(H2_ABL & 4)) dma_k((t + 2 < ntiles) ? t + 2 : ntiles - 1, buf);      // buffer buf is free: K(t) sits in kop since the last tile, V(t) was read above
    stage_fn(std::integral_constant<int, 2>{}, first_tag, std::true_type{}, buf ^ 1);    // K(t + 1) follows K(t) through kop
    stage_fn(std::integral_constant<int, 3>{}, first_tag, std::true_type{});
  };

#if H2_DIAG
  const unsigned long long diag_t0 = __builtin_amdgcn_s_memtime(), diag_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  stage_load(0);
  dma_k(0, 0);
  stage_store(0);
  if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  stage_load(ntiles > 1 ? 1 : 0);
  dma_k(ntiles > 1 ? 1 : 0, 1);
  load_k(0);
#pragma unroll
  for (int n = 0; n < 4 * NQK; ++n) qk_mfma(0, 0, n);
#if H2_DIAG == 2
  // scores of (query tile 0 of wave 0, key tile 0) of the first workgroup of every pair: S[0][kt][r] = s(query qblk0 + i16, key 16 kt + 4 g + r)
  if (tile.x == 0 && wave == 0) {
    float* dg = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(const_cast<__bf16*>(wsq + 8 * piece_n)) + 128) + lane * 16;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) dg[kt * 4 + r] = S[0][kt][r];
    // ... and the operands the lane holds: Q sets 0, 1 of query tile 0 and K sets 0, 1 of key tile 0 (re-read from LDS)
    unsigned* du = reinterpret_cast<unsigned*>(dg - lane * 16 + 64 * 16) + lane * 16;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      du[w] = qop[0][0][w]; du[4 + w] = qop[0][1][w];
      du[8 + w] = reinterpret_cast<const unsigned*>(smem[0] + kaddr[0])[w]; du[12 + w] = reinterpret_cast<const unsigned*>(smem[0] + kaddr[1])[w];
    }
  }
#endif
  tile_fn(std::true_type{}, 0);
  for (int t = 1; t < ntiles; ++t) tile_fn(std::false_type{}, t);
#pragma unroll
  for (int n = 0; n < NPV; ++n) pv_mfma(NQ - 1, 1, n);

#if H2_DIAG
  if (tid == 0) {
    unsigned long long* dg = reinterpret_cast<unsigned long long*>(reinterpret_cast<unsigned char*>(const_cast<__bf16*>(wsq + 8 * piece_n)) + 64) + 2 * tile.x;
    dg[0] = __builtin_amdgcn_s_memtime() - diag_t0;
    dg[1] = __builtin_amdgcn_s_memrealtime() - diag_r0;
  }
#endif
  float* obase = out + ((size_t)b * C + (size_t)head * D) * L;
  const float* vinv = reinterpret_cast<const float*>(wsq + 8 * piece_n);      // 2^-s per channel of this head
#pragma unroll
  for (int qt = 0; qt < NQ; ++qt) {
    float lt = l_run[qt];
    lt += __shfl_xor(lt, 16, 64);
    lt += __shfl_xor(lt, 32, 64);
    const bool bad = !(lt < OVERFLOW_LIMIT);            // NaN / inf inputs: hand this query block to the fp32 kernel's check pass
    const float inv = bad ? __builtin_nanf("") : 1.0f / lt;
    const int q = qblk0 + qt * 16 + i16;
    if (lse2 != nullptr && g == 0)
      lse2[((size_t)b * gridDim.y + head) * L + q] = bad ? __builtin_nanf("") : __builtin_amdgcn_logf(lt) - negm4[qt][0];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        obase[(size_t)(mt * 16 + 4 * g + r) * L + q] = (O[mt][qt][r] * inv) * vinv[mt * 16 + 4 * g + r];
  }
}

}  // namespace

namespace hdiff {
// V of qkv as two fp16 pieces of V 2^s (s per channel row) + the D factors 2^-s, into the V region of the pre-split workspace
void launch_v_split_h2(const float* qkv, void* ws, int B, int C, int heads, int L, hipStream_t stream) {
  if (C / heads == 16) hipLaunchKernelGGL((v_split_h2_kernel<16>), dim3(C, B), dim3(THREADS), 0, stream, qkv, (__bf16*)ws, C, L, 1.0f);
  else hipLaunchKernelGGL((v_split_h2_kernel<32>), dim3(C, B), dim3(THREADS), 0, stream, qkv, (__bf16*)ws, C, L, 1.0f);
}

// Bytes the d_head 16 kernel keeps behind the (sample, head) pairs of the workspace: the Q / K row maxima
int64_t mha_fwd_h2_tail_bytes(int B, int C) { return ((int64_t)B * 2 * C * 4 + 255) / 256 * 256; }

// Q and K as fp16 score operands (pieces 0, 1 and 2 .. 5 of every pair; d_head 16 / 32), the row maxima in the workspace's tail
void launch_qk_split_h2(const float* qkv, void* ws, int B, int C, int heads, int L, float qscale, hipStream_t stream) {
  float* rowmax = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(ws) + mha_fwd_x3p_workspace(B, C, heads, L) - mha_fwd_h2_tail_bytes(B, C));
  hipLaunchKernelGGL(qk_rowmax_kernel, dim3(2 * C, B), dim3(THREADS), 0, stream, qkv, rowmax, C, L);
  dim3 grid(cdiv(L, 256), 2 * heads, B);
  if (C / heads == 16) hipLaunchKernelGGL((qk_split_h2_kernel<16>), grid, dim3(THREADS), 0, stream, qkv, rowmax, (__bf16*)ws, C, L, qscale, 1.0f);
  else hipLaunchKernelGGL((qk_split_h2_kernel<32>), grid, dim3(THREADS), 0, stream, qkv, rowmax, (__bf16*)ws, C, L, qscale, 1.0f);
}

// The d_head 16 forward on fp16 pairs (scores: four balanced products; P.V: three) on its own operand layout in the workspace.
// Returns false when the shape is not covered or the workspace is missing (the caller then runs the bf16-triple kernels).
bool launch_mha_fwd_h2(const float* qkv, float* o, float* lse2, int B, int C, int heads, int L, float qscale, void* ws,
                       int64_t ws_bytes, hipStream_t stream) {
  const int64_t need = mha_fwd_x3p_workspace(B, C, heads, L);
  if (need == 0 || ws == nullptr || ws_bytes < need) return false;
  const int D = C / heads;
  if (D != 16) return false;
  launch_qk_split_h2(qkv, ws, B, C, heads, L, qscale, stream);
  launch_v_split_h2(qkv, ws, B, C, heads, L, stream);
  hipLaunchKernelGGL((mha_flash_fwd_h2_kernel<16, 4>), dim3(L / 256, heads, B), dim3(THREADS), 0, stream, (const __bf16*)ws, o, lse2,
                     C, L, 1.0f);
  return true;
}

}  // namespace hdiff
