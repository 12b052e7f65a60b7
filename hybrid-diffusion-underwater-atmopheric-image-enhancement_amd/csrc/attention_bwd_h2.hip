// Backward of the flash self-attention core at d_head 16 and 32 in the split-operand mode.  Round 3 built it on bf16 triples
// throughout (attention_bwd_x3.hip, retired in round 5: git show 6a5a4b1 has it); round 4 moved everything that touches
// P = exp2(S - lse) or the pair (dO, V) to fp16 PAIRS; round 5 moves the rest: EVERY operand is an fp16 pair now.
// Same contract, layouts and dQ slab protocol (reference: autograd through nn.MultiheadAttention, ModelCondition.py:189,
// 204-208, TrainCondition.py:60): bitwise reproducible.
//
// A pair is x = x0 + x1 with x0 = fp16(x) and x1 = fp16 of the exact residual: 2^-22 |x| at worst as long as x1 is a NORMAL fp16
// number.  Two ways to keep it one:
//   (i)  "balance per product term" (attention_h2.hip): store x1 2^8 and give its partner in the product 2^-8 -- stored (K, V:
//        stationary operands) or made by one v_pk_mul_f16 per operand word.  For operands whose magnitude is whatever the model
//        produces (Q, K with the forward's balance a; P', dS').
//   (ii) scale the whole tensor by a power of two that puts its maximum in [2^13, 2^14): then the UNshifted x1 resolves 2^-24
//        absolute = 2^-38 of the maximum.  For operands this file makes its own copy of anyway (below).
//   S    = Q K^T - lse2 + 14     q = Q qscale 2^-a, k = K 2^a (a per (sample, head)); four terms
//                                q0 k0 + (q1 2^8)(k0 2^-8) | (q0 2^-8)(k1 2^8) + q1 k1: two MFMAs per tile at d 16 (bf16 triples: three);
//                                d 32: three MFMAs, the 2^-24 term q1 k1 dropped
//   dP'  = dO' V'^T - delta'     V' = V 2^sv per (sample, head); dO'_q = dO_q 2^(so + t_q) per QUERY position: every row's maximum in
//                                [2^13, 2^14) (t_q = exponent of the head's loudest position minus that of position q, 0 .. 24).  One loud
//                                position no longer decides the scale of everybody's dS' row (tests: "one pixel of dO 1e4 x the rest").
//                                The same four-term form; |dP' - delta'| <= 2^(29 + log2 d)
//   P'   = exp2(S) = P 2^14      two pieces (p0, p1): v_cvt_pk_f16_f32 + v_fma_mixlo / mixhi_f16
//   dS'  = (P' o dP') 2^-(28 + log2 d)   <= 2^15: inside fp16 by construction; two pieces (s0, s1 2^8), the factor applied inside the split's own
//                                conversions (split2_scaled; the SHIFTED residual is what keeps a flat softmax over 65 536 keys precise)
//   dV'^T += (dO 2^so)^T P'      o0 p0 + o0 p1 + o1 p0: dO with the HEAD's scale alone (a second staged copy, form (ii): unshifted pieces) --
//                                P' needs no per-query factor and the three products share one accumulator
//   dK'^T += (Q 2^sq c_q)^T dS'  q0 s0 + (q0 2^-8)(s1 2^8) + q1 s0: a copy of the Q rows that gives the per-query factor c_q = 2^-t_q
//                                back, form (ii); q0 2^-8 by v_pk_mul_f16 after the transposed read
//   dQ'^T += k^T dS'^T           a: k0 s0, b: (k1 2^8) s0 + k0 (s1 2^8); a + b 2^-8 and the factor c_q when the tile leaves
//                                contracted ACROSS the eight-wave workgroup: every wave's dS images stay in LDS until the tile's barrier, then each wave
//                                takes one 16 x 16 output tile over the 128 keys of four waves -- two partial tiles to sum instead of eight (Geo::DQX)
// Per 16x16 (query, key) tile at d 16: 8.5 MFMAs and ~33 vector instructions per lane in the ISA (exp2 4, P split 6, dS product + split 16,
// operand shifts 3, the rest addressing and the dQ output path; round 4: 12.5 / ~36; round 3: 15 / ~52).
// Error class (tests/test_gpu_backward.py, profiles/r05_attention_bwd_error_ratio.txt): rms against float64 <= the fp32-input
// kernel's on every input family tried; the worst ELEMENT up to 2.3x (3x for one (sample, head) pair) -- where one score dominates
// a row the pair's 2^-22 operand rounding is the whole error.
#include <stdlib.h>

#include <atomic>
#include <type_traits>

#include "common.h"

using namespace hdiff;

namespace {

#ifndef H2B_ABL
#define H2B_ABL 0            // dev: timing ablations (bit mask), results are wrong with any bit set
#endif
constexpr int THREADS = 256;           // of the preparation kernels
#ifndef H2B_WAVES
#define H2B_WAVES 8
#endif
#ifndef H2B_WANT
#define H2B_WANT 1024
#endif
#ifndef H2B_DMA
#define H2B_DMA 1           // dev: the tile's piece rows global -> LDS by LDS-DMA (eight-wave form; 0: through registers; A/B)
#endif
#ifndef H2B_DQX
#define H2B_DQX 3           // dev: bit 0 = dQ across the workgroup at d 16, bit 1 = at d 32 (A/B)
#endif
// Waves of a main-kernel workgroup, 32 keys each.  EIGHT: one workgroup per CU instead of two of four waves -- every query tile is
// staged once per 256 keys instead of once per 128, and the workgroup's dQ partial tile (summed over its waves in LDS) goes to the
// slab once per 256 keys: half the L2 float adds (timing ablation on the four-wave build: the adds were 13 % of the launch).
constexpr int MW = H2B_WAVES;
constexpr int MTHREADS = 64 * MW;
constexpr int KB = 32 * MW;            // keys per workgroup block
constexpr int SROW = 72;               // bytes per key row of the dS image [key][32 queries] (64 + 8: conflict-free ds_write_b64)
constexpr int SPART = 32 * SROW;       // 2304 per piece
constexpr int SCRB = 2 * SPART;        // 4608 per wave (two pieces of dS)

// Geometry by head width.  d 16: tiles of 64 queries (two subtiles of 32 = one contraction of the q-summed products), rows of
// 32 bytes -- with ds_read_b128's real lane groups ({0-3, 12-15, 20-27}, ...) plain 32-byte rows are conflict-free for the row
// reads, the transposed reads and the staging stores alike (48-byte rows: a third of all LDS cycles were bank conflicts,
// SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.35).  d 32: tiles of 32 queries, rows of 64 bytes, two 16-row M tiles in the
// products whose output rows are d.
template <int D>
struct Geo {
  static constexpr int TQ = (D == 16) ? 64 : 32;      // queries per staged tile
  static constexpr int NSUB = TQ / 32;                // subtiles of 32 queries
  static constexpr int MT = D / 16;                   // 16-row M tiles of dV^T, dK^T, dQ^T
  static constexpr int GROW = 2 * D;                  // bytes per query row of a piece tensor in memory
  static constexpr int RROW = GROW;                   // ... and in LDS: no padding (d 32: rows padded to 96 bytes, conflict-free for
                                                      // row and transposed reads alike, measured no faster: 16.39 vs 16.35 ms)
  static constexpr int CPR = GROW / 16;               // 16-byte chunks per row
  static constexpr int RPART = TQ * RROW;
  // rows of (q0, q1 2^8); of Q 2^sq c_q (x0, x1: form (ii), unshifted); of dO scaled per query (o0, o1: form (ii)); of dO scaled per head likewise
  static constexpr int QA_OFF = 0, QE_OFF = 2 * RPART, OA_OFF = 4 * RPART, OH_OFF = 6 * RPART;
  static constexpr int SL_OFF = 8 * RPART, SD_OFF = SL_OFF + TQ * 4, SC_OFF = SD_OFF + TQ * 4, BUFB = SC_OFF + TQ * 4;
  static constexpr int DQS = TQ + 4;                  // row stride (floats) of a wave's dQ partial tile [D][DQS], aliased on its scratch
  static_assert(TQ * GROW == 2048, "staging geometry: 128 chunks per piece");
  static_assert(D * DQS * 4 <= SCRB, "dQ partial tile must fit in the wave's scratch");
  // dQ across the workgroup (eight waves): every wave's dS images of the tile stay in LDS until the tile's barrier, then wave w contracts
  // ONE 16 x 16 output tile (w & 3: d 16 -- 16-query group of the 64; d 32 -- (row half, 16-query group)) over the 128 keys of four waves
  // (half w >> 2) -- two partial tiles to sum instead of eight (timing ablations: the eight-way sum was ~40 of a tile's ~340 instructions
  // per thread).  d 16: -6 %; d 32: -2.6 % (the four waves' K^T operands cost it six spilled registers outside the tile loop).
  static constexpr bool DQX = (MW == 8) && (D == 16 ? (H2B_DQX & 1) != 0 : (H2B_DQX & 2) != 0);
  static constexpr int IMG_BYTES = MW * (DQX ? NSUB : 1) * SCRB;
  static constexpr int PART_BYTES = D * DQS * 4;                    // one partial tile [D][DQS] (DQX: two of them behind the images)
  static constexpr int LDS_BYTES = 2 * BUFB + IMG_BYTES + (DQX ? 2 * PART_BYTES : 0);      // dynamic (d 16: 119 KB, d 32: 70 KB)
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

// gfx950 transposing LDS read: per 16-lane group a block of 4 rows x 16 columns of 16-bit elements; lane 4q + p of the
// group supplies the address of row q, columns 4p .. 4p + 3; lane i receives column i of the 4 rows (row q in element q)
__device__ __forceinline__ u32x2 lds_read_tr16(const unsigned char* p) {
  return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p));
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for the wave's outstanding GLOBAL stores
// (release fence: s_waitcnt vmcnt(0)); with one slab store per tile in flight that wait exposed the store's latency at
// every barrier -- 23 ms of a 158 ms launch (timing ablation).  Nothing here hands global data to another wave.
__device__ __forceinline__ void lds_barrier() {
  if (H2B_ABL & 256) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

typedef __attribute__((address_space(3))) unsigned char lds_byte;
// One 1 KiB run global -> LDS without staging registers (global_load_lds_dwordx4: lane i's 16 bytes at src + voff land at lds_dst + 16 i).
// M0 is written in the statement that uses it and restored; the compiler does not count this load: the kernel waits with its own
// s_waitcnt vmcnt(0) in front of the barrier that publishes the tile (attention_h2.hip does the same for its K tiles).
__device__ __forceinline__ void dma_1k(const unsigned char* src, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(src), "s"(lds_dst)
               : "memory");
}
__device__ __forceinline__ f32x4 mfma_f16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// (a, b) -> two packed fp16 pairs, a = h0.lo + h1.lo up to 2^-23 |a| (or 2^-25 absolute); see attention_h2.hip.  `one` is
// 1.0f in a register the compiler cannot see through (the residual must stay an fma: v_fma_mixlo / mixhi_f16).
__device__ __forceinline__ void split2(float a, float b, float one, unsigned& h0, unsigned& h1) {
  const f16x2 p = {(_Float16)a, (_Float16)b};
  unsigned u = __builtin_bit_cast(unsigned, p);
  asm("" : "+v"(u));
  const f16x2 q = __builtin_bit_cast(f16x2, u);
  const f16x2 r = {(_Float16)__builtin_fmaf(a, one, -(float)q[0]), (_Float16)__builtin_fmaf(b, one, -(float)q[1])};
  h0 = u;
  h1 = __builtin_bit_cast(unsigned, r);
}
// (a c, b c) -> two packed fp16 pairs with the SECOND piece times 2^8: h0 = fp16(a c) by v_fma_mixlo / mixhi_f16 (fma(a, c, 0)), r = the
// EXACT residual fma(a, c, -h0) in fp32 (v_fma_mix_f32 reads h0 as fp16), h1 = fp16(256 r).  Three instructions per value (an unshifted
// residual would take two: but dS' is typically 2^-7 where the softmax is flat over 65 536 keys -- the bound 2^15 belongs to a row that
// is one-hot AND has |dP - delta| at its ceiling -- and an unshifted residual of that is a subnormal: dQ rms 1.45 x the fp32 kernel's at
// L = 65 536 with N(0, 1) inputs).  In asm: from C the compiler makes the first piece a v_mul_f32 + a conversion.  Two value pairs per
// statement so that no v_fma_mixhi follows its v_fma_mixlo directly.  `up` = 256.0f in a register.
__device__ __forceinline__ void split2_scaled(float a0, float a1, float b0, float b1, float c, float up, unsigned& ha0, unsigned& ha1,
                                              unsigned& hb0, unsigned& hb1) {
  float r0, r1, r2, r3;
  asm("v_fma_mixlo_f16 %0, %8, %12, 0\n\t"
      "v_fma_mixlo_f16 %2, %10, %12, 0\n\t"
      "v_fma_mixhi_f16 %0, %9, %12, 0\n\t"
      "v_fma_mixhi_f16 %2, %11, %12, 0\n\t"
      "s_nop 0\n\t"
      "v_fma_mix_f32 %4, %8, %12, -%0 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mix_f32 %6, %10, %12, -%2 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mix_f32 %5, %9, %12, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
      "v_fma_mix_f32 %7, %11, %12, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixlo_f16 %1, %4, %13, 0\n\t"
      "v_fma_mixlo_f16 %3, %6, %13, 0\n\t"
      "v_fma_mixhi_f16 %1, %5, %13, 0\n\t"
      "v_fma_mixhi_f16 %3, %7, %13, 0\n\t"
      "s_nop 0"
      : "=&v"(ha0), "=&v"(ha1), "=&v"(hb0), "=&v"(hb1), "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
      : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(c), "v"(up));
}
constexpr int PAIR_SHIFT = 8;                 // second pieces are stored times 2^8
constexpr unsigned PAIR_DOWN2 = 0x1c001c00u;  // (2^-8, 2^-8) as packed fp16
// exponent e of a tensor's maximum |x| (given as the bits of the fp32 value), clamped: all-zero / denormal / huge tensors get a fixed
// scale (inf and NaN stay what they are)
__device__ __forceinline__ int max_exp(unsigned amax_bits) {
  const int e = (int)((amax_bits >> 23) & 0xffu) - 127;
  return e < -100 ? -100 : (e > 100 ? 100 : e);
}
// V' = V 2^sv with the (sample, head)'s maximum in [2^13, 2^14).  dO is scaled PER QUERY: dO'_q = dO_q 2^(so + t_q), so the head's power
// of two (maximum of the loudest position in [2^13, 2^14)) and t_q = (exponent of the head's maximum) - (exponent of position q's
// maximum), clamped to 0 .. 24 -- every position's row sits at the top of fp16, and so does its row of dS'.  Where the contraction
// runs over queries the factor c_q = 2^-t_q is given back by the OTHER operand: dV^T takes a copy of dO with the head's scale alone,
// dK^T a copy of the Q rows times c_q (both staged beside the tiles of the score products); dQ_q is multiplied by c_q when it leaves.
__device__ __forceinline__ int scale_exp_o(unsigned amax_bits) { return 13 - max_exp(amax_bits); }
__device__ __forceinline__ int scale_exp_v(unsigned amax_bits) { return 13 - max_exp(amax_bits); }
__device__ __forceinline__ int scale_exp_q(unsigned amax_bits) { return 13 - max_exp(amax_bits); }      // of the Q rows that enter dK^T
constexpr int P_UP = 14;       // P' = exp2(S - lse2 + 14) <= 2^14, as in round 4 (the chain's initial value 14 - lse2 stays SMALL: with -17 - lse2
                               // instead -- tried -- every accumulation of S rounds at the magnitude of ~33 and the exponent loses three bits)
// dS' = P' (dP' - delta') 2^-DS_DOWN <= 2^15: |dP'| <= d 2^14 2^14, |delta'| = |sum_k P dP'| likewise, P' <= 2^14
template <int D> constexpr int DS_DOWN = (D == 16) ? 32 : 33;
// the score balance of this (sample, head): k = K 2^a, q = Q qscale 2^-a with the two maxima in the same binade (attention_h2.hip)
__device__ __forceinline__ int balance_exp(unsigned qmax_bits, unsigned kmax_bits, float qscale) {
  const float mq = __builtin_bit_cast(float, qmax_bits) * qscale;
  const int eq = (int)((__builtin_bit_cast(unsigned, mq) >> 23) & 0xffu), ek = (int)((kmax_bits >> 23) & 0xffu);
  int a = (eq == 0 || ek == 0 || eq == 255 || ek == 255) ? 0 : (eq - ek) / 2;
  return a < -60 ? -60 : (a > 60 ? 60 : a);
}

// Piece slots of one (sample, head), each L * D fp16: rows of q (q0, q1 2^8); rows of Q 2^sq c_q (x0, x1); rows of k (k0, k0 2^-8,
// k1 2^8, k1); k transposed [D][L] (k0, k1 2^8); rows of V' (v0, v1: form (ii), two slots used of the four reserved); rows of dO scaled per query (o0, o1: unshifted); rows of
// dO scaled per head (x0, x1); the L factors c_q (fp32) in the last slot
enum { S_Q = 0, S_QE = 2, S_K = 4, S_KT = 8, S_V = 10, S_O = 12, S_OH = 14, S_C = 16, S_COUNT = 17 };
enum { M_V = 0, M_O = 1, M_Q = 2, M_K = 3, M_COUNT = 4 };      // maxima per (sample, head)

// ---------------------------------------------------------------------------------------------------------------------
// max |V|, |dO|, |Q|, |K| per (sample, head) as fp32 bit patterns (non-negative floats order like unsigned integers):
// absmax[(b * heads + head) * 4 + {V, dO, Q, K}], zeroed by the launcher.  grid (L / 4096, 4 * heads, B).
// ---------------------------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(THREADS) void mha_bwd_absmax_kernel(const float* __restrict__ qkv, const float* __restrict__ d_o,
                                                                 unsigned* __restrict__ absmax, int C, int L) {
  const int heads = C / D;
  const int which = blockIdx.y / heads, head = blockIdx.y - which * heads, b = blockIdx.z;
  const int third = which == M_V ? 2 : (which == M_Q ? 0 : 1);
  const float* src = (which == M_O) ? d_o + ((size_t)b * C + (size_t)head * D) * L
                                    : qkv + ((size_t)b * 3 * C + (size_t)third * C + (size_t)head * D) * L;
  const int l0 = blockIdx.x * 4096;
  float m = 0.f;
  for (int d = 0; d < D; ++d)
    for (int i = threadIdx.x * 4; i < 4096 && l0 + i < L; i += THREADS * 4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(src + (size_t)d * L + l0 + i);
      m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(absmax + ((size_t)b * heads + head) * M_COUNT + which, __builtin_bit_cast(unsigned, m));
}

// ---------------------------------------------------------------------------------------------------------------------
// fp32 qkv [B][3C][L] and dO [B][C][L] -> the piece tensors above.  grid (L / 256, 4 * heads, B): blockIdx.y / heads selects
// Q, K, V or dO; thread = one position, all D channels (K^T: thread = two neighbouring positions of each channel).
// The fp32 scaled value and the packed first pieces are made opaque to the compiler: left alone it rounds x0 twice (once from
// the fp32 product for the stored piece, once from the exact product for the residual: attention_h2.hip).
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void pair4(float xa, float xb, float up, unsigned& x0, unsigned& x0s, unsigned& x1s, unsigned& x1) {
  asm("" : "+v"(xa), "+v"(xb));
  unsigned u0 = __builtin_bit_cast(unsigned, f16x2{(_Float16)xa, (_Float16)xb});
  asm("" : "+v"(u0));
  const f16x2 h = __builtin_bit_cast(f16x2, u0);
  const float ra = xa - (float)h[0], rb = xb - (float)h[1];            // exact
  const f16x2 dn = {(_Float16)(1.0f / (1 << PAIR_SHIFT)), (_Float16)(1.0f / (1 << PAIR_SHIFT))};
  x0 = u0;
  x0s = __builtin_bit_cast(unsigned, h * dn);
  x1s = __builtin_bit_cast(unsigned, f16x2{(_Float16)(ra * up), (_Float16)(rb * up)});
  x1 = __builtin_bit_cast(unsigned, f16x2{(_Float16)ra, (_Float16)rb});
}

template <int D>
__global__ __launch_bounds__(THREADS) void mha_bwd_split_h2_kernel(const float* __restrict__ qkv, const float* __restrict__ d_o,
                                                                   __bf16* __restrict__ ws, const unsigned* __restrict__ absmax,
                                                                   const float* __restrict__ lse2, const float* __restrict__ delta,
                                                                   int C, int L, float qscale, float one) {
  const int heads = C / D;
  const int which = blockIdx.y / heads, head = blockIdx.y - which * heads, b = blockIdx.z;     // 0: Q, 1: K, 2: V, 3: dO
  const float* src = (which < 3) ? qkv + ((size_t)b * 3 * C + (size_t)which * C + (size_t)head * D) * L
                                 : d_o + ((size_t)b * C + (size_t)head * D) * L;
  const size_t piece = (size_t)L * D;
  __bf16* base = ws + ((size_t)b * heads + head) * S_COUNT * piece;
  const unsigned* mx = absmax + ((size_t)b * heads + head) * M_COUNT;
  const int a = balance_exp(mx[M_Q], mx[M_K], qscale);
  const float sc = which == 0 ? qscale * __builtin_ldexpf(1.0f, -a) : which == 1 ? __builtin_ldexpf(1.0f, a)
                 : which == 2 ? __builtin_ldexpf(1.0f, scale_exp_v(mx[M_V])) : __builtin_ldexpf(1.0f, scale_exp_o(mx[M_O]));
  const float up = (float)(1 << PAIR_SHIFT) * one;
  const int l = blockIdx.x * THREADS + threadIdx.x;
  if (l < L) {
    // t_q of this position (the Q and the dO blocks): exponent of the head's dO maximum minus that of the position's own
    int tq = 0;
    if (which == 0 || which == 3) {
      const float* dsrc = d_o + ((size_t)b * C + (size_t)head * D) * L + l;
      float rm = 0.f;
#pragma unroll
      for (int d = 0; d < D; ++d) rm = fmaxf(rm, fabsf(dsrc[(size_t)d * L]));
      tq = max_exp(mx[M_O]) - ((int)((__builtin_bit_cast(unsigned, rm) >> 23) & 0xffu) - 127);
      tq = tq < 0 ? 0 : (tq > 24 ? 24 : tq);
      if (which == 3) {
        float* cq = reinterpret_cast<float*>(base + (size_t)S_C * piece);
        cq[l] = __builtin_ldexpf(1.0f, -tq);      // c_q
        // (pipelined kernel) the two per-query starting values of the d-contracted chains, per 32-query tile [tile][2][32] behind the L factors:
        // 14 - lse2 (S') and -delta'_q = -delta_q 2^so 2^sv 2^t_q (dP'), the powers of two one after the other (each a finite float)
        const size_t bh = (size_t)b * heads + head;
        float* nld = cq + L + (size_t)(l >> 5) * 64 + (l & 31);
        nld[0] = (float)P_UP - lse2[bh * L + l];
        nld[32] = ((-delta[bh * L + l] * __builtin_ldexpf(1.0f, scale_exp_o(mx[M_O]))) * __builtin_ldexpf(1.0f, scale_exp_v(mx[M_V]))) * __builtin_ldexpf(1.0f, tq);
      }
    }
    // the two powers of two one after the other: 2^(so + t_q) itself can leave fp32 (a head whose dO is all zero or below 2^-91 has so = 113,
    // a silent row t_q = 24: 0 x inf = NaN), each factor cannot, and after the first one every row's maximum is at most 2^14 2^-t_q
    const float tqf = which == 3 ? __builtin_ldexpf(1.0f, tq) : 1.0f;
    unsigned h[4][D / 2];          // x0, x0 2^-8, x1 2^8, x1 as packed fp16 pairs (channels 2 j, 2 j + 1)
#pragma unroll
    for (int j = 0; j < D / 2; ++j)
      pair4((src[(size_t)(2 * j) * L + l] * sc) * tqf, (src[(size_t)(2 * j + 1) * L + l] * sc) * tqf, up, h[0][j], h[1][j], h[2][j], h[3][j]);
    // q: (x0, x1 2^8); k: all four; V', dO' (maxima in [2^13, 2^14): form (ii)): (x0, x1)
    const int slot0 = which == 0 ? S_Q : which == 1 ? S_K : which == 2 ? S_V : S_O;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      if (which == 0 && (p == 1 || p == 3)) continue;
      if (which >= 2 && (p == 1 || p == 2)) continue;
      const int slot = slot0 + (which == 1 ? p : (p == 0 ? 0 : 1));
      u32x4* o = reinterpret_cast<u32x4*>(base + (size_t)slot * piece + (size_t)l * D);
#pragma unroll
      for (int j = 0; j < D / 8; ++j) o[j] = u32x4{h[p][4 * j], h[p][4 * j + 1], h[p][4 * j + 2], h[p][4 * j + 3]};
    }
    if (which == 0 || which == 3) {
      // the A operands of the products that sum over queries: rows of Q 2^sq c_q (dK^T) / of dO 2^so (dV^T), the head's maximum in
      // [2^13, 2^14) -- with that scale the SECOND piece needs no shift (x1 unshifted resolves 2^-24 absolute = 2^-38 of the maximum): (x0, x1)
      const float s2 = which == 0 ? __builtin_ldexpf(1.0f, scale_exp_q(mx[M_Q]) - tq) : sc;
#pragma unroll
      for (int j = 0; j < D / 2; ++j)
        pair4(src[(size_t)(2 * j) * L + l] * s2, src[(size_t)(2 * j + 1) * L + l] * s2, up, h[0][j], h[1][j], h[2][j], h[3][j]);
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        u32x4* o = reinterpret_cast<u32x4*>(base + (size_t)((which == 0 ? S_QE : S_OH) + p) * piece + (size_t)l * D);
#pragma unroll
        for (int j = 0; j < D / 8; ++j) o[j] = u32x4{h[3 * p][4 * j], h[3 * p][4 * j + 1], h[3 * p][4 * j + 2], h[3 * p][4 * j + 3]};
      }
    }
  }
  if (which == 1) {          // K transposed [D][L]: (k0, k1 2^8)
    __bf16* dst = base + (size_t)S_KT * piece;
    const int l2 = (int)blockIdx.x * (THREADS / 2) + (int)threadIdx.x;       // the block's 256 positions = 128 pairs: half the threads
    if ((int)threadIdx.x < THREADS / 2 && 2 * l2 < L) {
#pragma unroll 4
      for (int d = 0; d < D; ++d) {
        const f32x2 v = *reinterpret_cast<const f32x2*>(src + (size_t)d * L + 2 * l2);
        unsigned x0, x0s, x1s, x1;
        pair4(v[0] * sc, v[1] * sc, up, x0, x0s, x1s, x1);
        unsigned* o = reinterpret_cast<unsigned*>(dst + (size_t)d * L) + l2;
        o[0] = x0;
        o[piece / 2] = x1s;
      }
    }
  }
}

struct BwdH2Args {
  const __bf16* ws;           // piece tensors
  const float* lse2;
  const float* delta;
  float* dqkv;
  float* dq_part;             // partial dQ slabs, as in attention_bwd.hip
  size_t split_stride, batch_stride;
  int C, L, kb_per_split;
  float inv_sqrt_d, qscale;
  const unsigned* absmax;     // max |V|, |dO|, |Q|, |K| per (sample, head): mha_bwd_absmax_kernel
  float one;                  // 1.0f, opaque to the compiler (split2)
};

template <int D>
__global__ __launch_bounds__(MTHREADS) void mha_bwd_h2_kernel(const BwdH2Args a) {
  using G = Geo<D>;
  constexpr int TQ = G::TQ, NSUB = G::NSUB, MT = G::MT, RROW = G::RROW, GROW = G::GROW, CPR = G::CPR, RPART = G::RPART, BUFB = G::BUFB, DQS = G::DQS;
  constexpr bool DQX = G::DQX;
  constexpr int QA_OFF = G::QA_OFF, QE_OFF = G::QE_OFF, OA_OFF = G::OA_OFF, OH_OFF = G::OH_OFF, SL_OFF = G::SL_OFF, SD_OFF = G::SD_OFF, SC_OFF = G::SC_OFF;
  // d-contracted products (S, dP'), four terms x0 y0, (x1 2^8)(y0 2^-8), (x0 2^-8)(y1 2^8), x1 y1:
  //   d 16: two terms share an MFMA's 32 slots -- row operand A0 = (x0 | x1 2^8) as staged, A1 = A0 2^-8; stationary B0 = (y0 | y0 2^-8),
  //         B1 = (y1 2^8 | y1): 2 MFMAs;   d 32: one term per MFMA and the 2^-22 term x1 y1 dropped (the d_head 32 forward does the same:
  //         its error against float64 does not move) -- A = x0, x1 2^8, x0 2^-8; B = y0, y0 2^-8, y1 2^8: 3 MFMAs
  constexpr int NA = (D == 16) ? 1 : 2;        // row operands READ per 16 rows
  constexpr int NT = (D == 16) ? 2 : 3;        // MFMAs of one d-contracted 16x16 tile = stationary operand sets per 16 keys; operands NA .. NT - 1 are
                                               // the first NT - NA read ones times 2^-8
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // 2 * BUFB + MW * SCRB

  const int C = a.C, L = a.L;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  const TileId tile = xcd_tile();
  const int head = tile.head, b = tile.b, heads = gridDim.y, split = tile.x;
  const size_t piece_n = (size_t)L * D;
  const __bf16* wsh = a.ws + ((size_t)b * heads + head) * S_COUNT * piece_n;
  const float* lbase = a.lse2 + ((size_t)b * heads + head) * L;
  const float* dbase = a.delta + ((size_t)b * heads + head) * L;
  const float* cbase = reinterpret_cast<const float*>(wsh + (size_t)S_C * piece_n);      // c_q = 2^-t_q
  float* part = a.dq_part + (size_t)split * a.split_stride + (size_t)b * a.batch_stride + (size_t)head * D * L;
  float* kout = a.dqkv + ((size_t)b * 3 * C + (size_t)C + (size_t)head * D) * L;
  float* vout = kout + (size_t)C * L;
  const int ntiles = L / TQ;
  const int nkb_total = L / KB;
  const int kb_begin = split * a.kb_per_split;
  const int kb_end = (kb_begin + a.kb_per_split < nkb_total) ? kb_begin + a.kb_per_split : nkb_total;
  // the powers of two of this (sample, head): dO' = dO 2^so, V' = V 2^sv, k = K 2^bal, q = Q qscale 2^-bal
  const unsigned* mx = a.absmax + ((size_t)b * heads + head) * M_COUNT;
  const int so = scale_exp_o(mx[M_O]), sv = scale_exp_v(mx[M_V]), sq = scale_exp_q(mx[M_Q]), bal = balance_exp(mx[M_Q], mx[M_K], a.qscale);
  const float dscale_o = __builtin_ldexpf(1.0f, so), dscale_v = __builtin_ldexpf(1.0f, sv);      // delta' = delta 2^so 2^sv (two factors: each a finite float)
  const float one = a.one;
  const float dnb = __builtin_ldexpf(1.0f, -PAIR_SHIFT);     // a + b 2^-8
  const float dsdn = __builtin_ldexpf(one, -DS_DOWN<D>);        // dS' = P' (dP' - delta') 2^-DS_DOWN
  const float up8 = __builtin_ldexpf(one, PAIR_SHIFT);

  // contraction slots of this lane in the d-contracted products.  d 16: 8 consecutive d of one of the MFMA's two terms;
  // d 32: 8 consecutive d of the one term
  const int doff = (D == 16) ? 8 * (g & 1) : 8 * g;
  const bool hi = (D == 16) && (g >> 1);

  // ---- staging of one query tile: 1024 chunks of 16 bytes (two row pieces each of Q, of q c_q, of dO per query and of dO per head), two per thread;
  // 14 - lse2, -delta' and c_q of the tile's queries by the first TQ threads
  // the 1024 chunks: blocks {Q, q c_q, dO per query, dO per head} x 2 pieces x 128 chunks; thread tid takes chunk tid + MTHREADS i
  // (BDMA) the eight 2 KiB piece regions of a tile are contiguous in the workspace and in LDS (no row padding): wave w copies region w as two
  // 1 KiB LDS-DMA runs; only the three per-query vectors go through registers
  constexpr bool BDMA = DQX && (H2B_DMA != 0);
  static_assert(!BDMA || (RROW == GROW && RPART == 2048 && MW == 8), "LDS-DMA geometry");
  constexpr int NST = BDMA ? 0 : 1024 / MTHREADS;
  static_assert(MTHREADS == 256 || MTHREADS == 512, "staging geometry");
  static_assert(S_O - S_Q == S_OH - S_QE && G::OA_OFF - G::QA_OFF == G::OH_OFF - G::QE_OFF, "chunk i + 1 of a thread = chunk i moved by a fixed amount");
  unsigned goff0;
  int lds_off0;
  {
    const int blk = tid >> 8, cv = tid & 255;             // (512 threads) block 0 / 1 for chunk 0, 2 / 3 for chunk 1; (256) block i for chunk i
    const int p = cv >> 7, rem = cv & 127;
    const int row = rem / CPR, ch = rem % CPR;
    goff0 = (unsigned)(((blk ? S_QE : S_Q) + p) * piece_n * 2) + row * GROW + ch * 16;
    lds_off0 = (blk ? QE_OFF : QA_OFF) + p * RPART + row * RROW + ch * 16;
  }
  // chunk i: (512 threads) + i (S_O - S_Q) slots / + i (OA_OFF - QA_OFF); (256 threads) slots S_Q, S_QE, S_O, S_OH in turn
  auto gslot = [&](int i) -> size_t { return (size_t)(MTHREADS == 512 ? i * (S_O - S_Q) : (i == 0 ? 0 : i == 1 ? S_QE - S_Q : i == 2 ? S_O - S_Q : S_OH - S_Q)) * (piece_n * 2); };
  auto lslot = [&](int i) { return MTHREADS == 512 ? i * (OA_OFF - QA_OFF) : (i == 0 ? 0 : i == 1 ? QE_OFF : i == 2 ? OA_OFF : OH_OFF); };
  const unsigned char* wsb = reinterpret_cast<const unsigned char*>(wsh);
  u32x4 stage[NST ? NST : 1];
  float stage_l = 0.f, stage_d = 0.f, stage_c = 1.f;
  const unsigned lds0 = (unsigned)(size_t)(lds_byte*)smem;
  auto dma_tile = [&](int t, int buf) {
    if (!BDMA) return;
    const int w_ = __builtin_amdgcn_readfirstlane(wave);                 // region w: slots S_Q, S_Q + 1, S_QE, S_QE + 1, S_O, S_O + 1, S_OH, S_OH + 1
    const int slot = (w_ < 4) ? w_ : w_ + (S_O - 4);
    const unsigned char* src = wsb + (size_t)slot * (piece_n * 2) + (size_t)t * (TQ * GROW);
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + buf * BUFB + w_ * RPART);
    dma_1k(src, lane * 16, dst);
    dma_1k(src + 1024, lane * 16, dst + 1024);
  };
  auto stage_load = [&](int t) {
#pragma unroll
    for (int i = 0; i < NST; ++i) stage[i] = *reinterpret_cast<const u32x4*>(wsb + gslot(i) + goff0 + (size_t)t * (TQ * GROW));
    if (tid < TQ) { stage_l = lbase[t * TQ + tid]; stage_d = dbase[t * TQ + tid]; stage_c = cbase[t * TQ + tid]; }      // consumed when stored
  };
  auto stage_store = [&](int buf) {
    unsigned char* tb = smem + buf * BUFB;
#pragma unroll
    for (int i = 0; i < NST; ++i) *reinterpret_cast<u32x4*>(tb + lslot(i) + lds_off0) = stage[i];
    if (tid < TQ) {
      *reinterpret_cast<float*>(tb + SL_OFF + tid * 4) = (float)P_UP - stage_l;
      *reinterpret_cast<float*>(tb + SD_OFF + tid * 4) = ((-stage_d * dscale_o) * dscale_v) / stage_c;      // delta'_q = delta_q 2^(so + t_q + sv): c_q is a power of two
      *reinterpret_cast<float*>(tb + SC_OFF + tid * 4) = stage_c;
    }
  };

  // operand addresses inside a tile buffer
  int a1addr[NA];       // row reads (Q and dO alike, + QA_OFF / OA_OFF): (d 16) the piece of this lane half / (d 32) piece j; row i16, 16 bytes at doff
#pragma unroll
  for (int j = 0; j < NA; ++j) a1addr[j] = ((D == 16) ? (hi ? 1 : 0) : j) * RPART + i16 * RROW + doff * 2;
  // transposed reads of the same tiles (A operands of the products that sum over queries): lane 4q + p of a 16-lane group
  // addresses row 4g + q (then 16 + 4g + q), columns d = 16 mt + 4p .. + 3
  const int a3addr = (4 * g + (i16 >> 2)) * RROW + 8 * (i16 & 3);
  unsigned char* scr = smem + 2 * BUFB + wave * (DQX ? NSUB : 1) * SCRB;      // this wave's dS image(s): [subtile (DQX)][piece][key 0..31][SROW]
  float* sdq = reinterpret_cast<float*>(scr);                                  // (per-wave form) the wave's partial tile, aliased on its image
  unsigned char* const xpart = smem + 2 * BUFB + G::IMG_BYTES;                 // (DQX) partial tiles [key half][D][DQS]
  // the dS image of this wave: [piece][key 0..31][SROW bytes of 32 queries]
  const int swaddr = i16 * SROW + 8 * g;                              // + piece * SPART + kt * 16 * SROW + jq * 32
  const int sraddr = (8 * g + (i16 >> 2)) * SROW + 8 * (i16 & 3);     // + piece * SPART + jq * 32 (+ 4 * SROW: second half)
  // x 2^-8 of a packed-fp16 operand (asm: written in C, hipcc 7.2 broadcasts word 0 of such a tuple: attention_h2.hip)
  auto down = [&](const u32x4& x) {
    unsigned w4[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) asm("v_pk_mul_f16 %0, %1, %2" : "=v"(w4[w]) : "v"(x[w]), "v"(PAIR_DOWN2));
    return u32x4{w4[0], w4[1], w4[2], w4[3]};
  };

  for (int kb = kb_begin; kb < kb_end; ++kb) {
    const int key0 = kb * KB + wave * 32;
    // ---- stationary operands of the wave's 32 keys, straight from the workspace
    u32x4 kB[2][NT], vB[2][2], kT[MT][2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int p = (D == 16) ? 2 * j + (hi ? 1 : 0) : j;         // pieces (y0, y0 2^-8, y1 2^8, y1)
        const size_t off = (size_t)p * piece_n + (size_t)(key0 + kt * 16 + i16) * D + doff;
        kB[kt][j] = *reinterpret_cast<const u32x4*>(wsh + (size_t)S_K * piece_n + off);
      }
    // V' pieces (v0, v1), form (ii): d 16 -- MFMA j = (o0 | o1)(v_j | v_j); d 32 -- o0 v0, o1 v0, o0 v1
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int p = 0; p < 2; ++p)
        vB[kt][p] = *reinterpret_cast<const u32x4*>(wsh + (size_t)(S_V + p) * piece_n + (size_t)(key0 + kt * 16 + i16) * D + doff);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int p = 0; p < 2; ++p)
        if (!DQX) kT[mt][p] = *reinterpret_cast<const u32x4*>(wsh + (size_t)(S_KT + p) * piece_n + (size_t)(16 * mt + i16) * L + key0 + 8 * g);
    // (DQX) k^T of the four waves of this wave's key half: [source wave][piece], rows d = i16, 32 keys along the contraction
    u32x4 kT4[DQX ? 4 : 1][2];
    if (DQX) {
#pragma unroll
      for (int sw = 0; sw < 4; ++sw)
#pragma unroll
        for (int p = 0; p < 2; ++p)
          kT4[sw][p] = *reinterpret_cast<const u32x4*>(wsh + (size_t)(S_KT + p) * piece_n + (size_t)((D == 16 ? 0 : 16 * ((wave & 3) >> 1)) + i16) * L + kb * KB + (4 * (wave >> 2) + sw) * 32 + 8 * g);
    }
    f32x4 dKa[2][MT], dVa[2][MT];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        dKa[kt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        dVa[kt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
      }

    __syncthreads();              // the previous key block's last tile is fully consumed
    stage_load(0);
    dma_tile(0, 0);
    stage_store(0);
    if (BDMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!(H2B_ABL & 8)) stage_load(1);

    // dQ goes to the key range's slab (layout [tile][d][TQ queries]: 4 KB per tile, contiguous): a plain store during the
    // range's first key block, fire-and-forget L2 float adds afterwards.  Only THIS thread ever touches its four slab words,
    // in program order, so the sums are formed in a fixed order (bitwise reproducible) although the adder sits in L2 --
    // and the old value never travels to the CU: no load to wait for, half the slab bytes on the CU's memory path.
    const bool first_kb = (kb == kb_begin);
    for (int t = 0; t < ntiles; ++t) {
      const int buf = t & 1;
      const unsigned char* tb = smem + buf * BUFB;
      // (BDMA) the rows of tile t + 1 into the other buffer: its piece regions were last read before the previous tile's first barrier
      if (!(H2B_ABL & 8)) dma_tile(t + 1 < ntiles ? t + 1 : t, buf ^ 1);
      float* pdst = part + (size_t)t * (D * TQ) + tid;

      f32x4 dQt[NSUB][2][MT];
#pragma unroll
      for (int sub = 0; sub < NSUB; ++sub) {
        const unsigned char* sb = tb + sub * 32 * RROW;
        u32x4 qA[2][NT], oA[2][NT];          // [.][0 .. NA - 1] read, [.][NA ..] = those times 2^-8
#pragma unroll
        for (int jq = 0; jq < 2; ++jq) {
#pragma unroll
          for (int j = 0; j < NA; ++j) {
            qA[jq][j] = *reinterpret_cast<const u32x4*>(sb + QA_OFF + a1addr[j] + jq * 16 * RROW);
            oA[jq][j] = *reinterpret_cast<const u32x4*>(sb + OA_OFF + a1addr[j] + jq * 16 * RROW);
          }
#pragma unroll
          for (int j = NA; j < NT; ++j) { qA[jq][j] = down(qA[jq][j - NA]); oA[jq][j] = oA[jq][j - NA]; }      // dO', V': no shifts (form (ii))
        }
        f32x4 negl[2], negd[2];
#pragma unroll
        for (int jq = 0; jq < 2; ++jq) {
          negl[jq] = *reinterpret_cast<const f32x4*>(tb + SL_OFF + (32 * sub + 16 * jq + 4 * g) * 4);
          negd[jq] = *reinterpret_cast<const f32x4*>(tb + SD_OFF + (32 * sub + 16 * jq + 4 * g) * 4);
        }
        // A operands of the q-summed products for M tile mt: dO'^T / (q c_q)^T rows d = 16 mt .., 32 queries along the contraction;
        // piece 0 = x0, piece 1 = x1 2^8
        auto load_transposed = [&](int mt, u32x4 (&qT)[2], u32x4 (&oT)[2]) {
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            const unsigned char* sq = sb + QE_OFF + p * RPART + a3addr + 32 * mt;      // q c_q
            const u32x2 q0 = lds_read_tr16(sq), q1 = lds_read_tr16(sq + 16 * RROW);
            qT[p] = u32x4{q0[0], q0[1], q1[0], q1[1]};
            const unsigned char* sop = sb + OH_OFF + p * RPART + a3addr + 32 * mt;      // dO 2^so
            const u32x2 o0 = lds_read_tr16(sop), o1 = lds_read_tr16(sop + 16 * RROW);
            oT[p] = u32x4{o0[0], o0[1], o1[0], o1[1]};
          }
        };
        u32x4 qT0[2], oT0[2];
        if (MT == 1) load_transposed(0, qT0, oT0);       // d 16: read once per subtile, used by both key tiles

        // term order of the four-term products: d 16 -- MFMA j = A_j B_j; d 32 -- (x0, y0), (x1 2^8, y0 2^-8), (x0 2^-8, y1 2^8), (x1, y1):
        // row operand index {0, 1, 2, 3} = x0, x1 2^8, x0 2^-8, x1; stationary piece index {0, 1, 2, 3} = y0, y0 2^-8, y1 2^8, y1
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
          f32x4 S[2], dP[2];
#pragma unroll
          for (int jq = 0; jq < 2; ++jq) {
            f32x4 acc = negl[jq];
#pragma unroll
            for (int j = 0; j < NT; ++j) acc = mfma_f16(qA[jq][j], kB[kt][j], acc);          // large term first
            S[jq] = acc;
            acc = negd[jq];
            // (mutation test, bit 16: the cross product o0 v1 dropped -- with its MFMA partner o1 v1 at d 16)
#pragma unroll
            for (int j = 0; j < NT; ++j)
              if (!((HDIFF_MUTANT & 16) && j == (D == 16 ? 1 : 2))) acc = mfma_f16(oA[jq][j], vB[kt][D == 16 ? j : j >> 1], acc);
            dP[jq] = acc;
          }
          u32x4 Pp[2], Sp[2];
#pragma unroll
          for (int jq = 0; jq < 2; ++jq) {
            float p[4], ds[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              p[i] = __builtin_amdgcn_exp2f(S[jq][i]);
              ds[i] = p[i] * dP[jq][i];
            }
            if (H2B_ABL & 16) {
#pragma unroll
              for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int pc = 0; pc < 2; ++pc) {
                  Pp[pc][2 * jq + h] = __builtin_bit_cast(unsigned, S[jq][2 * h]) + pc;
                  Sp[pc][2 * jq + h] = __builtin_bit_cast(unsigned, dP[jq][2 * h + 1]) + pc;
                }
            } else {
              unsigned sa0, sa1, sb0, sb1;
              split2_scaled(ds[0], ds[1], ds[2], ds[3], dsdn, up8, sa0, sa1, sb0, sb1);
              if (HDIFF_MUTANT & 64) { sa1 &= 0xffe0ffe0u; sb1 &= 0xffe0ffe0u; }      // (mutation test, bit 64: the low five bits of the second piece of dS': 2^-17 of dS)
              Sp[0][2 * jq] = sa0; Sp[1][2 * jq] = sa1; Sp[0][2 * jq + 1] = sb0; Sp[1][2 * jq + 1] = sb1;
              unsigned pa0, pa1, pb0, pb1;
              split2(p[0], p[1], one, pa0, pa1);
              split2(p[2], p[3], one, pb0, pb1);
              Pp[0][2 * jq] = pa0; Pp[1][2 * jq] = pa1; Pp[0][2 * jq + 1] = pb0; Pp[1][2 * jq + 1] = pb1;
            }
            // the packed dS pieces of (key i16 of tile kt, queries 16 jq + 4g ..+3) into the wave's [key][query] image
            if (!(H2B_ABL & 4)) {
#pragma unroll
              for (int pc = 0; pc < 2; ++pc)
                *reinterpret_cast<u32x2*>(scr + (DQX ? sub * SCRB : 0) + pc * SPART + kt * 16 * SROW + jq * 32 + swaddr) = u32x2{Sp[pc][2 * jq], Sp[pc][2 * jq + 1]};
            }
          }
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            u32x4 qTm[2], oTm[2];
            if (MT > 1) load_transposed(mt, qTm, oTm);       // d 32: one M tile's operands at a time (registers)
            const u32x4 (&qT)[2] = (MT == 1 ? qT0 : qTm);
            const u32x4 (&oT)[2] = (MT == 1 ? oT0 : oTm);
            // small terms first; (mutation test, bit 32: the product o1 p0 of dV^T dropped)
            if (!(HDIFF_MUTANT & 32)) dVa[kt][mt] = mfma_f16(oT[1], Pp[0], dVa[kt][mt]);        // o1 p0
            dVa[kt][mt] = mfma_f16(oT[0], Pp[1], dVa[kt][mt]);        // o0 p1
            dVa[kt][mt] = mfma_f16(oT[0], Pp[0], dVa[kt][mt]);        // o0 p0
            dKa[kt][mt] = mfma_f16(qT[1], Sp[0], dKa[kt][mt]);        // q1 s0
            dKa[kt][mt] = mfma_f16(down(qT[0]), Sp[1], dKa[kt][mt]);  // (q0 2^-8) (s1 2^8)
            dKa[kt][mt] = mfma_f16(qT[0], Sp[0], dKa[kt][mt]);        // q0 s0
          }
        }

        // ---- (per-wave form) dQ^T of the subtile over this wave's 32 keys: the dS image read back transposed, keys along the contraction
        asm volatile("" ::: "memory");
#pragma unroll
        for (int jq = 0; jq < (DQX ? 0 : 2); ++jq) {
          u32x4 sT[2];
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            const unsigned char* src = scr + p * SPART + jq * 32 + sraddr;
            const u32x2 lo = lds_read_tr16(src), hi2 = lds_read_tr16(src + 4 * SROW);
            sT[p] = u32x4{lo[0], lo[1], hi2[0], hi2[1]};
          }
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            f32x4 acca = f32x4{0.f, 0.f, 0.f, 0.f}, accb = f32x4{0.f, 0.f, 0.f, 0.f};
            acca = mfma_f16(kT[mt][0], sT[0], acca);        // k0 s0
            accb = mfma_f16(kT[mt][1], sT[0], accb);        // (k1 2^8) s0
            accb = mfma_f16(kT[mt][0], sT[1], accb);        // k0 (s1 2^8)
#pragma unroll
            for (int r = 0; r < 4; ++r) dQt[sub][jq][mt][r] = __builtin_fmaf(accb[r], dnb, acca[r]);
          }
        }
        asm volatile("" ::: "memory");      // the next subtile's image stores stay behind these reads (same wave: in order)
      }

      if (DQX) {
        // ---- dQ^T across the workgroup: every wave's images of this tile are complete behind the barrier
        lds_barrier();
        // output tile nt of four: d 16 -- queries 16 nt .. + 15 (subtile nt >> 1, 16-query group nt & 1); d 32 -- 16-query group nt & 1, rows d = 16 (nt >> 1) ..;
        // keys of waves 4 kh .. 4 kh + 3
        const int nt = wave & 3, kh = wave >> 2;
        const int osub = (D == 16) ? (nt >> 1) : 0, ojq = nt & 1, omt = (D == 16) ? 0 : (nt >> 1);
        f32x4 acca = f32x4{0.f, 0.f, 0.f, 0.f}, accb = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sw = 0; sw < 4; ++sw) {
          const unsigned char* img = smem + 2 * BUFB + ((4 * kh + sw) * NSUB + osub) * SCRB + ojq * 32 + sraddr;
          u32x4 sT[2];
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            const u32x2 lo = lds_read_tr16(img + p * SPART), hi2 = lds_read_tr16(img + p * SPART + 4 * SROW);
            sT[p] = u32x4{lo[0], lo[1], hi2[0], hi2[1]};
          }
          acca = mfma_f16(kT4[sw][0], sT[0], acca);        // k0 s0
          accb = mfma_f16(kT4[sw][1], sT[0], accb);        // (k1 2^8) s0
          accb = mfma_f16(kT4[sw][0], sT[1], accb);        // k0 (s1 2^8)
        }
        if (!(H2B_ABL & 2)) {
          float* pw = reinterpret_cast<float*>(xpart + kh * G::PART_BYTES);
#pragma unroll
          for (int r = 0; r < 4; ++r) pw[(16 * omt + 4 * g + r) * DQS + 32 * osub + 16 * ojq + i16] = __builtin_fmaf(accb[r], dnb, acca[r]);
        } else if (acca[0] + accb[1] == 12345.f) *pdst = 1.f;
        // tile t + 1 into the other buffer BEFORE the barrier: the next tile starts without another one
        stage_store(buf ^ 1);
        if (BDMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's runs of tile t + 1 have landed
        lds_barrier();
        constexpr int NR = 1024 / MTHREADS;
        float sum[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const int e = tid + MTHREADS * r;
          const float* s0 = reinterpret_cast<const float*>(xpart) + (e / TQ) * DQS + (e % TQ);
          const float v = s0[0] + s0[G::PART_BYTES / 4];
          sum[r] = __builtin_ldexpf(v * a.inv_sqrt_d, -(P_UP - DS_DOWN<D> + so + sv + bal)) * *reinterpret_cast<const float*>(tb + SC_OFF + (e % TQ) * 4);
        }
        if (!(H2B_ABL & 8)) stage_load(t + 2 < ntiles ? t + 2 : t);
        if (!(H2B_ABL & 67) || t == 0) {
          if (first_kb || (H2B_ABL & 32)) {
#pragma unroll
            for (int r = 0; r < NR; ++r) pdst[MTHREADS * r] = sum[r];
          } else {
#pragma unroll
            for (int r = 0; r < NR; ++r) unsafeAtomicAdd(pdst + MTHREADS * r, sum[r]);
          }
        }
        continue;      // no closing barrier: the next tile's first LDS write that anyone else reads is behind ITS barrier
      }
      // (per-wave form) the wave's partial tile [d][query] over its own scratch (its reads above are done: same wave, in order)
      if (H2B_ABL & 2) {
        if (dQt[0][0][0][0] + dQt[0][1][0][1] == 12345.f) *pdst = 1.f;
        stage_store(buf ^ 1);
        if (!(H2B_ABL & 8)) stage_load(t + 2 < ntiles ? t + 2 : t);
      } else {
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
          for (int jq = 0; jq < 2; ++jq)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
              for (int r = 0; r < 4; ++r) sdq[(16 * mt + 4 * g + r) * DQS + 32 * sub + 16 * jq + i16] = dQt[sub][jq][mt][r];
        lds_barrier();
        // thread -> floats tid + MTHREADS r of the [D][TQ] block: row d = e / TQ, query e % TQ
        constexpr int NR = 1024 / MTHREADS;
        static_assert(D * TQ == 1024, "slab tile");
        float sum[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const int e = tid + MTHREADS * r;
          const float* s0 = reinterpret_cast<const float*>(smem + 2 * BUFB) + (e / TQ) * DQS + (e % TQ);
          float v = s0[0];
#pragma unroll
          for (int w = 1; w < MW; ++w) v += s0[w * (SCRB / 4)];
          // Y = dQ_q sqrt(d) 2^(bal + so + t_q + sv + 14 - DS_DOWN)
          sum[r] = __builtin_ldexpf(v * a.inv_sqrt_d, -(P_UP - DS_DOWN<D> + so + sv + bal)) * *reinterpret_cast<const float*>(tb + SC_OFF + (e % TQ) * 4);
        }
        // tile t + 1 into LDS, then the loads of tile t + 2
        stage_store(buf ^ 1);
        if (!(H2B_ABL & 8)) stage_load(t + 2 < ntiles ? t + 2 : t);
        if (!(H2B_ABL & 65) || t == 0) {
          if (first_kb || (H2B_ABL & 32)) {
#pragma unroll
            for (int r = 0; r < NR; ++r) pdst[MTHREADS * r] = sum[r];
          } else {
#pragma unroll
            for (int r = 0; r < NR; ++r) unsafeAtomicAdd(pdst + MTHREADS * r, sum[r]);
          }
        }
      }
      lds_barrier();
    }

    // ---- dK, dV of this key block (complete: the sweep covered every query).  X = dK sqrt(d) 2^(sq + so + sv + 14 - DS_DOWN) (the rows of Q 2^sq c_q),
    // Z = dV 2^(14 + so)
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      const int key = key0 + kt * 16 + i16;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int d = 16 * mt + 4 * g + r;
          kout[(size_t)d * L + key] = __builtin_ldexpf(dKa[kt][mt][r] * a.inv_sqrt_d, -(P_UP - DS_DOWN<D> + so + sv + sq));
          vout[(size_t)d * L + key] = __builtin_ldexpf(dVa[kt][mt][r], -(P_UP + so));
        }
    }
  }
}

// dqkv[b][head * D + d][q] (Q third) = sum over key ranges, in order, of the tile-major slabs
// [split][B][heads][L / TQ][D][TQ] (already scaled by 1/sqrt(d)).  Thread = four neighbouring queries of one (d, tile).
template <int D>
__global__ void mha_dq_reduce_h2_kernel(const float* __restrict__ part, float* __restrict__ dqkv, int nsplit, int C, int L,
                                        size_t split_stride) {
  constexpr int TQ = Geo<D>::TQ, Q4 = TQ / 4;
  const int b = blockIdx.y;
  const size_t per_sample = (size_t)C * L;
  const float* src = part + (size_t)b * per_sample;
  float* dst = dqkv + (size_t)b * 3 * per_sample;
  const size_t n4 = per_sample >> 2;
  const int tiles = L / TQ;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    // i indexes the slab in its own order: head, tile, d, TQ / 4 groups of 4 queries
    f32x4 acc = reinterpret_cast<const f32x4*>(src)[i];
    for (int sp = 1; sp < nsplit; ++sp) acc += reinterpret_cast<const f32x4*>(src + (size_t)sp * split_stride)[i];
    const int q4 = (int)(i % Q4), d = (int)((i / Q4) % D);
    const size_t ht = i / ((size_t)Q4 * D);        // head * tiles + tile
    const size_t head = ht / tiles, tile = ht - head * tiles;
    reinterpret_cast<f32x4*>(dst + (head * D + d) * (size_t)L + tile * TQ)[q4] = acc;
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// Round 6: the d_head 16 kernel as a SOFTWARE PIPELINE over 32-query tiles, ONE workgroup barrier per tile.
// Why (profiles/r05_pmc_summary.txt): the kernel above runs at ~52 % of its own issue floor, MFMA / vector co-execution
// 0.25 -- per 64-query tile all eight waves meet at two barriers, and between them sits a phase that is nothing but latency: the
// transposed reads of the dS images, a chain of dependent dQ MFMAs, the partial tile through LDS, a second barrier, the cross-wave sum and
// the slab adds, with the matrix AND the vector pipe idle on every SIMD at once (the dQ output path: 23 % of the launch).
// Here the three pieces of a tile's work run in three consecutive iterations, each between the same two barriers as the MAIN phase of a
// later tile, so their latencies hide under its instruction stream:
//   iteration u:   main(u)   S', dP', P', dS', dV^T, dK^T of tile u from tile buffer u % 3; dS' images -> image set u & 1
//                  dq(u - 1) dQ'^T of tile u - 1 contracted across the workgroup out of image set (u - 1) & 1 (complete since the last
//                            barrier): wave w = (key quarter w >> 1, 16-query group w & 1), partial tile -> xpart[(u - 1) & 1]
//                  sum(u - 2) the four key quarters of tile u - 2 summed out of xpart[u & 1], unscaled, to the slab (store / L2 float add)
//                  LDS-DMA of tile u + 2 into tile buffer (u + 2) % 3 (last read before the previous barrier); the ROW operands and chain starts of
//                  tile u + 1 (in LDS since the barrier before) are read behind the last MFMA that uses tile u's: the next iteration opens with an MFMA
//   barrier
// Nothing per query is needed after the main phase: the two chain starts (14 - lse2, -delta') come per tile from the split pass and the factor
// c_q = 2^-t_q is given back by the slab reduce kernel (a power of two: exact).  Same operand formats, same products, same summation order
// inside a tile as the kernel above; dQ's key-quarter partials are summed in a fixed order: bitwise reproducible.
// What it bought (profiles/r06_pmc_summary.txt, r06_attention_bwd_pipeline.txt): 7.5 % fewer cycles, co-execution 0.25 -> 0.41 -- and 1-2 % of
// wall time, because the board is on its power limit and returns the cycles as clock: the kernel is bound by the energy of its instructions.
// ---------------------------------------------------------------------------------------------------------------------
#ifndef H2B_PIPE
#define H2B_PIPE 1            // dev: 0 = d_head 16 on the two-barrier kernel above (A/B)
#endif
#ifndef H2P_OPT
#define H2P_OPT 3             // switches of the pipelined kernel (A/B builds with -DH2P_OPT=<bits>): 1 static priority for waves 4-7 (the second-dispatched
#endif                        // half loses every issue arbitration to its SIMD partner otherwise: -1.3 %), 2 no trailing s_nop in the dS split statements
                              // (the slot program puts an MFMA behind each: the partial-write hazard is covered; 0.0 %)
#ifndef H2P_T4
#define H2P_T4 1              // dev: 0 = the fourth terms of the two d-contracted products (q1 k1 in S', o1 v1 in dP': 2^-22 of their product) multiplied by
#endif                        // ZEROS -- the stationary operands' high-half lanes cleared once per key block: same MFMAs, idle multipliers (energy A/B)
#ifndef H2P_KTREG
#define H2P_KTREG 1           // 1 = the dq stage's k^T operands in 16 registers for the whole sweep (the steady loop has no spill with them; -1.3 %),
#endif                        // 0 = four LDS reads per tile out of a copy in MFMA operand order (A/B)
#ifndef H2P_ABL
#define H2P_ABL 0             // dev: timing ablations of the pipelined kernel (results wrong by construction): 1 no exp / split chain, 2 no dq stage,
#endif                        // 4 no sum stage, 8 no main-phase MFMAs, 16 no dS image stores, 32 no barrier, 64 no tile copies, 128 no operand LDS reads
struct GeoP {
  static constexpr int D = 16, TQ = 32, RROW = 32, RPART = TQ * RROW;      // a tile's piece region: 32 rows of 32 bytes = one 1 KiB LDS-DMA run
  static constexpr int QA_OFF = 0, QE_OFF = 2 * RPART, OA_OFF = 4 * RPART, OH_OFF = 6 * RPART, SL_OFF = 8 * RPART, SD_OFF = SL_OFF + TQ * 4;
  static constexpr int BUFB = SD_OFF + TQ * 4;                              // 8448
  static constexpr int IMGB = MW * SCRB;                                    // one image set: every wave's [piece][32 keys][SROW]
  static constexpr int DQS = TQ + 4;                                        // row stride (floats) of a partial tile [16 d][DQS]: conflict-free b32 stores
  static constexpr int PARTB = D * DQS * 4, XPB = 4 * PARTB;                // four key quarters
  static constexpr int NBUF = 3;                                            // tile buffers: tile t lives in buffer t % 3 (copied two iterations ahead)
  static constexpr int KTB = MW * 2 * 1024;                                 // k^T of every wave's 32 keys as MFMA A operands: [wave][piece][lane][16 bytes]
  static constexpr int IMG_OFF = NBUF * BUFB, XP_OFF = IMG_OFF + 2 * IMGB, KT_OFF = XP_OFF + 2 * XPB, LDS_BYTES = KT_OFF + KTB;      // 133 888
  static_assert(MW == 8 || !H2B_PIPE, "the pipelined kernel is written for eight waves");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

// 256 bytes global -> LDS: lane i's 4 bytes at src + 4 i land at lds_dst + 4 i (global_load_lds_dword; see dma_1k)
__device__ __forceinline__ void dma_256(const unsigned char* src, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(src), "s"(lds_dst)
               : "memory");
}

// the three stages of split2_scaled as separate statements (same instructions, same order per value): the pipelined kernel places them in
// different issue slots between its MFMAs
__device__ __forceinline__ void ds_split_a(float a0, float a1, float b0, float b1, float c, unsigned& ha0, unsigned& hb0) {
  asm("v_fma_mixlo_f16 %0, %2, %6, 0\n\t"
      "v_fma_mixlo_f16 %1, %4, %6, 0\n\t"
      "v_fma_mixhi_f16 %0, %3, %6, 0\n\t"
      "v_fma_mixhi_f16 %1, %5, %6, 0"
#if !(H2P_OPT & 2)
      "\n\ts_nop 0"
#endif
      : "=&v"(ha0), "=&v"(hb0)
      : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(c));
}
__device__ __forceinline__ void ds_split_b(float a0, float a1, float b0, float b1, float c, unsigned ha0, unsigned hb0, float& r0, float& r1, float& r2,
                                           float& r3) {
  asm("v_fma_mix_f32 %0, %4, %8, -%9 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mix_f32 %2, %6, %8, -%10 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mix_f32 %1, %5, %8, -%9 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
      "v_fma_mix_f32 %3, %7, %8, -%10 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
      : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
      : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(c), "v"(ha0), "v"(hb0));
}
__device__ __forceinline__ void ds_split_c(float r0, float r1, float r2, float r3, float up, unsigned& ha1, unsigned& hb1) {
  asm("v_fma_mixlo_f16 %0, %2, %6, 0\n\t"
      "v_fma_mixlo_f16 %1, %4, %6, 0\n\t"
      "v_fma_mixhi_f16 %0, %3, %6, 0\n\t"
      "v_fma_mixhi_f16 %1, %5, %6, 0"
#if !(H2P_OPT & 2)
      "\n\ts_nop 0"
#endif
      : "=&v"(ha1), "=&v"(hb1)
      : "v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(up));
}

__global__ __launch_bounds__(MTHREADS) void mha_bwd_h2p_kernel(const BwdH2Args a) {
  constexpr int D = 16;
  using G = GeoP;
  constexpr int TQ = G::TQ, RROW = G::RROW, RPART = G::RPART, BUFB = G::BUFB, DQS = G::DQS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int C = a.C, L = a.L;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  const TileId tile = xcd_tile();
  const int head = tile.head, b = tile.b, heads = gridDim.y, split = tile.x;
  const size_t piece_n = (size_t)L * D;
  const __bf16* wsh = a.ws + ((size_t)b * heads + head) * S_COUNT * piece_n;
  const unsigned char* wsb = reinterpret_cast<const unsigned char*>(wsh);
  const unsigned char* nld = wsb + (size_t)S_C * piece_n * 2 + (size_t)L * 4;      // [tile][2][32] floats: 14 - lse2, -delta'
  float* part = a.dq_part + (size_t)split * a.split_stride + (size_t)b * a.batch_stride + (size_t)head * D * L;
  float* kout = a.dqkv + ((size_t)b * 3 * C + (size_t)C + (size_t)head * D) * L;
  float* vout = kout + (size_t)C * L;
  const int ntiles = L / TQ;
  const int nkb_total = L / KB;
  const int kb_begin = split * a.kb_per_split;
  const int kb_end = (kb_begin + a.kb_per_split < nkb_total) ? kb_begin + a.kb_per_split : nkb_total;
  const unsigned* mx = a.absmax + ((size_t)b * heads + head) * M_COUNT;
  const int so = scale_exp_o(mx[M_O]), sv = scale_exp_v(mx[M_V]), sq = scale_exp_q(mx[M_Q]), bal = balance_exp(mx[M_Q], mx[M_K], a.qscale);
  const float one = a.one;
  const float dnb = __builtin_ldexpf(1.0f, -PAIR_SHIFT);
  const float dsdn = __builtin_ldexpf(one, -DS_DOWN<D>);
  const float up8 = __builtin_ldexpf(one, PAIR_SHIFT);
  const int dq_unscale = -(P_UP - DS_DOWN<D> + so + sv + bal);

  const int doff = 8 * (g & 1);
  const bool hi = g >> 1;
  const unsigned lds0 = (unsigned)(size_t)(lds_byte*)smem;
  // VMEM roles by wave half, so that no wave ever waits for a slab add (a no-return float add stays counted in vmcnt for ~3000 cycles with every
  // CU issuing, vmcnt retires in order, and a tile lasts less than that): waves 0-3 copy the tile -- wave w the two piece regions 2 w, 2 w + 1
  // (slots S_Q, S_QE, S_O, S_OH and their second pieces), wave 0 also the two start vectors -- and wait for their copies; waves 4-7 send the
  // finished partial tile to the slab (two floats per thread) and never wait for memory
  const int w_ = __builtin_amdgcn_readfirstlane(wave);
  const bool copier = w_ < 4;
  const unsigned char* dma_src = wsb + (size_t)(w_ == 0 ? S_Q : w_ == 1 ? S_QE : w_ == 2 ? S_O : S_OH) * (piece_n * 2);
  auto dma_tile = [&](int t, int boff) {
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + boff + (w_ & 3) * 2 * RPART);
    dma_1k(dma_src + (size_t)t * RPART, lane * 16, dst);
    dma_1k(dma_src + piece_n * 2 + (size_t)t * RPART, lane * 16, dst + RPART);
    if (w_ == 0) dma_256(nld + (size_t)t * 256, lane * 4, __builtin_amdgcn_readfirstlane(lds0 + boff + G::SL_OFF));
  };

  const int a1addr = (hi ? 1 : 0) * RPART + i16 * RROW + doff * 2;                  // row reads (Q, dO): the piece of this lane half, row i16
  const int a3addr = (4 * g + (i16 >> 2)) * RROW + 8 * (i16 & 3);                   // transposed reads of the same tiles
  const int swaddr = i16 * SROW + 8 * g;                                            // dS image stores: + piece * SPART + kt * 16 * SROW + jq * 32
  const int sraddr = (8 * g + (i16 >> 2)) * SROW + 8 * (i16 & 3);                   // dS image transposed reads
  // dq role of this wave: 16-query group, key quarter (source waves 2 kq, 2 kq + 1)
  const int ojq = wave & 1, kq = wave >> 1;
  auto down = [&](const u32x4& x) {
    unsigned w4[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) asm("v_pk_mul_f16 %0, %1, %2" : "=v"(w4[w]) : "v"(x[w]), "v"(PAIR_DOWN2));
    return u32x4{w4[0], w4[1], w4[2], w4[3]};
  };
#define SLOT() __builtin_amdgcn_sched_barrier(0)

  if ((H2P_OPT & 1) && !copier) __builtin_amdgcn_s_setprio(1);
  for (int kb = kb_begin; kb < kb_end; ++kb) {
    const int key0 = kb * KB + wave * 32;
    u32x4 kB[2][2], vB[2][2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const size_t row = (size_t)(key0 + kt * 16 + i16) * D + doff;
        kB[kt][j] = *reinterpret_cast<const u32x4*>(wsh + (size_t)(S_K + 2 * j + (hi ? 1 : 0)) * piece_n + row);      // (k0 | k0 2^-8), (k1 2^8 | k1)
        vB[kt][j] = *reinterpret_cast<const u32x4*>(wsh + (size_t)(S_V + j) * piece_n + row);                          // (o0 | o1)(v_j | v_j)
        if (!H2P_T4 && j == 1 && hi) { kB[kt][j] = u32x4{0u, 0u, 0u, 0u}; vB[kt][j] = u32x4{0u, 0u, 0u, 0u}; }
      }
    // k^T of this wave's 32 keys (rows d = i16, keys along the contraction) in MFMA operand order into LDS: the dq stage reads the two waves
    // of its key quarter from there every tile (lane-contiguous 16-byte reads) instead of holding 16 registers for the whole sweep
    u32x4 kTown[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) kTown[p] = *reinterpret_cast<const u32x4*>(wsh + (size_t)(S_KT + p) * piece_n + (size_t)i16 * L + key0 + 8 * g);
    f32x4 dKa[2], dVa[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) { dKa[kt] = f32x4{0.f, 0.f, 0.f, 0.f}; dVa[kt] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    __syncthreads();              // the previous key block's last iterations are done with every LDS region
#pragma unroll
    for (int p = 0; p < 2; ++p) *reinterpret_cast<u32x4*>(smem + G::KT_OFF + (wave * 2 + p) * 1024 + lane * 16) = kTown[p];
    if (copier) { dma_tile(0, 0); dma_tile(1, BUFB); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // the row operands and chain starts of a tile are read one iteration AHEAD (behind the last MFMA that uses the current ones), so that an
    // iteration opens with matrix work instead of an LDS round trip that both waves of a SIMD sit out together (timing ablation: the operand
    // reads were 18 % of the launch); tile u + 1 is therefore in LDS before the barrier that ends iteration u - 1: three tile buffers
    u32x4 qA[2][2], oA[2];
    f32x4 negl[2], negd[2];
    auto rd_rows = [&](int jq, int boff) {
      const unsigned char* tb = smem + boff;
      if (H2P_ABL & 128) { qA[jq][0] = kB[0][0]; oA[jq] = vB[0][0]; negl[jq] = f32x4{0.f, 0.f, 0.f, 0.f}; negd[jq] = negl[jq]; return; }
      qA[jq][0] = *reinterpret_cast<const u32x4*>(tb + G::QA_OFF + a1addr + jq * 16 * RROW);
      oA[jq] = *reinterpret_cast<const u32x4*>(tb + G::OA_OFF + a1addr + jq * 16 * RROW);
      negl[jq] = *reinterpret_cast<const f32x4*>(tb + G::SL_OFF + (16 * jq + 4 * g) * 4);
      negd[jq] = *reinterpret_cast<const f32x4*>(tb + G::SD_OFF + (16 * jq + 4 * g) * 4);
    };
    rd_rows(0, 0); rd_rows(1, 0);
    qA[0][1] = down(qA[0][0]); qA[1][1] = down(qA[1][0]);
    u32x4 kT2s[2][2];
    if (H2P_KTREG) {
#pragma unroll
      for (int sw = 0; sw < 2; ++sw)
#pragma unroll
        for (int p = 0; p < 2; ++p) kT2s[sw][p] = *reinterpret_cast<const u32x4*>(smem + G::KT_OFF + ((2 * kq + sw) * 2 + p) * 1024 + lane * 16);
    }

    // One iteration = issue slots in a fixed order, fenced (sched_barrier) so that the order written here is the order issued: about one
    // MFMA and one group of ~4 vector instructions per slot (an MFMA holds the vector issue port for 8 of its 16 cycles; left to the scheduler
    // the matrix products clump and the dq / sum stages sink to the end of the block as a serial tail).  Chains of dependent MFMAs are
    // interleaved two and two.
    auto iteration = [&](auto main_tag, auto dq_tag, auto sum_tag, auto dma_tag, auto pre_tag, auto first_tag, int u, int bcur, int bnext, int bnn) {
      constexpr bool MAIN = decltype(main_tag)::value, DQ = decltype(dq_tag)::value && !(H2P_ABL & 2), SUM = decltype(sum_tag)::value && !(H2P_ABL & 4);
      constexpr bool DMA = decltype(dma_tag)::value, PRE = decltype(pre_tag)::value, FIRST = decltype(first_tag)::value;
      const int par = u & 1;
      const unsigned char* tb = smem + bcur;
      unsigned char* scr = smem + G::IMG_OFF + par * G::IMGB + wave * SCRB;
      const unsigned char* img = smem + G::IMG_OFF + (par ^ 1) * G::IMGB + (2 * kq) * SCRB + ojq * 32 + sraddr;      // source wave 2 kq (+ SCRB: 2 kq + 1)
      u32x4 qT[2], oT[2], qTd, sT[2][2];
      u32x4 kT2l[2][2];
      u32x4 (&kT2)[2][2] = H2P_KTREG ? kT2s : kT2l;
      f32x4 S[2][2], dP[2][2], acca, accb;
      unsigned Pp[2][2][4], Sp[2][2][4];          // [key tile][piece][word]: the B operands of dV^T / dK^T over the tile's 32 queries
      auto vec4 = [](const unsigned (&w)[4]) { return u32x4{w[0], w[1], w[2], w[3]}; };
      float pe[2][2][4], ds[2][2][4], rr[2][2][4];
      float sraw[4] = {0.f, 0.f, 0.f, 0.f};
      const float* xs0 = reinterpret_cast<const float*>(smem + G::XP_OFF + par * G::XPB) + (tid >> 5 & 7) * DQS + (tid & 31);
      // ---- memory roles first: the copy of tile u + 1 / the partial tile of tile u - 2 out of xpart[par]
      if (copier) {
        if constexpr (MAIN && DMA && !(H2P_ABL & 64)) dma_tile(u + 2, bnn);
      } else if constexpr (SUM) {
#pragma unroll
        for (int q = 0; q < 4; ++q) sraw[q] = xs0[q * (G::PARTB / 4)];      // rows d = 0 .. 7; summed a few slots later: no wait here
      }
      SLOT();
      auto rd_tr = [&](int p) {
        if (H2P_ABL & 128) { qT[p] = kB[1][p]; oT[p] = vB[1][p]; return; }
        const unsigned char* sqp = tb + G::QE_OFF + p * RPART + a3addr;
        const u32x2 q0 = lds_read_tr16(sqp), q1 = lds_read_tr16(sqp + 16 * RROW);
        qT[p] = u32x4{q0[0], q0[1], q1[0], q1[1]};
        const unsigned char* sop = tb + G::OH_OFF + p * RPART + a3addr;
        const u32x2 o0 = lds_read_tr16(sop), o1 = lds_read_tr16(sop + 16 * RROW);
        oT[p] = u32x4{o0[0], o0[1], o1[0], o1[1]};
      };
      auto rd_img = [&](int sw) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          if (!H2P_KTREG) kT2[sw][p] = *reinterpret_cast<const u32x4*>(smem + G::KT_OFF + ((2 * kq + sw) * 2 + p) * 1024 + lane * 16);
          const u32x2 lo = lds_read_tr16(img + sw * SCRB + p * SPART), hi2 = lds_read_tr16(img + sw * SCRB + p * SPART + 4 * SROW);
          sT[sw][p] = u32x4{lo[0], lo[1], hi2[0], hi2[1]};
        }
      };
      // MFMA steps
      auto mS = [&](int kt, int jq, int j) {
        if (H2P_ABL & 8) { if (j == 0) S[kt][jq] = negl[jq] + __builtin_bit_cast(f32x4, qA[jq][0]); return; }
        S[kt][jq] = mfma_f16(qA[jq][j], kB[kt][j], j == 0 ? negl[jq] : S[kt][jq]); };      // large term first
      auto mP = [&](int kt, int jq, int j) {      // (mutation test, bit 16: the cross product o0 v1 dropped, with its MFMA partner o1 v1)
        if (H2P_ABL & 8) { if (j == 0) dP[kt][jq] = negd[jq] + __builtin_bit_cast(f32x4, oA[jq]); return; }
        if (!((HDIFF_MUTANT & 16) && j == 1)) dP[kt][jq] = mfma_f16(oA[jq], vB[kt][j], j == 0 ? negd[jq] : dP[kt][jq]);
      };
      auto mV = [&](int kt, int j) {              // small terms first: o1 p0, o0 p1, o0 p0; (mutation test, bit 32: o1 p0 dropped)
        if (H2P_ABL & 8) { if (j == 0) dVa[kt] += __builtin_bit_cast(f32x4, vec4(Pp[kt][0])) + __builtin_bit_cast(f32x4, vec4(Pp[kt][1])) + __builtin_bit_cast(f32x4, oT[0]) + __builtin_bit_cast(f32x4, oT[1]); return; }
        if (!((HDIFF_MUTANT & 32) && j == 0)) dVa[kt] = mfma_f16(oT[j == 0 ? 1 : 0], vec4(Pp[kt][j == 1 ? 1 : 0]), dVa[kt]);
      };
      auto mK = [&](int kt, int j) {              // q1 s0, (q0 2^-8)(s1 2^8), q0 s0
        if (H2P_ABL & 8) { if (j == 0) dKa[kt] += __builtin_bit_cast(f32x4, vec4(Sp[kt][0])) + __builtin_bit_cast(f32x4, vec4(Sp[kt][1])) + __builtin_bit_cast(f32x4, qT[0]) + __builtin_bit_cast(f32x4, qT[1]) + __builtin_bit_cast(f32x4, qTd); return; }
        dKa[kt] = mfma_f16(j == 0 ? qT[1] : j == 1 ? qTd : qT[0], vec4(Sp[kt][j == 1 ? 1 : 0]), dKa[kt]);
      };
      auto mQ = [&](int sw, int j) {              // j 0: k0 s0 -> a; 1: (k1 2^8) s0 -> b; 2: k0 (s1 2^8) -> b
        if (j == 0) acca = mfma_f16(kT2[sw][0], sT[sw][0], acca);
        else accb = mfma_f16(kT2[sw][j == 1 ? 1 : 0], sT[sw][j == 1 ? 0 : 1], accb);
      };
      // vector steps of one (key tile, 16-query group): 4 scores per lane
      auto cE = [&](int kt, int jq) {
        if (H2P_ABL & 1) {
#pragma unroll
          for (int w = 0; w < 2; ++w)
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) {
              Pp[kt][pc][2 * jq + w] = __builtin_bit_cast(unsigned, S[kt][jq][2 * w]) + pc;
              Sp[kt][pc][2 * jq + w] = __builtin_bit_cast(unsigned, dP[kt][jq][2 * w + 1]) + pc;
            }
          return;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) pe[kt][jq][i] = __builtin_amdgcn_exp2f(S[kt][jq][i]);
      };
      auto cM = [&](int kt, int jq) {
        if (H2P_ABL & 1) return;
#pragma unroll
        for (int i = 0; i < 4; ++i) ds[kt][jq][i] = pe[kt][jq][i] * dP[kt][jq][i];
      };
      auto cA = [&](int kt, int jq) {
        if (H2P_ABL & 1) return; ds_split_a(ds[kt][jq][0], ds[kt][jq][1], ds[kt][jq][2], ds[kt][jq][3], dsdn, Sp[kt][0][2 * jq], Sp[kt][0][2 * jq + 1]); };
      auto cB = [&](int kt, int jq) {
        if (H2P_ABL & 1) return;
        ds_split_b(ds[kt][jq][0], ds[kt][jq][1], ds[kt][jq][2], ds[kt][jq][3], dsdn, Sp[kt][0][2 * jq], Sp[kt][0][2 * jq + 1], rr[kt][jq][0], rr[kt][jq][1],
                   rr[kt][jq][2], rr[kt][jq][3]);
      };
      auto cC = [&](int kt, int jq) {
        if (H2P_ABL & 1) return;
        unsigned h1a, h1b;
        ds_split_c(rr[kt][jq][0], rr[kt][jq][1], rr[kt][jq][2], rr[kt][jq][3], up8, h1a, h1b);
        if (HDIFF_MUTANT & 64) { h1a &= 0xffe0ffe0u; h1b &= 0xffe0ffe0u; }      // (mutation test, bit 64)
        Sp[kt][1][2 * jq] = h1a; Sp[kt][1][2 * jq + 1] = h1b;
      };
      auto cP = [&](int kt, int jq) {
        if (H2P_ABL & 1) return;
        unsigned pa0, pa1, pb0, pb1;
        split2(pe[kt][jq][0], pe[kt][jq][1], one, pa0, pa1);
        split2(pe[kt][jq][2], pe[kt][jq][3], one, pb0, pb1);
        Pp[kt][0][2 * jq] = pa0; Pp[kt][1][2 * jq] = pa1; Pp[kt][0][2 * jq + 1] = pb0; Pp[kt][1][2 * jq + 1] = pb1;
      };
      auto cW = [&](int kt, int jq) {
        if (H2P_ABL & 16) return;             // (key i16 of tile kt, queries 16 jq + 4 g ..+3) into the wave's [key][query] image
#pragma unroll
        for (int pc = 0; pc < 2; ++pc)
          *reinterpret_cast<u32x2*>(scr + pc * SPART + kt * 16 * SROW + jq * 32 + swaddr) = u32x2{Sp[kt][pc][2 * jq], Sp[kt][pc][2 * jq + 1]};
      };

      if constexpr (DQ) { rd_img(0); acca = f32x4{0.f, 0.f, 0.f, 0.f}; accb = f32x4{0.f, 0.f, 0.f, 0.f}; }
      SLOT();
      if constexpr (MAIN) {
        mS(0, 0, 0); SLOT();
        mP(0, 0, 0); SLOT();
        mS(0, 0, 1); SLOT();
        mP(0, 0, 1); SLOT();
      }
      // the slab floats of tile u - 2 (waves 4-7): thread -> floats tid - 256 (rows d < 8) and tid (rows d >= 8) of the [16 d][32 q] block
      auto slab_out = [&](int r) {
        if (!copier) {
          float* pdst = part + (size_t)(u - 2) * (D * TQ) + (tid - 256) + 256 * r;
          const float v = ((sraw[0] + sraw[1]) + sraw[2]) + sraw[3];
          const float y = __builtin_ldexpf(v * a.inv_sqrt_d, dq_unscale);         // dQ_q 2^t_q: the reduce kernel gives c_q back
          if constexpr (FIRST) *pdst = y;
          else unsafeAtomicAdd(pdst, y);
          if (r == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) sraw[q] = xs0[8 * DQS + q * (G::PARTB / 4)];
          }
        }
      };
      if constexpr (SUM) { slab_out(0); SLOT(); }
      if constexpr (MAIN) {
        mS(0, 1, 0); cE(0, 0); SLOT();
        mP(0, 1, 0); cM(0, 0); SLOT();
        mS(0, 1, 1); cA(0, 0); rd_tr(0); SLOT();
        mP(0, 1, 1); cB(0, 0); SLOT();
      }
      if constexpr (SUM) { slab_out(1); SLOT(); }
      if constexpr (MAIN) {
      }
      if constexpr (DQ) {
        mQ(0, 1); if constexpr (MAIN) cC(0, 0); SLOT();
        mQ(0, 0); if constexpr (MAIN) cP(0, 0); SLOT();
        mQ(0, 2); if constexpr (MAIN) { cW(0, 0); cE(0, 1); } rd_img(1); SLOT();
      } else if constexpr (MAIN) { cC(0, 0); cP(0, 0); cW(0, 0); cE(0, 1); SLOT(); }
      if constexpr (MAIN) {
        mS(1, 0, 0); cM(0, 1); rd_tr(1); SLOT();
        mP(1, 0, 0); cA(0, 1); SLOT();
        mS(1, 0, 1); cB(0, 1); SLOT();
        mP(1, 0, 1); cC(0, 1); SLOT();
        mS(1, 1, 0); cP(0, 1); SLOT();
        mP(1, 1, 0); cW(0, 1); qTd = down(qT[0]); SLOT();
        mS(1, 1, 1); cE(1, 0); SLOT();
        mP(1, 1, 1); cM(1, 0); SLOT();
        if constexpr (PRE) { rd_rows(0, bnext); rd_rows(1, bnext); SLOT(); }
        mV(0, 0); cA(1, 0); SLOT();
        mK(0, 0); cB(1, 0); SLOT();
        mV(0, 1); cC(1, 0); SLOT();
        mK(0, 1); cP(1, 0); SLOT();
        mV(0, 2); cW(1, 0); cE(1, 1); SLOT();
        mK(0, 2); cM(1, 1); SLOT();
      }
      if constexpr (DQ) {
        mQ(1, 1); if constexpr (MAIN) cA(1, 1); SLOT();
        mQ(1, 0); if constexpr (MAIN) cB(1, 1); SLOT();
        mQ(1, 2); if constexpr (MAIN) cC(1, 1); SLOT();
      } else if constexpr (MAIN) { cA(1, 1); cB(1, 1); cC(1, 1); SLOT(); }
      if constexpr (MAIN) {
        cP(1, 1); SLOT();
        cW(1, 1); mV(1, 0); SLOT();
        mK(1, 0); SLOT();
      }
      if constexpr (DQ) {
        float* pw = reinterpret_cast<float*>(smem + G::XP_OFF + (par ^ 1) * G::XPB + kq * G::PARTB);
#pragma unroll
        for (int r = 0; r < 4; ++r) pw[(4 * g + r) * DQS + 16 * ojq + i16] = __builtin_fmaf(accb[r], dnb, acca[r]);
        SLOT();
      }
      if constexpr (MAIN) {
        mV(1, 1); if constexpr (PRE) qA[0][1] = down(qA[0][0]); SLOT();
        mK(1, 1); if constexpr (PRE) qA[1][1] = down(qA[1][0]); SLOT();
        mV(1, 2); mK(1, 2); SLOT();
      }
      if constexpr (MAIN && DMA)
        if (copier) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's LDS-DMA runs of tile u + 1 have landed
      if (H2P_ABL & 32) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      else lds_barrier();
    };
    using T = std::true_type;
    using F = std::false_type;
    auto sweep = [&](auto first_tag) {
      int b0 = 0, b1 = BUFB, b2 = 2 * BUFB;            // buffers of tiles u, u + 1, u + 2
      auto rot = [&] { const int t = b0; b0 = b1; b1 = b2; b2 = t; };
      //        main dq   sum  copy u+2  rows u+1
      iteration(T{}, F{}, F{}, T{}, T{}, first_tag, 0, b0, b1, b2); rot();
      iteration(T{}, T{}, F{}, T{}, T{}, first_tag, 1, b0, b1, b2); rot();
      for (int u = 2; u < ntiles - 2; ++u) { iteration(T{}, T{}, T{}, T{}, T{}, first_tag, u, b0, b1, b2); rot(); }
      iteration(T{}, T{}, T{}, F{}, T{}, first_tag, ntiles - 2, b0, b1, b2); rot();
      iteration(T{}, T{}, T{}, F{}, F{}, first_tag, ntiles - 1, b0, b1, b2);
      iteration(F{}, T{}, T{}, F{}, F{}, first_tag, ntiles, 0, 0, 0);
      iteration(F{}, F{}, T{}, F{}, F{}, first_tag, ntiles + 1, 0, 0, 0);
    };
    if (kb == kb_begin) sweep(T{});
    else sweep(F{});

    // ---- dK, dV of this key block.  X = dK sqrt(d) 2^(sq + so + sv + 14 - DS_DOWN), Z = dV 2^(14 + so)
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      const int key = key0 + kt * 16 + i16;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int d = 4 * g + r;
        kout[(size_t)d * L + key] = __builtin_ldexpf(dKa[kt][r] * a.inv_sqrt_d, -(P_UP - DS_DOWN<D> + so + sv + sq));
        vout[(size_t)d * L + key] = __builtin_ldexpf(dVa[kt][r], -(P_UP + so));
      }
    }
  }
#undef SLOT
}

// dqkv[b][head * D + d][q] (Q third) = c_q x the sum over key ranges, in order, of the pipelined kernel's slabs [split][B][heads][L / 32][16][32]
__global__ void mha_dq_reduce_h2p_kernel(const float* __restrict__ part, const __bf16* __restrict__ pieces, float* __restrict__ dqkv, int nsplit,
                                         int C, int L, size_t split_stride) {
  constexpr int D = 16, TQ = GeoP::TQ, Q4 = TQ / 4;
  const int b = blockIdx.y, heads = C / D;
  const size_t per_sample = (size_t)C * L;
  const float* src = part + (size_t)b * per_sample;
  float* dst = dqkv + (size_t)b * 3 * per_sample;
  const size_t n4 = per_sample >> 2;
  const int tiles = L / TQ;
  const size_t piece_n = (size_t)L * D;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    f32x4 acc = reinterpret_cast<const f32x4*>(src)[i];
    for (int sp = 1; sp < nsplit; ++sp) acc += reinterpret_cast<const f32x4*>(src + (size_t)sp * split_stride)[i];
    const int q4 = (int)(i % Q4), d = (int)((i / Q4) % D);
    const size_t ht = i / ((size_t)Q4 * D);        // head * tiles + tile
    const size_t head = ht / tiles, tile = ht - head * tiles;
    const float* cq = reinterpret_cast<const float*>(pieces + (((size_t)b * heads + head) * S_COUNT + S_C) * piece_n);
    const f32x4 c = *reinterpret_cast<const f32x4*>(cq + tile * TQ + 4 * q4);
    reinterpret_cast<f32x4*>(dst + (head * D + d) * (size_t)L + tile * TQ)[q4] = acc * c;
  }
}

struct H2Geom { int nkb_total, per, nsplit; };
H2Geom h2_geometry(int B, int heads, int L, int D) {
  H2Geom g;
  g.nkb_total = L / KB;
  int want = cdiv(H2B_WANT, B * heads);                  // workgroups wanted in all: four rounds of one per CU on 256 CUs
  // At larger batches that leaves few key ranges per (sample, head) pair, and the workgroups of a pair are the ones that share
  // its Q / dO tile stream in their XCD's L2: up to 16 ranges per pair (batch 16: 567 -> 553 ms per launch; 32: 547) as
  // long as the slabs stay below the cap (16 GiB, or HDIFF_BWD_SLAB_GIB: mha_bwd_slab_cap_bytes).
  {
    const long long per_range = (long long)B * heads * D * L * 4;
    int cap = (int)(hdiff::mha_bwd_slab_cap_bytes() / per_range);
    int more = 16 < cap ? 16 : cap;
    if (more > want) want = more;
  }
  if (want > g.nkb_total) want = g.nkb_total;
  if (want < 1) want = 1;
  g.per = cdiv(g.nkb_total, want);
  g.nsplit = cdiv(g.nkb_total, g.per);
  return g;
}

}  // namespace

namespace hdiff {

// Upper bound on the dQ partial slabs (it sets how many key ranges a (sample, head) pair is cut into at large batches): 16 GiB, or
// HDIFF_BWD_SLAB_GIB gibibytes (1 ... 256; a deployment setting, read once per process; tests/test_gpu_backward.py).
long long mha_bwd_slab_cap_bytes() {
  static const long long cap = [] {
    const char* e = getenv("HDIFF_BWD_SLAB_GIB");
    long long g = e ? atoll(e) : 16;
    if (g < 1) g = 1;
    if (g > 256) g = 256;
    return g << 30;
  }();
  return cap;
}

// shapes the kernel covers: at least one 256-key block per CU (counted in 128-key units below, as the four-wave build did) -- below that (one sample at L <= 1024) the split pass and the
// slab reduce cost more than the matrix core gains (56 vs 47 us at B = 1, L = 1024; 0.62 vs 0.94 ms at B = 4, L = 4096)
bool mha_bwd_x3_shape_ok(int B, int C, int heads, int L) {
  const int D = C / heads;
  return C % heads == 0 && (D == 16 || D == 32) && L % 256 == 0 && L >= 512 && (int64_t)B * heads * (L / 128) >= 256;
}
bool mha_bwd_x3_applicable(int B, int C, int heads, int L) {
  return contraction_mode() == HDIFF_CONTRACT_BF16X3 && mha_bwd_x3_shape_ok(B, C, heads, L);
}

// slabs (tile-major, one per key range: even a single range goes through the reduce kernel, which restores the [C][L] layout)
// followed by the piece tensors (S_COUNT = 17 two-byte piece slots per element: the enum above) and the B * heads * 4 tensor maxima, in floats
int64_t mha_bwd_x3_workspace_floats(int B, int C, int heads, int L) {
  const H2Geom g = h2_geometry(B, heads, L, C / heads);
  const int64_t pieces_bytes = (int64_t)B * C * L * S_COUNT * 2;
  return (int64_t)g.nsplit * B * C * L + (pieces_bytes + 3) / 4 + 4 + (int64_t)B * heads * M_COUNT + 4;
}

// More than 64 KB of LDS per workgroup has to be asked for -- per DEVICE (the attribute belongs to the device's copy of the kernel): asked
// once per device of this process, remembered, and a refusal is reported to the caller BEFORE anything is launched.
static bool big_lds_granted() {
  constexpr int MAXDEV = 64;
  static std::atomic<signed char> state[MAXDEV];      // 0: not asked yet, 1: granted, -1: refused
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return false;
  signed char st = state[dev].load(std::memory_order_acquire);
  if (st == 0) {
    const bool ok =
        hipFuncSetAttribute(reinterpret_cast<const void*>(&mha_bwd_h2_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, Geo<16>::LDS_BYTES) == hipSuccess &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(&mha_bwd_h2_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, Geo<32>::LDS_BYTES) == hipSuccess &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(&mha_bwd_h2p_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, GeoP::LDS_BYTES) == hipSuccess;
    if (!ok) (void)hipGetLastError();
    st = ok ? 1 : -1;
    state[dev].store(st, std::memory_order_release);
  }
  return st > 0;
}

// delta has been computed by the caller (mha_delta_kernel).  false: the device refused the kernel's LDS size, nothing was launched
bool launch_mha_bwd_h2(const float* qkv, const float* d_o, const float* lse2, const float* delta, float* dqkv, float* ws,
                       int B, int C, int heads, int L, hipStream_t stream) {
  const int D = C / heads;
  const H2Geom g = h2_geometry(B, heads, L, D);
  const size_t per_sample = (size_t)C * L;
  const int64_t slab = (int64_t)g.nsplit * B * C * L;
  uintptr_t pw = reinterpret_cast<uintptr_t>(ws + slab);
  pw = (pw + 15) & ~(uintptr_t)15;
  __bf16* pieces = reinterpret_cast<__bf16*>(pw);
  unsigned* absmax = reinterpret_cast<unsigned*>(pieces + (size_t)B * C * L * S_COUNT);      // 4-byte aligned: the pieces are a multiple of 4 bytes
  BwdH2Args a;
  a.ws = pieces; a.lse2 = lse2; a.delta = delta; a.dqkv = dqkv;
  a.C = C; a.L = L; a.kb_per_split = g.per;
  a.inv_sqrt_d = 1.0f / sqrtf((float)D);
  a.qscale = 1.4426950408889634f * a.inv_sqrt_d;
  a.dq_part = ws; a.split_stride = (size_t)B * per_sample; a.batch_stride = per_sample;
  a.absmax = absmax; a.one = 1.0f;
  const float qscale = a.qscale;
  const dim3 mgrid(cdiv(L, 4096), M_COUNT * heads, B), sgrid(cdiv(L, THREADS), 4 * heads, B), grid(g.nsplit, heads, B);
  const size_t n4 = per_sample / 4;
  const int bx = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
  if (!big_lds_granted()) return false;      // nothing has been enqueued yet: the caller runs the fp32-input backward instead
  (void)hipMemsetAsync(absmax, 0, (size_t)B * heads * M_COUNT * sizeof(unsigned), stream);
  if (D == 16) {
    hipLaunchKernelGGL(mha_bwd_absmax_kernel<16>, mgrid, dim3(THREADS), 0, stream, qkv, d_o, absmax, C, L);
    hipLaunchKernelGGL(mha_bwd_split_h2_kernel<16>, sgrid, dim3(THREADS), 0, stream, qkv, d_o, pieces, absmax, lse2, delta, C, L, qscale, 1.0f);
    if (H2B_PIPE && MW == 8) {
      hipLaunchKernelGGL(mha_bwd_h2p_kernel, grid, dim3(MTHREADS), GeoP::LDS_BYTES, stream, a);
      hipLaunchKernelGGL(mha_dq_reduce_h2p_kernel, dim3(bx, B), dim3(256), 0, stream, ws, pieces, dqkv, g.nsplit, C, L, a.split_stride);
    } else {
      hipLaunchKernelGGL(mha_bwd_h2_kernel<16>, grid, dim3(MTHREADS), Geo<16>::LDS_BYTES, stream, a);
      hipLaunchKernelGGL(mha_dq_reduce_h2_kernel<16>, dim3(bx, B), dim3(256), 0, stream, ws, dqkv, g.nsplit, C, L, a.split_stride);
    }
  } else {
    hipLaunchKernelGGL(mha_bwd_absmax_kernel<32>, mgrid, dim3(THREADS), 0, stream, qkv, d_o, absmax, C, L);
    hipLaunchKernelGGL(mha_bwd_split_h2_kernel<32>, sgrid, dim3(THREADS), 0, stream, qkv, d_o, pieces, absmax, lse2, delta, C, L, qscale, 1.0f);
    hipLaunchKernelGGL(mha_bwd_h2_kernel<32>, grid, dim3(MTHREADS), Geo<32>::LDS_BYTES, stream, a);
    hipLaunchKernelGGL(mha_dq_reduce_h2_kernel<32>, dim3(bx, B), dim3(256), 0, stream, ws, dqkv, g.nsplit, C, L, a.split_stride);
  }
  return true;
}

}  // namespace hdiff
