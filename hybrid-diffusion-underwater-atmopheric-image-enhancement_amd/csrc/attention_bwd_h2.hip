// Backward of the flash self-attention core at d_head 16 and 32 in the split-operand mode.  Round 3 built it on bf16 triples
// throughout (attention_bwd_x3.hip, retired in round 5: git show 6a5a4b1 has it); round 4 moved everything that touches
// P = exp2(S - lse) or the pair (dO, V) to fp16 PAIRS
// (attention_h2.hip has the argument and the instructions: v_cvt_pk_f16_f32 + v_fma_mixlo/hi_f16, 1.5 per value against 5.5).
// Same contract, layouts and dQ slab protocol (reference: autograd through nn.MultiheadAttention, ModelCondition.py:189,
// 204-208, TrainCondition.py:60): bitwise reproducible.
//
//   S   = Q K^T - lse2 + 14      bf16 triples, six products (an error in S is an error in an exponent: stays as it was)
//   dP' = dO' V'^T - delta'      dO' = dO 2^so, V' = V 2^sv as fp16 pairs, FOUR products (a pair holds 22-23 bits, the
//                                fp32-class sum of products needs both cross terms; tools/h2_sim_qk.py); so, sv: powers of
//                                two per (sample, head) that put the tensor's maximum in [2^14, 2^15) (mha_bwd_absmax_kernel);
//                                delta' = delta 2^(so + sv) is scaled when the tile is staged
//   P'  = exp2(S)                = P 2^14 <= 2^14: always inside fp16, no reference to move; two fp16 pieces
//   dS' = P' o dP'               = dS 2^(14 + so + sv), fp32; its magnitude is unbounded both ways -- bf16 triples as before
//   dV'^T += dO'^T P'            three products (o0 p0, o1 p0, o0 p1) instead of six
//   dK'^T += Q^T dS' ,  dQ'^T += K^T dS'^T    unchanged; the powers of two leave with the final scaling of each output
// Per 16x16 (query, key) tile at d 16: 12.5 MFMAs and ~36 vector instructions per lane against 15 and ~52.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

using namespace hdiff;

namespace {

#ifndef H2B_ABL
#define H2B_ABL 0            // dev: timing ablations (bit mask), results are wrong with any bit set
#endif
constexpr int THREADS = 256;
constexpr int KB = 128;                // keys per workgroup block (32 per wave)
constexpr int SROW = 72;               // bytes per key row of the dS image [key][32 queries] (64 + 8: conflict-free ds_write_b64)
constexpr int SPART = 32 * SROW;       // 2304 per piece
constexpr int SCRB = 3 * SPART;        // 6912 per wave

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
  static constexpr int QA_OFF = 0, OA_OFF = 3 * RPART;           // three bf16 pieces of Q rows, two fp16 pieces of dO rows
  static constexpr int SL_OFF = 5 * RPART, SD_OFF = SL_OFF + TQ * 4, BUFB = SD_OFF + TQ * 4;
  static constexpr int DQS = TQ + 4;                  // row stride (floats) of a wave's dQ partial tile [D][DQS], aliased on its scratch
  static_assert(TQ * GROW == 2048, "staging geometry: 128 chunks per piece");
  static_assert(D * DQS * 4 <= SCRB, "dQ partial tile must fit in the wave's scratch");
  static_assert(2 * BUFB + 4 * SCRB <= 65536, "static LDS");
};

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
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

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for the wave's outstanding GLOBAL stores
// (release fence: s_waitcnt vmcnt(0)); with one slab store per tile in flight that wait exposed the store's latency at
// every barrier -- 23 ms of a 158 ms launch (timing ablation).  Nothing here hands global data to another wave.
__device__ __forceinline__ void lds_barrier() {
  if (H2B_ABL & 256) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ f32x4 mfma_bf16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
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
// power-of-two scale that puts a tensor's maximum |x| (given as the bits of the fp32 value) into [2^14, 2^15)
__device__ __forceinline__ int scale_exp(unsigned amax_bits) {
  int e = (int)((amax_bits >> 23) & 0xffu) - 127;
  e = e < -50 ? -50 : (e > 50 ? 50 : e);       // all-zero / denormal / huge tensors: a fixed scale (inf and NaN stay what they are)
  return 14 - e;
}

// split-product terms kept (piece of the A-side tensor, piece of the B-side tensor): all i + j <= 2, small terms last
__device__ constexpr int TERM_A[6] = {0, 1, 0, 2, 1, 0};
__device__ constexpr int TERM_B[6] = {0, 0, 1, 0, 1, 2};

// piece tensors of one (sample, head), each 3 pieces of L * D bf16
enum { T_QA = 0, T_KB = 1, T_KT = 2, T_VB = 3, T_OA = 4, T_COUNT = 5 };

// ---------------------------------------------------------------------------------------------------------------------
// max |V| and max |dO| per (sample, head) as fp32 bit patterns (non-negative floats order like unsigned integers):
// absmax[(b * heads + head) * 2 + {0: V, 1: dO}], zeroed by the launcher.  grid (L / 4096, 2 * heads, B).
// ---------------------------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(THREADS) void mha_bwd_absmax_kernel(const float* __restrict__ qkv, const float* __restrict__ d_o,
                                                                 unsigned* __restrict__ absmax, int C, int L) {
  const int heads = C / D;
  const int which = blockIdx.y / heads, head = blockIdx.y - which * heads, b = blockIdx.z;
  const float* src = (which == 0) ? qkv + ((size_t)b * 3 * C + 2 * (size_t)C + (size_t)head * D) * L
                                  : d_o + ((size_t)b * C + (size_t)head * D) * L;
  const int l0 = blockIdx.x * 4096;
  float m = 0.f;
  for (int d = 0; d < D; ++d)
    for (int i = threadIdx.x * 4; i < 4096 && l0 + i < L; i += THREADS * 4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(src + (size_t)d * L + l0 + i);
      m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(absmax + ((size_t)b * heads + head) * 2 + which, __builtin_bit_cast(unsigned, m));
}

// ---------------------------------------------------------------------------------------------------------------------
// fp32 qkv [B][3C][L] and dO [B][C][L] -> the five piece tensors.  grid (L / 256, 4 * heads, B): blockIdx.y / heads
// selects Q, K, V or dO.  Q (pre-scaled), K: bf16 triples as rows [L][D], K also transposed [D][L].  V, dO: fp16 pairs of
// the tensor times its power of two (scale_exp of the maxima above), rows [L][D], in piece slots 0 and 1.
// ---------------------------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(THREADS) void mha_bwd_split_h2_kernel(const float* __restrict__ qkv, const float* __restrict__ d_o,
                                                                   __bf16* __restrict__ ws, const unsigned* __restrict__ absmax,
                                                                   int C, int L, float qscale, float one) {
  const int heads = C / D;
  const int which = blockIdx.y / heads, head = blockIdx.y - which * heads, b = blockIdx.z;
  const float* src = (which < 3) ? qkv + ((size_t)b * 3 * C + (size_t)which * C + (size_t)head * D) * L
                                 : d_o + ((size_t)b * C + (size_t)head * D) * L;
  const size_t piece = (size_t)L * D;
  __bf16* base = ws + ((size_t)b * heads + head) * (T_COUNT * 3) * piece;
  const int t_rows = which == 0 ? T_QA : which == 1 ? T_KB : which == 2 ? T_VB : T_OA;
  const int t_tr = which == 1 ? T_KT : -1;
  const int l = blockIdx.x * THREADS + threadIdx.x;
  if (which >= 2) {
    if (l >= L) return;
    const float sc = __builtin_ldexpf(1.0f, scale_exp(absmax[((size_t)b * heads + head) * 2 + (which - 2)]));
    unsigned h[2][D / 2];
#pragma unroll
    for (int j = 0; j < D / 2; ++j) split2(src[(size_t)(2 * j) * L + l] * sc, src[(size_t)(2 * j + 1) * L + l] * sc, one, h[0][j], h[1][j]);
    __bf16* dst = base + (size_t)t_rows * 3 * piece;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      u32x4* o = reinterpret_cast<u32x4*>(dst + p * piece + (size_t)l * D);
#pragma unroll
      for (int j = 0; j < D / 8; ++j) o[j] = u32x4{h[p][4 * j], h[p][4 * j + 1], h[p][4 * j + 2], h[p][4 * j + 3]};
    }
    return;
  }
  const float sc = which == 0 ? qscale : 1.0f;
  {
    if (l < L) {
      unsigned h[3][D / 2];
#pragma unroll
      for (int j = 0; j < D / 2; ++j) {
        const float a = src[(size_t)(2 * j) * L + l] * sc, c = src[(size_t)(2 * j + 1) * L + l] * sc;
        split3(a, c, h[0][j], h[1][j], h[2][j]);
      }
      __bf16* dst = base + (size_t)t_rows * 3 * piece;
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        u32x4* o = reinterpret_cast<u32x4*>(dst + p * piece + (size_t)l * D);
#pragma unroll
        for (int j = 0; j < D / 8; ++j) o[j] = u32x4{h[p][4 * j], h[p][4 * j + 1], h[p][4 * j + 2], h[p][4 * j + 3]};
      }
    }
  }
  if (t_tr >= 0) {
    __bf16* dst = base + (size_t)t_tr * 3 * piece;
    const int l2 = (int)blockIdx.x * (THREADS / 2) + (int)threadIdx.x;       // the block's 256 positions = 128 pairs: half the threads
    if ((int)threadIdx.x < THREADS / 2 && 2 * l2 < L) {
#pragma unroll 4
      for (int d = 0; d < D; ++d) {
        const f32x2 v = *reinterpret_cast<const f32x2*>(src + (size_t)d * L + 2 * l2);
        unsigned h0, h1, h2;
        split3(v[0] * sc, v[1] * sc, h0, h1, h2);
        unsigned* o = reinterpret_cast<unsigned*>(dst + (size_t)d * L) + l2;
        o[0] = h0;
        o[piece / 2] = h1;
        o[piece] = h2;
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
  float inv_sqrt_d;
  const unsigned* absmax;     // max |V|, max |dO| per (sample, head): mha_bwd_absmax_kernel
  float one;                  // 1.0f, opaque to the compiler (split2)
};

template <int D>
__global__ __launch_bounds__(THREADS, 2) void mha_bwd_h2_kernel(const BwdH2Args a) {
  using G = Geo<D>;
  constexpr int TQ = G::TQ, NSUB = G::NSUB, MT = G::MT, RROW = G::RROW, GROW = G::GROW, CPR = G::CPR, RPART = G::RPART, BUFB = G::BUFB, DQS = G::DQS;
  constexpr int QA_OFF = G::QA_OFF, OA_OFF = G::OA_OFF, SL_OFF = G::SL_OFF, SD_OFF = G::SD_OFF;
  constexpr int TPM = 32 / D;                  // terms per d-contracted MFMA: d 16 packs two piece products along the 32 slots
  constexpr int NQK = 6 / TPM;                 // MFMAs of one 16x16 score tile
  constexpr int NOP = 3;                       // Q / K: operand registers per 16 rows of a d-contracted operand (sets at d 16, pieces at d 32)
  constexpr int NOA = (D == 16) ? 1 : 2;       // dO: [o0 | o1] along the 32 slots at d 16, the two pieces at d 32
  constexpr int NVB = 2;                       // V: [v0 | v0], [v1 | v1] at d 16, the two pieces at d 32
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BUFB + 4 * SCRB];

  const int C = a.C, L = a.L;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  const TileId tile = xcd_tile();
  const int head = tile.head, b = tile.b, heads = gridDim.y, split = tile.x;
  const size_t piece_n = (size_t)L * D;
  const __bf16* wsh = a.ws + ((size_t)b * heads + head) * (T_COUNT * 3) * piece_n;
  const float* lbase = a.lse2 + ((size_t)b * heads + head) * L;
  const float* dbase = a.delta + ((size_t)b * heads + head) * L;
  float* part = a.dq_part + (size_t)split * a.split_stride + (size_t)b * a.batch_stride + (size_t)head * D * L;
  float* kout = a.dqkv + ((size_t)b * 3 * C + (size_t)C + (size_t)head * D) * L;
  float* vout = kout + (size_t)C * L;
  const int ntiles = L / TQ;
  const int nkb_total = L / KB;
  const int kb_begin = split * a.kb_per_split;
  const int kb_end = (kb_begin + a.kb_per_split < nkb_total) ? kb_begin + a.kb_per_split : nkb_total;
  // the powers of two of this (sample, head): dO' = dO 2^so, V' = V 2^sv
  const int so = scale_exp(a.absmax[((size_t)b * heads + head) * 2 + 1]), sv = scale_exp(a.absmax[((size_t)b * heads + head) * 2]);
  const float dscale = __builtin_ldexpf(1.0f, so + sv);      // delta' = delta 2^(so + sv)
  const float one = a.one;

  // contraction slots of this lane in the d-contracted products.  d 16: 8 consecutive d of one of the MFMA's two terms;
  // d 32: 8 consecutive d of the one term
  const int doff = (D == 16) ? 8 * (g & 1) : 8 * g;
  const bool hi = (D == 16) && (g >> 1);

  // ---- staging of one query tile: 640 chunks of 16 bytes (three row pieces of Q, two of dO), three per thread -- the spare
  // half round repeats dO chunks; 14 - lse2 and -delta' by the first 2 TQ threads
  unsigned goff[3];
  int lds_off[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    int c = i * THREADS + tid;
    if (c >= 640) c -= 128;
    const int sel = c / 384, cv = c - sel * 384;           // 0: Q rows, 1: dO rows
    const int p = cv >> 7, rem = cv & 127;
    const int row = rem / CPR, ch = rem % CPR;
    goff[i] = (unsigned)(((sel == 0 ? T_QA : T_OA) * 3 + p) * piece_n * 2) + row * GROW + ch * 16;
    lds_off[i] = (sel == 0 ? QA_OFF : OA_OFF) + p * RPART + row * RROW + ch * 16;
  }
  const unsigned char* wsb = reinterpret_cast<const unsigned char*>(wsh);
  const float* ldsrc = (tid < TQ) ? lbase + tid : dbase + (tid & (TQ - 1));     // used by the first 2 TQ threads only
  u32x4 stage[3];
  float stage_ld = 0.f;
  auto stage_load = [&](int t) {
#pragma unroll
    for (int i = 0; i < 3; ++i) stage[i] = *reinterpret_cast<const u32x4*>(wsb + goff[i] + (size_t)t * (TQ * GROW));
    if (tid < 2 * TQ) stage_ld = ldsrc[t * TQ];      // negated when stored: nothing here may consume a load at once
  };
  auto stage_store = [&](int buf) {
    unsigned char* tb = smem + buf * BUFB;
#pragma unroll
    for (int i = 0; i < 3; ++i) *reinterpret_cast<u32x4*>(tb + lds_off[i]) = stage[i];
    if (tid < 2 * TQ) *reinterpret_cast<float*>(tb + SL_OFF + tid * 4) = (tid < TQ) ? 14.0f - stage_ld : -stage_ld * dscale;    // sL then sD, contiguous
  };

  // operand addresses inside a tile buffer
  int a1addr[NOP];      // row reads: (d 16) the term's piece of this lane half / (d 32) piece j; row i16, 16 bytes at doff
#pragma unroll
  for (int j = 0; j < NOP; ++j)
    a1addr[j] = ((D == 16) ? (hi ? TERM_B[2 * j + 1] : TERM_B[2 * j]) : j) * RPART + i16 * RROW + doff * 2;
  int o1addr[NOA];      // dO rows: (d 16) piece 0 in the low half of the contraction, piece 1 in the high half / (d 32) piece j
#pragma unroll
  for (int j = 0; j < NOA; ++j) o1addr[j] = ((D == 16) ? (hi ? 1 : 0) : j) * RPART + i16 * RROW + doff * 2;
  // transposed reads of the same tiles (A operands of the products that sum over queries): lane 4q + p of a 16-lane group
  // addresses row 4g + q (then 16 + 4g + q), columns d = 16 mt + 4p .. + 3
  const int a3addr = (4 * g + (i16 >> 2)) * RROW + 8 * (i16 & 3);
  unsigned char* scr = smem + 2 * BUFB + wave * SCRB;
  float* sdq = reinterpret_cast<float*>(scr);
  // the dS image of this wave: [piece][key 0..31][SROW bytes of 32 queries]
  const int swaddr = i16 * SROW + 8 * g;                              // + piece * SPART + kt * 16 * SROW + jq * 32
  const int sraddr = (8 * g + (i16 >> 2)) * SROW + 8 * (i16 & 3);     // + piece * SPART + jq * 32 (+ 4 * SROW: second half)
  // reduction of the tile's dQ over the four waves: thread = floats tid, 256 + tid, 512 + tid, 768 + tid of the tile's
  // [D][TQ] block (1024 floats either way) -- every slab instruction of a wave then covers 256 contiguous bytes (with four
  // NEIGHBOURING floats per thread the four L2 adds of a wave hit the same eight lines back to back: 95 ms of a 225 ms launch)

  for (int kb = kb_begin; kb < kb_end; ++kb) {
    const int key0 = kb * KB + wave * 32;
    // ---- stationary operands of the wave's 32 keys, straight from the workspace
    u32x4 kB[2][NOP], vB[2][NVB], kT[MT][3];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int j = 0; j < NOP; ++j) {
        const int p = (D == 16) ? (hi ? TERM_A[2 * j + 1] : TERM_A[2 * j]) : j;
        const size_t off = (size_t)p * piece_n + (size_t)(key0 + kt * 16 + i16) * D + doff;
        kB[kt][j] = *reinterpret_cast<const u32x4*>(wsh + (size_t)T_KB * 3 * piece_n + off);
      }
#pragma unroll
      for (int j = 0; j < NVB; ++j)       // piece j of V' for every lane: at d 16 both halves of the contraction carry it
        vB[kt][j] = *reinterpret_cast<const u32x4*>(wsh + (size_t)T_VB * 3 * piece_n + (size_t)j * piece_n +
                                                    (size_t)(key0 + kt * 16 + i16) * D + doff);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int p = 0; p < 3; ++p)
        kT[mt][p] = *reinterpret_cast<const u32x4*>(wsh + (size_t)(T_KT * 3 + p) * piece_n + (size_t)(16 * mt + i16) * L + key0 + 8 * g);
    f32x4 dKt[2][MT], dVt[2][MT];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) { dKt[kt][mt] = f32x4{0.f, 0.f, 0.f, 0.f}; dVt[kt][mt] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    __syncthreads();              // the previous key block's last tile is fully consumed
    stage_load(0);
    stage_store(0);
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
      float* pdst = part + (size_t)t * (D * TQ) + tid;

      f32x4 dQt[NSUB][2][MT];
#pragma unroll
      for (int sub = 0; sub < NSUB; ++sub) {
        const unsigned char* sb = tb + sub * 32 * RROW;
        u32x4 qA[2][NOP], oA[2][NOA];
#pragma unroll
        for (int jq = 0; jq < 2; ++jq) {
#pragma unroll
          for (int j = 0; j < NOP; ++j) qA[jq][j] = *reinterpret_cast<const u32x4*>(sb + QA_OFF + a1addr[j] + jq * 16 * RROW);
#pragma unroll
          for (int j = 0; j < NOA; ++j) oA[jq][j] = *reinterpret_cast<const u32x4*>(sb + OA_OFF + o1addr[j] + jq * 16 * RROW);
        }
        f32x4 negl[2], negd[2];
#pragma unroll
        for (int jq = 0; jq < 2; ++jq) {
          negl[jq] = *reinterpret_cast<const f32x4*>(tb + SL_OFF + (32 * sub + 16 * jq + 4 * g) * 4);
          negd[jq] = *reinterpret_cast<const f32x4*>(tb + SD_OFF + (32 * sub + 16 * jq + 4 * g) * 4);
        }
        // A operands of the q-summed products for M tile mt: dO^T / Q^T rows d = 16 mt .., 32 queries along the contraction
        auto load_transposed = [&](int mt, u32x4 (&qT)[3], u32x4 (&oT)[2]) {
#pragma unroll
          for (int p = 0; p < 3; ++p) {
            const unsigned char* sq = sb + QA_OFF + p * RPART + a3addr + 32 * mt;
            const u32x2 q0 = lds_read_tr16(sq), q1 = lds_read_tr16(sq + 16 * RROW);
            qT[p] = u32x4{q0[0], q0[1], q1[0], q1[1]};
            if (p < 2) {
              const unsigned char* sop = sb + OA_OFF + p * RPART + a3addr + 32 * mt;
              const u32x2 o0 = lds_read_tr16(sop), o1 = lds_read_tr16(sop + 16 * RROW);
              oT[p] = u32x4{o0[0], o0[1], o1[0], o1[1]};
            }
          }
        };
        u32x4 qT0[3], oT0[2];
        if (MT == 1) load_transposed(0, qT0, oT0);       // d 16: read once per subtile, used by both key tiles

#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
          f32x4 S[2], dP[2];
#pragma unroll
          for (int jq = 0; jq < 2; ++jq) {
            f32x4 acc = negl[jq];
            if constexpr (D == 16) {
#pragma unroll
              for (int j = 0; j < 3; ++j) acc = mfma_bf16(qA[jq][j], kB[kt][j], acc);
            } else {
#pragma unroll
              for (int term = 5; term >= 0; --term) acc = mfma_bf16(qA[jq][TERM_B[term]], kB[kt][TERM_A[term]], acc);
            }
            S[jq] = acc;
            acc = negd[jq];
            // (mutation test, bit 16: the cross product o0 v1 dropped -- at d 16 with its MFMA partner o1 v1, a 2^-22 term)
            if constexpr (D == 16) {        // (o0 v1 + o1 v1), then (o0 v0 + o1 v0): the small products first
              if (!(HDIFF_MUTANT & 16)) acc = mfma_f16(oA[jq][0], vB[kt][1], acc);
              acc = mfma_f16(oA[jq][0], vB[kt][0], acc);
            } else {
              acc = mfma_f16(oA[jq][1], vB[kt][1], acc);
              if (!(HDIFF_MUTANT & 16)) acc = mfma_f16(oA[jq][0], vB[kt][1], acc);
              acc = mfma_f16(oA[jq][1], vB[kt][0], acc);
              acc = mfma_f16(oA[jq][0], vB[kt][0], acc);
            }
            dP[jq] = acc;
          }
          u32x4 Pp[2], Sp[3];
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
                for (int pc = 0; pc < 3; ++pc) {
                  if (pc < 2) Pp[pc][2 * jq + h] = __builtin_bit_cast(unsigned, S[jq][2 * h]) + pc;
                  Sp[pc][2 * jq + h] = __builtin_bit_cast(unsigned, dP[jq][2 * h + 1]) + pc;
                }
            } else {
#pragma unroll
              for (int h = 0; h < 2; ++h) {
                unsigned h0, h1, h2;
                split2(p[2 * h], p[2 * h + 1], one, h0, h1);
                Pp[0][2 * jq + h] = h0; Pp[1][2 * jq + h] = h1;
                split3(ds[2 * h], ds[2 * h + 1], h0, h1, h2);
                Sp[0][2 * jq + h] = h0; Sp[1][2 * jq + h] = h1; Sp[2][2 * jq + h] = h2;
              }
            }
            // the packed dS pieces of (key i16 of tile kt, queries 16 jq + 4g ..+3) into the wave's [key][query] image
            if (!(H2B_ABL & 4)) {
#pragma unroll
              for (int pc = 0; pc < 3; ++pc)
                *reinterpret_cast<u32x2*>(scr + pc * SPART + kt * 16 * SROW + jq * 32 + swaddr) = u32x2{Sp[pc][2 * jq], Sp[pc][2 * jq + 1]};
            }
          }
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            u32x4 qTm[3], oTm[2];
            if (MT > 1) load_transposed(mt, qTm, oTm);       // d 32: one M tile's operands at a time (registers)
#pragma unroll
            for (int term = 5; term >= 0; --term) {      // small terms first
              // (mutation test, bit 32: the product o1 p0 of dV^T dropped; bit 64: the 2^-16 term q0 s2 of dK^T -- and k0 s2 of dQ^T below)
              if (term < 3 && !((HDIFF_MUTANT & 32) && term == 1))
                dVt[kt][mt] = mfma_f16((MT == 1 ? oT0 : oTm)[TERM_A[term]], Pp[TERM_B[term]], dVt[kt][mt]);   // o0 p1, o1 p0, o0 p0
              if (!((HDIFF_MUTANT & 64) && term == 5))
                dKt[kt][mt] = mfma_bf16((MT == 1 ? qT0 : qTm)[TERM_A[term]], Sp[TERM_B[term]], dKt[kt][mt]);
            }
          }
        }

        // ---- dQ^T of the subtile over this wave's 32 keys: the dS image read back transposed, keys along the contraction
        asm volatile("" ::: "memory");
#pragma unroll
        for (int jq = 0; jq < 2; ++jq) {
          u32x4 sT[3];
#pragma unroll
          for (int p = 0; p < 3; ++p) {
            const unsigned char* src = scr + p * SPART + jq * 32 + sraddr;
            const u32x2 lo = lds_read_tr16(src), hi2 = lds_read_tr16(src + 4 * SROW);
            sT[p] = u32x4{lo[0], lo[1], hi2[0], hi2[1]};
          }
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int term = 5; term >= 0; --term)
              if (!((HDIFF_MUTANT & 64) && term == 5)) acc = mfma_bf16(kT[mt][TERM_A[term]], sT[TERM_B[term]], acc);
            dQt[sub][jq][mt] = acc;
          }
        }
        asm volatile("" ::: "memory");      // the next subtile's image stores stay behind these reads (same wave: in order)
      }

      // the wave's partial tile [d][query] over its own scratch (its reads above are done: same wave, in order)
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
        // thread -> floats tid + 256 r of the [D][TQ] block: row d = e / TQ, query e % TQ
        f32x4 sum;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int e = tid + 256 * r;
          const float* s0 = reinterpret_cast<const float*>(smem + 2 * BUFB) + (e / TQ) * DQS + (e % TQ);
          float v = s0[0];
#pragma unroll
          for (int w = 1; w < 4; ++w) v += s0[w * (SCRB / 4)];
          sum[r] = __builtin_ldexpf(v * a.inv_sqrt_d, -(14 + so + sv));
        }
        // tile t + 1 into LDS, then the loads of tile t + 2
        stage_store(buf ^ 1);
        if (!(H2B_ABL & 8)) stage_load(t + 2 < ntiles ? t + 2 : t);
        if (!(H2B_ABL & 65) || t == 0) {
          if (first_kb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) pdst[256 * r] = sum[r];
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) unsafeAtomicAdd(pdst + 256 * r, sum[r]);
          }
        }
      }
      lds_barrier();
    }

    // ---- dK, dV of this key block (complete: the sweep covered every query).  dK carries Q pre-scaled by log2(e)/sqrt(d)
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      const int key = key0 + kt * 16 + i16;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int d = 16 * mt + 4 * g + r;
          kout[(size_t)d * L + key] = __builtin_ldexpf(dKt[kt][mt][r] * 0.6931471805599453f, -(14 + so + sv));
          vout[(size_t)d * L + key] = __builtin_ldexpf(dVt[kt][mt][r], -(14 + so));
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

struct H2Geom { int nkb_total, per, nsplit; };
H2Geom h2_geometry(int B, int heads, int L, int D) {
  H2Geom g;
  g.nkb_total = L / KB;
  int want = cdiv(1024, B * heads);                      // ~2 rounds of 2 workgroups per CU on 256 CUs
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

// shapes the kernel covers: at least one 128-key block per CU -- below that (one sample at L <= 1024) the split pass and the
// slab reduce cost more than the matrix core gains (56 vs 47 us at B = 1, L = 1024; 0.62 vs 0.94 ms at B = 4, L = 4096)
bool mha_bwd_x3_shape_ok(int B, int C, int heads, int L) {
  const int D = C / heads;
  return C % heads == 0 && (D == 16 || D == 32) && L % 256 == 0 && L >= 512 && (int64_t)B * heads * (L / KB) >= 256;
}
bool mha_bwd_x3_applicable(int B, int C, int heads, int L) {
  return contraction_mode() == HDIFF_CONTRACT_BF16X3 && mha_bwd_x3_shape_ok(B, C, heads, L);
}

// slabs (tile-major, one per key range: even a single range goes through the reduce kernel, which restores the [C][L] layout)
// followed by the piece tensors (fifteen 2-byte piece slots per element: the layout round 3's bf16-triple kernel introduced; the V
// and dO tensors fill two of their three) and the B * heads * 2 tensor maxima, in floats
int64_t mha_bwd_x3_workspace_floats(int B, int C, int heads, int L) {
  const H2Geom g = h2_geometry(B, heads, L, C / heads);
  const int64_t pieces_bytes = (int64_t)B * C * L * (T_COUNT * 3) * 2;
  return (int64_t)g.nsplit * B * C * L + (pieces_bytes + 3) / 4 + 4 + (int64_t)B * heads * 2 + 4;
}

// delta has been computed by the caller (mha_delta_kernel)
void launch_mha_bwd_h2(const float* qkv, const float* d_o, const float* lse2, const float* delta, float* dqkv, float* ws,
                       int B, int C, int heads, int L, hipStream_t stream) {
  const int D = C / heads;
  const H2Geom g = h2_geometry(B, heads, L, D);
  const size_t per_sample = (size_t)C * L;
  const int64_t slab = (int64_t)g.nsplit * B * C * L;
  uintptr_t pw = reinterpret_cast<uintptr_t>(ws + slab);
  pw = (pw + 15) & ~(uintptr_t)15;
  __bf16* pieces = reinterpret_cast<__bf16*>(pw);
  unsigned* absmax = reinterpret_cast<unsigned*>(pieces + (size_t)B * C * L * (T_COUNT * 3));      // 4-byte aligned: the pieces are a multiple of 4 bytes
  BwdH2Args a;
  a.ws = pieces; a.lse2 = lse2; a.delta = delta; a.dqkv = dqkv;
  a.C = C; a.L = L; a.kb_per_split = g.per;
  a.inv_sqrt_d = 1.0f / sqrtf((float)D);
  a.dq_part = ws; a.split_stride = (size_t)B * per_sample; a.batch_stride = per_sample;
  a.absmax = absmax; a.one = 1.0f;
  const float qscale = 1.4426950408889634f * a.inv_sqrt_d;
  const dim3 mgrid(cdiv(L, 4096), 2 * heads, B), sgrid(cdiv(L, THREADS), 4 * heads, B), grid(g.nsplit, heads, B);
  const size_t n4 = per_sample / 4;
  const int bx = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
  (void)hipMemsetAsync(absmax, 0, (size_t)B * heads * 2 * sizeof(unsigned), stream);
  if (D == 16) {
    hipLaunchKernelGGL(mha_bwd_absmax_kernel<16>, mgrid, dim3(THREADS), 0, stream, qkv, d_o, absmax, C, L);
    hipLaunchKernelGGL(mha_bwd_split_h2_kernel<16>, sgrid, dim3(THREADS), 0, stream, qkv, d_o, pieces, absmax, C, L, qscale, 1.0f);
    hipLaunchKernelGGL(mha_bwd_h2_kernel<16>, grid, dim3(THREADS), 0, stream, a);
    hipLaunchKernelGGL(mha_dq_reduce_h2_kernel<16>, dim3(bx, B), dim3(256), 0, stream, ws, dqkv, g.nsplit, C, L, a.split_stride);
  } else {
    hipLaunchKernelGGL(mha_bwd_absmax_kernel<32>, mgrid, dim3(THREADS), 0, stream, qkv, d_o, absmax, C, L);
    hipLaunchKernelGGL(mha_bwd_split_h2_kernel<32>, sgrid, dim3(THREADS), 0, stream, qkv, d_o, pieces, absmax, C, L, qscale, 1.0f);
    hipLaunchKernelGGL(mha_bwd_h2_kernel<32>, grid, dim3(THREADS), 0, stream, a);
    hipLaunchKernelGGL(mha_dq_reduce_h2_kernel<32>, dim3(bx, B), dim3(256), 0, stream, ws, dqkv, g.nsplit, C, L, a.split_stride);
  }
}

}  // namespace hdiff
