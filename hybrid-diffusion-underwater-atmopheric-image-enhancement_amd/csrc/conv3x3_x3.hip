// 3x3 / stride-1 convolution with fp32 operands carried as three bf16 pieces on the bf16 matrix core
// (v_mfma_f32_32x32x16_bf16, fp32 accumulation) -- the HDIFF_CONTRACT_BF16X3 counterpart of conv_igemm.hip's fast path
// (reference call sites: ModelCondition.py:172, 186, 88 / diffusion/Model.py:275, 288, 184).
//
// Arithmetic: x = x0 + x1 + x2 exactly (top-16-bit truncations of successive exact remainders, attention_x3.hip), and
// w * x = sum over the six piece pairs with i + j <= 2 (dropped terms <= 3 * 2^-24 relative): fp32-class, checked against
// float64 next to the fp32-MFMA kernel in tests/test_gpu_ops.py.  Why it pays more here than in attention: both operands
// are split ONCE per staged element (the weights even once per model, by hdiff_pack_conv_weight_x3) and then used by 9 taps
// x 64 channels, so the split is noise and the kernel gets the 6/16 matrix-time ratio.
//
// Structure per workgroup (64 output channels x 8x32 pixels, 4 waves x 64 pixels), per chunk of 16 input channels:
//   * the (8+2) x (32+2) activation patch: global -> registers (prefetched one chunk ahead) -> GroupNorm affine + Swish ->
//     three bf16 pieces -> LDS as [piece][pixel][16 channels]: a lane's B operand (8 consecutive channels of one pixel) is one
//     16-byte LDS read, and a tap is just a pixel offset;
//   * the weights are not staged: hdiff_pack_conv_weight_x3 lays them out as [chunk][tap][piece][channel][16 ci], so a lane's
//     A operand is one 16-byte global load (L1/L2 resident: every workgroup of the launch reads the same 55 KB per chunk);
//   * per tap 6 piece pairs x (2 x 2) tiles = 24 MFMAs; the next tap's weight loads are issued before them.
//
// PAIR (round 4): the same kernel with both operands as fp16 PAIRS instead of bf16 triples (v_mfma_f32_32x32x16_f16), for
// convolutions with the GroupNorm + Swish prologue.  fp16 carries 11 significand bits, a pair x' = h0 + h1 holds 22-23 of
// fp32's 24 (attention_h2.hip has the split: v_cvt_pk_f16_f32 + v_fma_mixlo / mixhi_f16, 1.5 instructions per value against
// 5.5), and THREE products (w1 x0, w0 x1, w0 x0; dropped: w1 x1 <= 2^-22) replace six.  What fp16 lacks is range, and here the
// range is known before a single activation is read: the staged value is swish(gamma xhat + beta) with |xhat| <= sqrt(n - 1)
// for a group of n elements, so |value| <= sqrt(n - 1) max |gamma| + max |beta| =: A (hdiff_gn_act_scale evaluates it from the
// GroupNorm weights alone).  Activations are staged times 2^s with A 2^s < 2^15, weights are packed times 2^t with
// max |w| 2^t in [2^14, 2^15) (hdiff_pack_conv_weight_h2); the accumulator leaves through one multiplication by 2^-(s + t).
// Error of an operand: 2^-23 relative, or 2^-25 absolute in the scaled domain for values below 2^-3 there -- i.e. below
// 2^-18 of A resp. max |w|: fp32-class against the sum it enters (tests/test_gpu_ops.py holds the kernel to the bf16-triple
// kernel's gate: error against float64 within 1.5x of the fp32-MFMA kernel's).  The chain has 3 roundings per 16 input
// channels and tap instead of 6, and measures a SMALLER error than the bf16 triples (tools/h2_sim_conv.py).  A caller whose
// gn_scale / gn_shift are not GroupNorm statistics of x can break the bound: the fp16 conversion then yields inf and the
// output NaN -- loud, never silently wrong.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

using namespace hdiff;

namespace {

#ifndef CONVH2_ABL
#define CONVH2_ABL 0   // dev: timing ablations of the fp16-pair form (wrong results with any bit set; tools/README.md): 1 no staging of
#endif                 // the next chunk, 2 no weight loads in the loop, 4 no barrier, 8 no B operand reads in the loop, 16 no Swish
constexpr int THREADS = 256;
constexpr int PH = 10, PW = 34, PPIX = PH * PW;      // patch of an 8 x 32 tile
constexpr int NSLOT = (4 * PPIX + THREADS - 1) / THREADS;   // (channel quad, pixel) staging slots per thread: 6
constexpr int HALF_WORDS = PPIX * 4;                  // one half-plane: [pixel][4 words] = channel pairs 4h .. 4h+3 of every pixel
constexpr int PIECE_WORDS = 2 * HALF_WORDS;           // 32-bit words of one piece: [half h][pixel][4 channel pairs]
constexpr int PSTRIDE = PIECE_WORDS + 8;              // plane stride: each plane is followed by a dump area ...
constexpr int DUMP_WORD = PIECE_WORDS;                // ... where the unused staging slot of a thread stores (branch-free staging)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float swish_fast(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

__device__ __forceinline__ unsigned pack_hi16(float lo, float hi) {
  return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, hi), __builtin_bit_cast(unsigned, lo), 0x07060302u);
}
__device__ __forceinline__ float top16(float x) {
  return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & 0xffff0000u);
}
// (a, b) -> three packed bf16 pairs, a = a0 + a1 + a2 exactly (plain VALU only: see attention_x3.hip)
__device__ __forceinline__ void split3(float a, float b, unsigned& h0, unsigned& h1, unsigned& h2) {
  h0 = pack_hi16(a, b);
  const float ra = a - top16(a), rb = b - top16(b);
  h1 = pack_hi16(ra, rb);
  const float sa = ra - top16(ra), sb = rb - top16(rb);
  h2 = pack_hi16(sa, sb);
}

__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma_f16(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// (a, b) -> two packed fp16 pairs with a = h0.lo + h1.lo up to 2^-23 |a| (or 2^-25 absolute), b likewise in the high halves
// (attention_h2.hip).  `one` is 1.0f in a register the compiler cannot see through: fma(a, 1, -h) must stay an fma
// (v_fma_mixlo / mixhi_f16: the residual is exact in fp32 and rounded once).
__device__ __forceinline__ void split2(float a, float b, float one, unsigned& h0, unsigned& h1) {
  const f16x2 p = {(_Float16)a, (_Float16)b};                 // v_cvt_pk_f16_f32: round to nearest even
  unsigned u = __builtin_bit_cast(unsigned, p);
  asm("" : "+v"(u));
  const f16x2 q = __builtin_bit_cast(f16x2, u);
  const f16x2 r = {(_Float16)__builtin_fmaf(a, one, -(float)q[0]), (_Float16)__builtin_fmaf(b, one, -(float)q[1])};
  h0 = u;
  h1 = __builtin_bit_cast(unsigned, r);
}

// split-product terms kept (piece of W, piece of X): all i + j <= 2, small ones first
__device__ constexpr int TERM_W[6] = {2, 1, 0, 1, 0, 0};
__device__ constexpr int TERM_X[6] = {0, 1, 2, 0, 1, 0};

// NT: taps of the launch (9 = the 3x3 conv, 6 / 4 = transposed-conv phases with fewer taps); OUTMAP: the output pixel of
// (vy, vx) is (vy * out_sy + out_oy, vx * out_sx + out_ox) of an OH x OW plane (transposed-conv phases) instead of (vy, vx).
// PAIR: fp16 pairs and three products instead of bf16 triples and six (header).
template <int NT, bool OUTMAP, bool PAIR, int OCC = 1>
__global__ __launch_bounds__(THREADS, OCC) void conv3x3_x3_kernel(const ConvX3K p) {
  constexpr int NP = PAIR ? 2 : 3;         // pieces per operand
  __shared__ __attribute__((aligned(16))) unsigned sXbuf[2][NP * PSTRIDE];     // double-buffered: chunk c + 1 is staged beside chunk c's MFMAs
  extern __shared__ __attribute__((aligned(16))) float sG[];     // [2][Cin]: GroupNorm scale | shift of this sample

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int b = blockIdx.z;
  // Workgroup -> (pixel tile, channel block).  The channel blocks of ONE pixel tile read the same activation patch; in grid
  // order they are a whole plane of tiles apart and each fetched it from HBM again (measured on the fp32 kernel: input bytes x
  // Cout / 64).  Workgroups go to the 8 XCDs round robin by linear id, so the channel blocks of a tile are given linear ids
  // 8 apart: same XCD, consecutive in time -- the later ones find the patch in that XCD's L2.
  int tile_id = blockIdx.x, cob = blockIdx.y;
  if ((gridDim.x & 7u) == 0u && gridDim.y > 1u) {
    const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y, per = 8u * gridDim.y;
    const unsigned grp = lin / per, r = lin - grp * per;
    tile_id = (int)(grp * 8u + (r & 7u));
    cob = (int)(r >> 3);
  }
  const int co0 = cob * 64;
  const int tile_y = tile_id / p.tiles_x, tile_x = tile_id - tile_y * p.tiles_x;
  const int vy0 = tile_y * 8, vx0 = tile_x * 32;
  const bool has_gn = p.gn_scale != nullptr;
  const size_t HW = (size_t)p.H * p.W;
  const float xs = PAIR ? p.act_scale[0] : 1.0f;       // 2^s of the staged activations
  const float one = p.one;

  // LDS layout of a piece: two half-planes [h][pixel][4 words]; word w of half h holds the bf16 pieces of input channels
  // 8h + 2w, 8h + 2w + 1.  A lane's B operand (8 channels of one pixel) is ONE 16-byte read and consecutive lanes read
  // consecutive 16-byte blocks: conflict-free.  (Round 2's [pixel][8 words] put lanes 8 words apart: the operand reads were
  // 2-way and the 4-byte staging stores 8-way bank-conflicted -- SQ_LDS_BANK_CONFLICT was 60 % of the LDS-active cycles.)
  // staging slots: slot e = (channel quad jq = e / 340: channels 4 jq .. 4 jq + 3, patch pixel e % 340); 8-byte stores
  int s_goff[NSLOT];      // iy * W + ix of the pixel, or -1 (zero padding / unused slot)
  int s_lds[NSLOT];       // word offset inside a piece, or the dump word (unused slot)
  int s_quad[NSLOT];
#pragma unroll
  for (int i = 0; i < NSLOT; ++i) {
    const int e = tid + i * THREADS;
    const int jq = e / PPIX, pos = e - jq * PPIX;
    const int py = pos / PW, px = pos - py * PW;
    const int iy = vy0 - 1 + py, ix = vx0 - 1 + px;
    const bool used = jq < 4;
    s_quad[i] = used ? jq : 3;
    s_lds[i] = used ? (jq >> 1) * HALF_WORDS + pos * 4 + (jq & 1) * 2 : DUMP_WORD;
    s_goff[i] = (used && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) ? iy * p.W + ix : -1;
  }
  if (has_gn) {
    for (int i = tid; i < 2 * p.Cin; i += THREADS)
      sG[i] = (i < p.Cin) ? p.gn_scale[b * p.Cin + i] : p.gn_shift[b * p.Cin + (i - p.Cin)];
  }

  // Wave tile: 32 output channels x 4 pixel rows (wm = channel half, wn = row half of the 64 x (8 x 32) workgroup tile).
  // With all four waves on the same 64 channels (round 2: waves split the pixels only) every wave loaded the SAME weight
  // operands: 24 KB per tap per workgroup through the CU's 64 B/clk vector L1 -- 62 B/clk with two workgroups per CU, and
  // the PMC showed the waves parked on those loads 36 % of their time (MFMA busy 0.47).  The 2 x 2 split halves the
  // weight traffic and doubles the LDS operand reads, of which there is bandwidth to spare.
  const int wm = wave & 1, wn = wave >> 1;
  f32x16 acc[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;

  // XAHEAD (the fp16-pair form): the patch of chunk c + 2 is requested at the START of chunk c into a second register set and
  // moved over at its end -- requested at the end of chunk c (the bf16-triple form, whose taps take twice as long) the first
  // staging slot of chunk c + 1 consumed its load 400 cycles after it had left.
  constexpr bool XAHEAD = PAIR;
  f32x4 xv[NSLOT], xv2[XAHEAD ? NSLOT : 1];
  auto issue_loads_to = [&](f32x4 (&dst)[XAHEAD ? NSLOT : 1], int c0) {
    const float* xbase = (c0 < p.C0) ? p.x0 + ((size_t)b * p.C0 + c0) * HW : p.x1 + ((size_t)b * p.C1 + (c0 - p.C0)) * HW;
#pragma unroll
    for (int i = 0; i < (XAHEAD ? NSLOT : 0); ++i) {
      const bool ok = s_goff[i] >= 0;
      const float* src = xbase + (size_t)(4 * s_quad[i]) * HW + (ok ? s_goff[i] : 0);
      const float v0 = src[0], v1 = src[HW], v2 = src[2 * HW], v3 = src[3 * HW];
      dst[i] = f32x4{ok ? v0 : 0.f, ok ? v1 : 0.f, ok ? v2 : 0.f, ok ? v3 : 0.f};
    }
  };
  auto issue_loads = [&](int c0) {
    // a 16-channel chunk never straddles the concat seam (C0 % 16 == 0 is checked on the host)
    const float* xbase = (c0 < p.C0) ? p.x0 + ((size_t)b * p.C0 + c0) * HW : p.x1 + ((size_t)b * p.C1 + (c0 - p.C0)) * HW;
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) {
      const bool ok = s_goff[i] >= 0;
      const float* src = xbase + (size_t)(4 * s_quad[i]) * HW + (ok ? s_goff[i] : 0);
      const float v0 = src[0], v1 = src[HW], v2 = src[2 * HW], v3 = src[3 * HW];   // unconditional: the address is always valid
      xv[i] = f32x4{ok ? v0 : 0.f, ok ? v1 : 0.f, ok ? v2 : 0.f, ok ? v3 : 0.f};
    }
  };
  // Branch-free: a slot outside the image stages 0 by a select, a thread's unused last slot writes to a dump word behind the
  // piece planes -- with per-slot branches the GroupNorm table reads of a slot could not be issued before the previous slot
  // had finished, and every slot paid an LDS round trip of its own (11 per chunk and wave).
  auto stage_slot = [&](auto gn_tag, int i, int c0, unsigned* sX) {
    constexpr bool GN = decltype(gn_tag)::value;
    f32x4 v = xv[i];
    if (GN) {
      const int ci = c0 + 4 * s_quad[i];
      const f32x4 sc = *reinterpret_cast<const f32x4*>(&sG[ci]);
      const f32x4 sh = *reinterpret_cast<const f32x4*>(&sG[p.Cin + ci]);
      const bool inside = s_goff[i] >= 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float gk = (PAIR && (CONVH2_ABL & 16)) ? fmaf(v[k], sc[k], sh[k]) : swish_fast(fmaf(v[k], sc[k], sh[k]));
        v[k] = inside ? gk : 0.f;
      }
    }
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    if constexpr (PAIR) {
      unsigned a0, a1, c0w, c1w;
      split2(v[0] * xs, v[1] * xs, one, a0, a1);
      split2(v[2] * xs, v[3] * xs, one, c0w, c1w);
      if (HDIFF_MUTANT & 1) { a1 &= 0xffe0ffe0u; c1w &= 0xffe0ffe0u; }      // (mutation test: 2^-16 of every activation dropped)
      *reinterpret_cast<u32x2*>(&sX[s_lds[i]]) = u32x2{a0, c0w};
      *reinterpret_cast<u32x2*>(&sX[PSTRIDE + s_lds[i]]) = u32x2{a1, c1w};
    } else {
      unsigned a0, a1, a2, c0w, c1w, c2w;
      split3(v[0], v[1], a0, a1, a2);
      split3(v[2], v[3], c0w, c1w, c2w);
      *reinterpret_cast<u32x2*>(&sX[s_lds[i]]) = u32x2{a0, c0w};
      *reinterpret_cast<u32x2*>(&sX[PSTRIDE + s_lds[i]]) = u32x2{a1, c1w};
      *reinterpret_cast<u32x2*>(&sX[2 * PSTRIDE + s_lds[i]]) = u32x2{a2, c2w};
    }
  };
  auto store_staged = [&](auto gn_tag, int c0, unsigned* sX) {
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) stage_slot(gn_tag, i, c0, sX);
  };

  // operand addresses: B = 8 channels (h picks the half) of pixel (row, l31) of this wave's N tile; A = 8 input channels of
  // output channel co0 + mt*32 + l31
  int boff[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) boff[nt] = h * HALF_WORDS + ((wn * 4 + nt) * PW + l31) * 4;
  const unsigned* wlane = p.wp3 + ((size_t)(co0 + wm * 32 + l31) * 8 + h * 4);
  const size_t w_piece = (size_t)p.CoutPad * 8;            // words between pieces
  const size_t w_tap = NP * w_piece, w_chunk = NT * w_tap;

  auto load_w = [&](u32x4 (&wa)[NP], int chunk, int tap) {
    const unsigned* wb = wlane + (size_t)chunk * w_chunk + (size_t)tap * w_tap;
#pragma unroll
    for (int pc = 0; pc < NP; ++pc) wa[pc] = *reinterpret_cast<const u32x4*>(wb + pc * w_piece);
  };
  // One unit = (tap, pixel row nt): 6 MFMAs on acc[nt] (PAIR: 3).  The B operands of unit u + 1 are read from LDS at the start of unit
  // u and the weights of tap + 2 are requested at the start of tap (two taps = 1 500 MFMA cycles ahead: an L2 hit under load
  // takes about one tap), so that no unit starts by waiting for its own operands.
  auto load_x = [&](u32x4 (&xp)[NP], const unsigned* sX, int tap, int nt) {
    const int toff = p.tap_off[tap];                    // ((dy + 1) * PW + (dx + 1)) * 4, wave-uniform
#pragma unroll
    for (int pc = 0; pc < NP; ++pc) xp[pc] = *reinterpret_cast<const u32x4*>(&sX[pc * PSTRIDE + boff[nt] + toff]);
  };
  auto mma_unit = [&](const u32x4 (&wa)[NP], const u32x4 (&xp)[NP], int nt) {
    if constexpr (PAIR) {                  // small products first
      acc[nt] = mfma_f16(wa[1], xp[0], acc[nt]);
      acc[nt] = mfma_f16(wa[0], xp[1], acc[nt]);
      acc[nt] = mfma_f16(wa[0], xp[0], acc[nt]);
    } else {
#pragma unroll
      for (int t = 0; t < 6; ++t)
        if (!((HDIFF_MUTANT & 1) && TERM_W[t] == 0 && TERM_X[t] == 2)) acc[nt] = mfma_bf16(wa[TERM_W[t]], xp[TERM_X[t]], acc[nt]);
    }
  };

  // Chunk loop, ONE barrier per chunk: while the matrix core works through chunk c (LDS buffer c & 1), the vector pipe turns
  // the raw patch of chunk c + 1 (in registers since the previous chunk) into split pieces in the other buffer, one or two
  // staging slots behind every tap's MFMAs; the global loads of chunk c + 2 leave once those registers are free.
  // The weight operands form ONE stream over (chunk, tap), two taps ahead of their use, through a ring of three register sets
  // (WSTREAM: a tap count that is a multiple of three keeps the ring's phase from chunk to chunk; before, every chunk began by
  // requesting its first two taps and waiting for them -- with half the matrix time per tap the fp16-pair form spent 38 % of
  // its wave time waiting, SQ_WAIT_ANY).
  constexpr bool WSTREAM = (NT % 3 == 0);
  u32x4 w[3][NP];
  auto chunk = [&](auto gn_tag, auto more_tag, int c, int nchunks) {
    constexpr bool more = decltype(more_tag)::value;      // compile time: the staging must not sit in a block of its own
    const unsigned* sX = sXbuf[c & 1];
    unsigned* sNext = sXbuf[(c + 1) & 1];
    u32x4 xp[2][NP];
    if (XAHEAD && c + 2 < nchunks) issue_loads_to(xv2, (c + 2) * 16);
    if (!WSTREAM || c == 0) {
      load_w(w[0], c, 0);
      load_w(w[1], c, 1);
    }
    load_x(xp[0], sX, 0, 0);
#pragma unroll
    for (int tap = 0; tap < NT; ++tap) {
      if (PAIR && (CONVH2_ABL & 2)) {
      } else if (tap + 2 < NT) load_w(w[(tap + 2) % 3], c, tap + 2);
      else if (WSTREAM && more) load_w(w[(tap + 2) % 3], c + 1, tap + 2 - NT);      // the next chunk's first two taps
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int u = tap * 4 + nt;
        if (u + 1 < 4 * NT && !(PAIR && (CONVH2_ABL & 8))) load_x(xp[(u + 1) & 1], sX, (u + 1) / 4, (u + 1) % 4);
        mma_unit(w[tap % 3], xp[(PAIR && (CONVH2_ABL & 8)) ? 0 : (u & 1)], nt);
      }
      if (more && tap < NSLOT && !(PAIR && (CONVH2_ABL & 1))) stage_slot(gn_tag, tap, (c + 1) * 16, sNext);
    }
    if (more) {
#pragma unroll
      for (int i = NT; i < NSLOT; ++i) stage_slot(gn_tag, i, (c + 1) * 16, sNext);     // fewer taps than staging slots
    }
    if constexpr (XAHEAD) {
      if (c + 2 < nchunks) {
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) xv[i] = xv2[i];
      }
    } else {
      if (c + 2 < nchunks) issue_loads((c + 2) * 16);
    }
    if (!(PAIR && (CONVH2_ABL & 4))) __syncthreads();
  };

  const int nchunks = p.Cin / 16;
  issue_loads(0);
  __syncthreads();     // sG visible
  if (has_gn) store_staged(std::true_type{}, 0, sXbuf[0]);
  else store_staged(std::false_type{}, 0, sXbuf[0]);
  if (nchunks > 1) issue_loads(16);
  __syncthreads();
  if (has_gn) {
    for (int c = 0; c + 1 < nchunks; ++c) chunk(std::true_type{}, std::true_type{}, c, nchunks);
    chunk(std::true_type{}, std::false_type{}, nchunks - 1, nchunks);
  } else {
    for (int c = 0; c + 1 < nchunks; ++c) chunk(std::false_type{}, std::true_type{}, c, nchunks);
    chunk(std::false_type{}, std::false_type{}, nchunks - 1, nchunks);
  }

  // ---- epilogue: + bias + per-sample channel vector + residual, NCHW store (32 consecutive pixels per register row).
  // A full tile (workgroup-uniform test: everything but edge tiles and a channel tail) takes the branch-free form: the 16
  // bias / vector values of the lane's channels are fetched together, then per pixel row all 16 residual loads are in
  // flight before the first add -- with per-element tests every output waited for its own three loads in turn (64 dependent
  // round trips per lane: a quarter of the kernel's time at 128 channels).
  if constexpr (PAIR) {                    // out of the scaled domain: 2^-(s + t), exact
    const float os = p.act_scale[1] * p.w_scale[1];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][r] *= os;
  }
  const size_t OHW = OUTMAP ? (size_t)p.OH * p.OW : HW;
  const int osy = OUTMAP ? p.out_sy : 1, ooy = OUTMAP ? p.out_oy : 0, osx = OUTMAP ? p.out_sx : 1, oox = OUTMAP ? p.out_ox : 0;
  const int OW = OUTMAP ? p.OW : p.W;
  if (co0 + 64 <= p.Cout && vy0 + 8 <= p.H && vx0 + 32 <= p.W) {
    float add[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      float a = 0.f;
      if (p.bias) a += p.bias[co];
      if (p.addvec) a += p.addvec[b * p.Cout + co];
      add[r] = a;
    }
    const size_t lane_base = ((size_t)b * p.Cout + co0 + wm * 32 + 4 * h) * OHW +
                             (size_t)((vy0 + wn * 4) * osy + ooy) * OW + (size_t)(vx0 + l31) * osx + oox;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const size_t row = lane_base + (size_t)(nt * osy) * OW;
      float res[16];
      if (p.residual) {
#pragma unroll
        for (int r = 0; r < 16; ++r) res[r] = p.residual[row + (size_t)((r & 3) + 8 * (r >> 2)) * OHW];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[nt][r] + add[r];
        if (p.residual) v += res[r];
        p.out[row + (size_t)((r & 3) + 8 * (r >> 2)) * OHW] = v;
      }
    }
    return;
  }
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const int vy = vy0 + wn * 4 + nt, vx = vx0 + l31;
    if (vy >= p.H || vx >= p.W) continue;
    const size_t pix = (size_t)(vy * osy + ooy) * OW + (size_t)vx * osx + oox;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (co < p.Cout) {
        float a = 0.f;                               // the same order of additions as the full-tile form
        if (p.bias) a += p.bias[co];
        if (p.addvec) a += p.addvec[b * p.Cout + co];
        float v = acc[nt][r] + a;
        const size_t o = ((size_t)b * p.Cout + co) * OHW + pix;
        if (p.residual) v += p.residual[o];
        p.out[o] = v;
      }
    }
  }
}

// fp32 weights -> [Cin/16][tap][piece][CoutPad][8 words]: word j of a row = bf16 pieces of input channels 2j, 2j+1.  Tap t
// reads kernel element (ky[t], kx[t]) of a KH x KW kernel stored [Cout][Cin][KH][KW] (mode 0) or [Cin][Cout][KH][KW] (mode 1:
// nn.ConvTranspose2d's layout, and the layout of a forward weight seen from its input-gradient convolution).
struct PackX3K {
  int mode, Cout, Cin, KH, KW, ntaps, CoutPad;
  int ky[9], kx[9];
};
__global__ void pack_conv_weight_x3_kernel(const float* __restrict__ w, unsigned* __restrict__ wp3, const PackX3K q) {
  const size_t n = (size_t)(q.Cin / 16) * q.ntaps * q.CoutPad * 8;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int j = (int)(i % 8);
    size_t r = i / 8;
    const int co = (int)(r % q.CoutPad);
    r /= q.CoutPad;
    const int tap = (int)(r % q.ntaps);
    const int chunk = (int)(r / q.ntaps);
    float a = 0.f, c = 0.f;
    if (co < q.Cout) {
      const int ci = chunk * 16 + 2 * j;
      const size_t k = (size_t)q.ky[tap] * q.KW + q.kx[tap], kk = (size_t)q.KH * q.KW;
      if (q.mode == 1) {
        a = w[((size_t)ci * q.Cout + co) * kk + k];
        c = w[((size_t)(ci + 1) * q.Cout + co) * kk + k];
      } else {
        a = w[((size_t)co * q.Cin + ci) * kk + k];
        c = w[((size_t)co * q.Cin + ci + 1) * kk + k];
      }
    }
    unsigned h0, h1, h2;
    split3(a, c, h0, h1, h2);
    const size_t base = ((size_t)(chunk * q.ntaps + tap) * 3) * q.CoutPad * 8 + (size_t)co * 8 + j;
    wp3[base] = h0;
    wp3[base + (size_t)q.CoutPad * 8] = h1;
    wp3[base + 2 * (size_t)q.CoutPad * 8] = h2;
  }
}

// ---- fp16-pair weights: [Cin/16][tap][2 pieces][CoutPad][8 words] of w 2^t, then a tail of 4 words:
//      {bits of max |w| (scratch of the pack), 2^-t, 2^t, 0}.  Two launches: the maximum, then the split.
__global__ void conv_weight_absmax_kernel(const float* __restrict__ w, size_t n, unsigned* __restrict__ tail) {
  float m = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(w[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(tail, __builtin_bit_cast(unsigned, m));      // non-negative floats order like unsigned integers
}
__global__ void pack_conv_weight_h2_kernel(const float* __restrict__ w, unsigned* __restrict__ wp2, const PackX3K q, float one) {
  const size_t n = (size_t)(q.Cin / 16) * q.ntaps * q.CoutPad * 8;
  unsigned* tail = wp2 + 2 * n;
  // 2^t with max |w| 2^t in [2^14, 2^15); an all-zero / denormal / non-finite tensor gets a fixed scale (inf and NaN stay what they are)
  int e = (int)((tail[0] >> 23) & 0xffu) - 127;
  e = e < -60 ? -60 : (e > 60 ? 60 : e);
  const float sc = __builtin_bit_cast(float, (unsigned)(14 - e + 127) << 23);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int j = (int)(i % 8);
    size_t r = i / 8;
    const int co = (int)(r % q.CoutPad);
    r /= q.CoutPad;
    const int tap = (int)(r % q.ntaps);
    const int chunk = (int)(r / q.ntaps);
    float a = 0.f, c = 0.f;
    if (co < q.Cout) {
      const int ci = chunk * 16 + 2 * j;
      const size_t k = (size_t)q.ky[tap] * q.KW + q.kx[tap], kk = (size_t)q.KH * q.KW;
      a = w[((size_t)co * q.Cin + ci) * kk + k];
      c = w[((size_t)co * q.Cin + ci + 1) * kk + k];
    }
    unsigned h0, h1;
    split2(a * sc, c * sc, one, h0, h1);
    const size_t base = ((size_t)(chunk * q.ntaps + tap) * 2) * q.CoutPad * 8 + (size_t)co * 8 + j;
    wp2[base] = h0;
    wp2[base + (size_t)q.CoutPad * 8] = h1;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    reinterpret_cast<float*>(tail)[1] = __builtin_bit_cast(float, (unsigned)(e - 14 + 127) << 23);
    reinterpret_cast<float*>(tail)[2] = sc;
    tail[3] = 0u;
  }
}

// out[0] = 2^s, out[1] = 2^-s with (sqrt(n - 1) max |gamma| + max |beta|) 2^s in [2^13, 2^14): the bound of a GroupNorm + Swish
// output (header) with a factor of two to spare.  One workgroup.
__global__ void gn_act_scale_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, int C, float sqrt_n1,
                                    float gain, float* __restrict__ out) {
  __shared__ float red[4];
  float m = 0.f;
  for (int i = threadIdx.x; i < C; i += blockDim.x) m = fmaxf(m, fmaf(sqrt_n1, fabsf(gamma[i]), fabsf(beta[i])));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) * gain;
    int e = (int)((__builtin_bit_cast(unsigned, m) >> 23) & 0xffu) - 127;
    e = e < -60 ? -60 : (e > 60 ? 60 : e);           // NaN / inf weights: a fixed scale, the activations are NaN anyway
    out[0] = __builtin_bit_cast(float, (unsigned)(13 - e + 127) << 23);
    out[1] = __builtin_bit_cast(float, (unsigned)(e - 13 + 127) << 23);
  }
}

}  // namespace

namespace hdiff {

void launch_conv3x3_x3(const ConvX3K& k, int B, hipStream_t stream) {
  const int tiles_y = cdiv(k.H, 8);
  dim3 grid(k.tiles_x * tiles_y, cdiv(k.Cout, 64), B);
  const size_t dyn = (size_t)2 * k.Cin * sizeof(float);
  const bool outmap = !(k.out_sy == 1 && k.out_oy == 0 && k.out_sx == 1 && k.out_ox == 0 && k.OH == k.H && k.OW == k.W);
  if (k.act_scale != nullptr) {            // fp16 pairs: the plain 3x3 conv behind GroupNorm + Swish (the dispatcher checked the shape)
    hipLaunchKernelGGL((conv3x3_x3_kernel<9, false, true>), grid, dim3(THREADS), dyn, stream, k);
    return;
  }
  if (k.ntaps == 9 && !outmap) hipLaunchKernelGGL((conv3x3_x3_kernel<9, false, false>), grid, dim3(THREADS), dyn, stream, k);
  else if (k.ntaps == 9) hipLaunchKernelGGL((conv3x3_x3_kernel<9, true, false>), grid, dim3(THREADS), dyn, stream, k);
  else if (k.ntaps == 6) hipLaunchKernelGGL((conv3x3_x3_kernel<6, true, false>), grid, dim3(THREADS), dyn, stream, k);
  else hipLaunchKernelGGL((conv3x3_x3_kernel<4, true, false>), grid, dim3(THREADS), dyn, stream, k);
}

}  // namespace hdiff

static int pack_x3(const float* w, void* wp3, int mode, int Cout, int Cin, int KH, int KW, int ntaps, const int* ky, const int* kx,
                   int CoutPad, hipStream_t stream) {
  PackX3K q{};
  q.mode = mode; q.Cout = Cout; q.Cin = Cin; q.KH = KH; q.KW = KW; q.ntaps = ntaps; q.CoutPad = CoutPad;
  for (int t = 0; t < ntaps; ++t) { q.ky[t] = ky[t]; q.kx[t] = kx[t]; }
  const size_t n = (size_t)(Cin / 16) * ntaps * CoutPad * 8;
  const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  (void)hipGetLastError();  // drop any stale error left by another HIP user in this thread
  hipLaunchKernelGGL(pack_conv_weight_x3_kernel, dim3(blocks), dim3(256), 0, stream, w, (unsigned*)wp3, q);
  HDIFF_CHECK_LAUNCH("pack_conv_weight_x3_kernel");
  return HDIFF_OK;
}

extern "C" int hdiff_pack_conv_weight_x3(const float* w, void* wp3, int Cout, int Cin, int CoutPad, int transposed,
                                         hdiff_stream_t stream) {
  HDIFF_CHECK_ARG(w && wp3, "pack_conv_weight_x3: null pointer");
  HDIFF_CHECK_ARG(Cout > 0 && Cin > 0 && Cin % 16 == 0 && CoutPad >= Cout && CoutPad % 64 == 0,
                  "pack_conv_weight_x3: needs Cin %% 16 == 0 and CoutPad %% 64 == 0 (Cin %d, Cout %d, CoutPad %d)", Cin, Cout, CoutPad);
  int ky[9], kx[9];
  for (int t = 0; t < 9; ++t) {                 // the input-gradient conv reads the forward weight transposed and tap-mirrored
    ky[t] = transposed ? 2 - t / 3 : t / 3;
    kx[t] = transposed ? 2 - t % 3 : t % 3;
  }
  return pack_x3(w, wp3, transposed ? 1 : 0, Cout, Cin, 3, 3, 9, ky, kx, CoutPad, (hipStream_t)stream);
}

extern "C" int hdiff_pack_conv_weight_h2_words(int Cout, int Cin, int CoutPad, int64_t* words_out) {
  HDIFF_CHECK_ARG(words_out, "pack_conv_weight_h2_words: null pointer");
  HDIFF_CHECK_ARG(Cout > 0 && Cin > 0 && Cin % 16 == 0 && CoutPad >= Cout && CoutPad % 64 == 0,
                  "pack_conv_weight_h2_words: needs Cin %% 16 == 0 and CoutPad %% 64 == 0 (Cin %d, Cout %d, CoutPad %d)", Cin, Cout, CoutPad);
  *words_out = (int64_t)(Cin / 16) * 9 * 2 * CoutPad * 8 + 4;
  return HDIFF_OK;
}

extern "C" int hdiff_pack_conv_weight_h2(const float* w, void* wp2, int Cout, int Cin, int CoutPad, hdiff_stream_t stream) {
  HDIFF_CHECK_ARG(w && wp2, "pack_conv_weight_h2: null pointer");
  HDIFF_CHECK_ARG(Cout > 0 && Cin > 0 && Cin % 16 == 0 && CoutPad >= Cout && CoutPad % 64 == 0,
                  "pack_conv_weight_h2: needs Cin %% 16 == 0 and CoutPad %% 64 == 0 (Cin %d, Cout %d, CoutPad %d)", Cin, Cout, CoutPad);
  PackX3K q{};
  q.mode = 0; q.Cout = Cout; q.Cin = Cin; q.KH = 3; q.KW = 3; q.ntaps = 9; q.CoutPad = CoutPad;
  for (int t = 0; t < 9; ++t) { q.ky[t] = t / 3; q.kx[t] = t % 3; }
  const size_t n = (size_t)(Cin / 16) * 9 * CoutPad * 8, nw = (size_t)Cout * Cin * 9;
  unsigned* tail = (unsigned*)wp2 + 2 * n;
  const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  const int mblocks = (int)((nw + 255) / 256 < 1024 ? (nw + 255) / 256 : 1024);
  (void)hipGetLastError();  // drop any stale error left by another HIP user in this thread
  if (hipMemsetAsync(tail, 0, 16, (hipStream_t)stream) != hipSuccess) {
    hdiff::set_error("pack_conv_weight_h2: hipMemsetAsync failed");
    return HDIFF_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(conv_weight_absmax_kernel, dim3(mblocks), dim3(256), 0, (hipStream_t)stream, w, nw, tail);
  HDIFF_CHECK_LAUNCH("conv_weight_absmax_kernel");
  hipLaunchKernelGGL(pack_conv_weight_h2_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, (unsigned*)wp2, q, 1.0f);
  HDIFF_CHECK_LAUNCH("pack_conv_weight_h2_kernel");
  return HDIFF_OK;
}

extern "C" int hdiff_gn_act_scale(const float* gamma, const float* beta, int C, int64_t group_elems, float gain, float* out2,
                                  hdiff_stream_t stream) {
  HDIFF_CHECK_ARG(gamma && beta && out2, "gn_act_scale: null pointer");
  HDIFF_CHECK_ARG(C > 0 && group_elems > 0 && gain >= 1.0f && gain <= 1024.0f, "gn_act_scale: bad sizes (C %d, gain %g)", C, (double)gain);
  (void)hipGetLastError();  // drop any stale error left by another HIP user in this thread
  hipLaunchKernelGGL(gn_act_scale_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, gamma, beta, C,
                     sqrtf((float)(group_elems > 1 ? group_elems - 1 : 1)), gain, out2);
  HDIFF_CHECK_LAUNCH("gn_act_scale_kernel");
  return HDIFF_OK;
}

extern "C" int hdiff_pack_conv_weight_x3_taps(const float* w, void* wp3, int mode, int Cout, int Cin, int KH, int KW, int ntaps,
                                              const int* tap_ky, const int* tap_kx, int CoutPad, hdiff_stream_t stream) {
  HDIFF_CHECK_ARG(w && wp3 && tap_ky && tap_kx, "pack_conv_weight_x3_taps: null pointer");
  HDIFF_CHECK_ARG(Cout > 0 && Cin > 0 && Cin % 16 == 0 && CoutPad >= Cout && CoutPad % 64 == 0 && (mode == 0 || mode == 1),
                  "pack_conv_weight_x3_taps: needs Cin %% 16 == 0, CoutPad %% 64 == 0, mode 0 / 1 (Cin %d, Cout %d, CoutPad %d, mode %d)",
                  Cin, Cout, CoutPad, mode);
  HDIFF_CHECK_ARG(ntaps >= 1 && ntaps <= 9 && KH > 0 && KW > 0, "pack_conv_weight_x3_taps: ntaps %d out of range", ntaps);
  for (int t = 0; t < ntaps; ++t)
    HDIFF_CHECK_ARG(tap_ky[t] >= 0 && tap_ky[t] < KH && tap_kx[t] >= 0 && tap_kx[t] < KW,
                    "pack_conv_weight_x3_taps: tap %d reads kernel element (%d, %d) of a %d x %d kernel", t, tap_ky[t], tap_kx[t], KH, KW);
  return pack_x3(w, wp3, mode, Cout, Cin, KH, KW, ntaps, tap_ky, tap_kx, CoutPad, (hipStream_t)stream);
}
