// 1x1 / stride-1 convolution in the split-bf16 contraction mode: the GEMM of conv1x1_direct.hip with every fp32 operand carried
// as three bf16 pieces on v_mfma_f32_32x32x16_bf16 (x = x0 + x1 + x2 exactly, six piece products with i + j <= 2, fp32
// accumulation: the error class of an fp32 fma chain, conv3x3_x3.hip / attention_x3.hip).  Call sites in the reference: the packed
// in- / out-projections of nn.MultiheadAttention viewed as 1x1 convs (ModelCondition.py:189, diffusion/Model.py:291) and the
// ResBlock shortcut (ModelCondition.py:192, Model.py:294); in training also their input-gradient convs.
//
// Why triples and not the fp16 pairs of the 3x3 kernel: these inputs (a ResBlock's output, the attention output, a block's
// input) have no range that is known before they are read, and a maximum pass per launch costs a fifth of what the pairs would
// save (profiles/r04_rejected_candidates.txt).  bf16 has fp32's exponent range: no scale, no pass.
//
// Why its own kernel (round 3 routed the 1x1 convs through the 3x3 kernel's centre tap and measured them SLOWER than the fp32
// kernel, 61-84 against 76-86 TFLOP/s): there one tap gives the staging split nothing to hide behind -- a workgroup split a
// 10 x 34 patch through LDS for 64 channels and one tap.  Here, as in conv1x1_direct.hip, the activations do not go through LDS:
// a lane loads the 8 input channels of its two pixels straight from the NCHW rows (8-byte loads, coalesced over the lanes),
// splits them in registers (5.5 vector instructions per value) into the B operands of its two pixel tiles, and each split value
// feeds 64 output channels x 6 products; the weights arrive pre-split from hdiff_pack_conv_weight_x3_taps (one tap) and are
// staged once per workgroup and chunk in LDS (WLDS below), from where every wave reads its 16-byte A operands.  Per 16-channel step and wave: 24 MFMAs (768 matrix cycles) beside ~120 vector
// instructions; the fp32 kernel spends 2048 matrix cycles on the same step.  Measured: 84-86 TFLOP/s-eq for 128 -> 384 at 256^2,
// batch 16 (fp32 kernel 79-81), 108-112 for 256 -> 768 at 128^2 and 384 -> 128 at 256^2 (84-90), error against float64 0.88x the
// fp32 kernel's -- a modest step, not the 150 the matrix time alone would allow: see the ablation note in the loop.
#include <stdlib.h>

#include "common.h"

using namespace hdiff;

namespace {

#ifndef C1X3_ABL
#define C1X3_ABL 0   // dev: timing ablations (wrong results with any bit set): 1 no X loads in the loop, 2 no weight loads in the loop,
#endif               // 4 no split (raw bits as pieces), 8 no MFMAs, 16 no epilogue loads / stores (one store per lane)
constexpr int THREADS = 256;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
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
__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// split-product terms kept (piece of W, piece of X): all i + j <= 2, small ones first
__device__ constexpr int TERM_W[6] = {2, 1, 0, 1, 0, 0};
__device__ constexpr int TERM_X[6] = {0, 1, 2, 0, 1, 0};

// A wave owns 64 output channels x 64 pixels: 2 x 2 accumulators of 32x32.  N tile nt = the pixel set {2 l + nt} (one 8-byte
// load of X per channel serves both tiles), M tile mt = channels co0 + 32 mt + l; a workgroup = 4 waves = 256 consecutive pixels.
// Preconditions (checked by the dispatcher in conv_igemm.hip): HW % 256 == 0, Cin % 16 == 0, C0 % 16 == 0 (a 16-channel step
// never straddles the concat seam: the activation base pointer is wave-uniform), CoutPad % 64 == 0.
// WLDS: the chunk's weight pieces (6 KB per workgroup) are staged ONCE per workgroup in LDS and every wave reads its A operands
// from there, instead of every wave fetching its own 6 KB from L2 -- the 48 KB of a channel block's weights do not fit the 32 KB
// vector L1, so without this the four waves of a workgroup quadruple the L2 -> L1 traffic of the weights (one LDS-only barrier per
// chunk).  Measured +4 ... +8 % (1.237 -> 1.186 ms for 128 -> 384 at 256^2, batch 16; 0.994 -> 0.915 for 384 -> 128): the default.
template <bool WLDS>
__global__ __launch_bounds__(THREADS, 3) void conv1x1_x3_kernel(const Conv1x1X3K p) {
  constexpr int MT = 2, WN = 2;
  __shared__ __attribute__((aligned(16))) u32x4 sW[WLDS ? 2 : 1][WLDS ? 3 * 64 * 2 : 1];      // [buffer][piece][channel][half]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  // Workgroup -> (pixel tile, channel block): the channel blocks of one pixel tile read the same X rows.  Workgroups go to the 8
  // XCDs round robin by linear id, so those blocks get linear ids 8 apart -- same XCD, consecutive in time: the later ones find
  // the rows in that XCD's L2 (conv3x3_x3.hip has the same map).
  int cob = blockIdx.x, tile_id = blockIdx.y;
  if ((gridDim.y & 7u) == 0u && gridDim.x > 1u) {
    const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y, per = 8u * gridDim.x;
    const unsigned grp = lin / per, r = lin - grp * per;
    tile_id = (int)(grp * 8u + (r & 7u));
    cob = (int)(r >> 3);
  }
  const int co0 = cob * 64;
  const long px0 = ((long)tile_id * 4 + __builtin_amdgcn_readfirstlane(wave)) * 64;      // wave-uniform
  const int b = blockIdx.z;
  const int C1 = p.Cin - p.C0;
  const unsigned hw = (unsigned)p.HW;
  const float* x0b = p.x0 + (size_t)b * p.C0 * p.HW + px0;
  const float* x1b = p.x1 ? p.x1 + (size_t)b * C1 * p.HW + px0 - (size_t)p.C0 * p.HW : x0b;
  // weights: [Cin/16][1 tap][3 pieces][CoutPad][8 words]; a lane's A operand = words 4h .. 4h+3 of channel row co0 + 32 mt + l31
  const unsigned* wlane = p.wp3 + ((size_t)(co0 + l31) * 8 + h * 4);
  const size_t w_piece = (size_t)p.CoutPad * 8, w_chunk = 3 * w_piece;

  f32x16 acc[MT][WN];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < WN; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  // X and the weights are requested one chunk ahead of their use.  (Measured, profiles/r04_conv1x1_x3.txt: three chunks of
  // lookahead for X, or three waves per SIMD instead of two, change nothing -- the kernel is not latency-bound; its timing
  // ablations are additive: stores 34 %, MFMAs 25 %, weight loads 16 %, X loads 15 %, the split 5 %.)
  constexpr int XD = 1;
  const int nchunks = p.Cin / 16;
  f32x2 xr[XD + 1][8];
  u32x4 wr[2][MT][3];
  auto load_x = [&](f32x2 (&x)[8], int c) {
    c = min(c, nchunks - 1);                                      // uniform; the pipeline's overshoot re-reads the last chunk
    const float* xb = (16 * c < p.C0) ? x0b : x1b;                 // uniform select
    const unsigned k0 = (unsigned)(16 * c + 8 * h);
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = *reinterpret_cast<const f32x2*>(xb + (size_t)((k0 + j) * hw + 2u * l31));
  };
  auto load_w = [&](u32x4 (&w)[MT][3], int c) {
    const unsigned* wb = wlane + (size_t)min(c, nchunks - 1) * w_chunk;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) w[mt][pc] = *reinterpret_cast<const u32x4*>(wb + pc * w_piece + (size_t)mt * 32 * 8);
  };
  auto mma = [&](const f32x2 (&x)[8], const u32x4 (&w)[MT][3]) {
    u32x4 xp[WN][3];
#pragma unroll
    for (int nt = 0; nt < WN; ++nt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        unsigned h0, h1, h2;
        if (C1X3_ABL & 4) { h0 = __builtin_bit_cast(unsigned, x[2 * j][nt]); h1 = __builtin_bit_cast(unsigned, x[2 * j + 1][nt]); h2 = h0 ^ h1; }
        else split3(x[2 * j][nt], x[2 * j + 1][nt], h0, h1, h2);
        xp[nt][0][j] = h0; xp[nt][1][j] = h1; xp[nt][2][j] = h2;
      }
#pragma unroll
    for (int t = 0; t < 6; ++t) {
      if ((HDIFF_MUTANT & 1) && TERM_W[t] == 0 && TERM_X[t] == 2) continue;      // (mutation test: the 2^-16 term w0 x2 dropped)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < WN; ++nt) {
          if (C1X3_ABL & 8) { acc[mt][nt][t] += __builtin_bit_cast(float, w[mt][TERM_W[t]][nt] ^ xp[nt][TERM_X[t]][mt]); }
          else acc[mt][nt] = mfma_bf16(w[mt][TERM_W[t]], xp[nt][TERM_X[t]], acc[mt][nt]);
        }
    }
  };

  if constexpr (WLDS) {
    // staging: 384 units of 16 bytes per chunk; unit u = (piece, channel, half); thread tid takes u = tid and u = 256 + tid (< 384)
    const unsigned* wsrc[2];
    int wdst[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int u = i * THREADS + tid;
      if (u >= 384) u -= 128;                       // the spare half round repeats units 256 .. 383 (same bytes to the same place)
      const int pc = u >> 7, rem = u & 127;
      wsrc[i] = p.wp3 + (size_t)pc * w_piece + (size_t)(co0 + (rem >> 1)) * 8 + (rem & 1) * 4;
      wdst[i] = u;
    }
    u32x4 wst[2];
    auto wload = [&](int c) {
      const size_t off = (size_t)min(c, nchunks - 1) * w_chunk;
#pragma unroll
      for (int i = 0; i < 2; ++i) wst[i] = *reinterpret_cast<const u32x4*>(wsrc[i] + off);
    };
    auto wstore = [&](int buf) {
#pragma unroll
      for (int i = 0; i < 2; ++i) sW[buf][wdst[i]] = wst[i];
    };
    auto wread = [&](u32x4 (&w)[MT][3], int buf) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) w[mt][pc] = sW[buf][(pc * 64 + 32 * mt + l31) * 2 + h];
    };
    wload(0);
    load_x(xr[0], 0);
    wstore(0);
    __syncthreads();
    for (int c = 0; c < nchunks; c += 2) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (c + i < nchunks) {                       // uniform
          wload(c + i + 1);
          load_x(xr[(i + 1) & 1], c + i + 1);
          wread(wr[0], i);
          mma(xr[i], wr[0]);
          wstore((i + 1) & 1);                       // buffer (i + 1) & 1 was last read one chunk ago, before the previous barrier
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // orders LDS only: the X loads of the next chunk stay in flight
        }
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < XD; ++i) load_x(xr[i], i);
    load_w(wr[0], 0);
    // the ring indices must be compile-time constants (register arrays): XD + 1 chunks per trip, weights alternate
    for (int c = 0; c < nchunks; c += XD + 1) {
#pragma unroll
      for (int i = 0; i < XD + 1; ++i) {
        if (c + i < nchunks) {                         // uniform
          if (!(C1X3_ABL & 1)) load_x(xr[(i + XD) % (XD + 1)], c + i + XD);
          if (!(C1X3_ABL & 2)) load_w(wr[(i + 1) & 1], c + i + 1);
          mma(xr[i], wr[i & 1]);
        }
      }
    }
  }

  // ---- epilogue: + bias + per-sample channel vector + residual, 8-byte stores (pixels 2 l31, 2 l31 + 1 of a channel row).
  // accumulator register r of tile mt holds channel co0 + 32 mt + (r & 3) + 8 (r >> 2) + 4 h
  const size_t blk = ((size_t)b * p.Cout) * p.HW + px0;
  const bool full = co0 + 64 <= p.Cout;
  if (C1X3_ABL & 16) {
    float t = 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < WN; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) t += acc[mt][nt][r];
    p.out[blk + (size_t)(co0 + 4 * h) * p.HW + 2u * l31] = t;
    return;
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
    for (int r0 = 0; r0 < 16; r0 += 8) {
      f32x2 res[8];
      float add[8];
      bool ok[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int r = r0 + j;
        const int co = co0 + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * h;
        ok[j] = full || co < p.Cout;
        const int cc = ok[j] ? co : 0;
        float a = 0.f;
        if (p.bias) a += p.bias[cc];
        if (p.addvec) a += p.addvec[(size_t)b * p.Cout + cc];
        add[j] = a;
        if (p.residual) res[j] = *reinterpret_cast<const f32x2*>(p.residual + blk + (size_t)cc * p.HW + 2u * l31);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int r = r0 + j;
        const int co = co0 + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * h;
        f32x2 v = {acc[mt][0][r], acc[mt][1][r]};
        v += add[j];
        if (p.residual) v += res[j];
        if (ok[j]) *reinterpret_cast<f32x2*>(p.out + blk + (size_t)co * p.HW + 2u * l31) = v;
      }
    }
  }
}

}  // namespace

namespace hdiff {

void launch_conv1x1_x3(const Conv1x1X3K& k, int B, hipStream_t stream) {
  dim3 grid(cdiv(k.Cout, 64), (unsigned)(k.HW / 256), B);
  hipLaunchKernelGGL(conv1x1_x3_kernel<true>, grid, dim3(THREADS), 0, stream, k);      // <false> (every wave fetches its own weight operands from L2): tools builds
}

}  // namespace hdiff
