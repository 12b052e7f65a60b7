// Implicit-GEMM convolution on the fp32-input MFMA (v_mfma_f32_32x32x2_f32), gfx950.
//
// One kernel serves every convolution of the hot path (reference call sites: ModelCondition.py:71-75, 82-88, 172, 186,
// 192, 219, 251 and the 1x1-conv view of nn.MultiheadAttention's projections, :189):
//     D[co][pixel] = sum_{tap, ci} Wp[tap][ci][co] * act(X[ci][pixel*stride + tap])
// GEMM roles: A = weights (M = 64 output channels per block), B = activations (N = 128 or 256 output pixels per block),
// K = taps x input channels, walked in chunks of CK channels.  Per chunk the block stages
//   * the NCHW input patch (tile + halo) of CK channels into LDS once -- every tap reads it at a shifted address, and
//     GroupNorm-affine + Swish (the reference's block1/block2 prologue) is applied while staging,
//   * the packed weight slab [ntaps][CK][64] with 16-byte loads.
// Global loads of chunk c+1 are issued before the MFMAs of chunk c and stay in flight under them; LDS operands are
// fetched one (tap, k-pair) step ahead of the MFMAs that consume them.
// Each wave owns 64 channels x (32*WN) pixels: 2 x WN accumulators of 32x32 (16 VGPRs each).
// fp32 MFMA is exact fp32 (a k-ordered fma chain), so results differ from the reference only by summation order.
#include <stdlib.h>

#include "common.h"

using namespace hdiff;

namespace {

constexpr int BM = 64;
constexpr int NTHREADS = 256;

struct ConvK {
  const float* x0;
  const float* x1;
  int C0, C1, Cin, H, W;
  const float* wp;
  int CinPad, CoutPad, Cout;
  const float* bias;
  const float* gn_scale;
  const float* gn_shift;
  const float* addvec;
  const float* residual;
  float* out;
  int OH, OW, VH, VW, in_stride, out_sy, out_oy, out_sx, out_ox;
  int ntaps, dy_min, dx_min, PH, PW, PWp, PLANE, XFLOATS, WFLOATS;
  int tw_log2, TH, tiles_x;
  int rows;                        // patch rows actually staged: min(TH, VH) (tiny images: the tile is taller than the image)
  int nx, nw;                      // staging slots in use per thread (activation floats, weight float4s)
  int B, ksplit, chunks_per_split; // split-K over input-channel chunks for small grids (partials -> conv_splitk_reduce)
  float* partial;                  // [ksplit][B][Cout][VH*VW]
  int tap_off[HDIFF_MAX_TAPS];
};

// Swish with the hardware reciprocal (1 ulp) instead of an IEEE division: the prologue runs once per staged element
// and, on gfx950, every VALU instruction is issue time taken from the fp32 MFMA stream.
__device__ __forceinline__ float swish_fast(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

// Pipeline per CK-channel chunk (one LDS buffer, two barriers):
//   [registers of chunk c hold the prefetched patch + weight slab]
//   transform (GroupNorm affine + Swish) and store them to LDS | barrier | issue the global loads of chunk c+1 (they stay
//   in flight under the MFMAs) | taps x k-pairs of MFMAs, operands prefetched from LDS one step ahead | barrier
// SPEC = 1: the 3x3 / stride-1 / 8x32-pixel-tile case (every heavy conv of the U-Net).  Patch geometry is then a
// compile-time constant (row stride 35, plane 350), the (tap, k-pair) loop is fully unrolled and every LDS operand address
// is base register + immediate: no VALU instruction is issued between the MFMAs (each one would cost MFMA issue time).
// TWOM: both 32-channel M tiles are computed (any launch with Cout > 32).  The packed weights are zero padded to 64-channel
// blocks, so the second tile is always safe to compute and rows >= Cout are simply not stored; keeping this a template
// parameter (not a runtime test) matters: a branch inside the unrolled MFMA loop stops the compiler from hoisting the LDS
// operand reads across k-steps (it does not change the measured rate: the kernel is not bound by those reads).
template <int WN, int CK, int NXS, int NWS, int SPEC, bool TWOM>
__global__ __launch_bounds__(NTHREADS, 2) void conv_igemm_kernel(const ConvK p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sX = smem;                       // [CK][PLANE]
  float* sW = smem + p.XFLOATS;           // [ntaps][CK][BM]
  float* sG = sW + p.WFLOATS;             // [2][Cin]: GroupNorm scale | shift of this sample
  float* sE = sG + 2 * p.Cin;             // [2][BM]: bias | per-sample vector of the block's channels (the uniform epilogue)

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int b = blockIdx.z / p.ksplit, kslice = blockIdx.z - b * p.ksplit;
  const int co0 = blockIdx.y * BM;
  const int tile_y = blockIdx.x / p.tiles_x, tile_x = blockIdx.x - tile_y * p.tiles_x;
  const int TWm1 = (1 << p.tw_log2) - 1;
  const int vy0 = tile_y * p.TH, vx0 = tile_x << p.tw_log2;
  const int iy0 = vy0 * p.in_stride + p.dy_min, ix0 = vx0 * p.in_stride + p.dx_min;
  const bool has_gn = p.gn_scale != nullptr;
  const size_t HW = (size_t)p.H * p.W;
  const int c_begin = kslice * p.chunks_per_split * CK;
  const int c_end = min(p.CinPad, c_begin + p.chunks_per_split * CK);

  // ---- per-thread staging slots, fixed for the whole kernel (the patch geometry is the same for every chunk)
  int x_soff[NXS];     // ci*H*W + iy*W + ix inside the chunk's first plane, or -1 when the position is zero padding
  int x_meta[NXS];     // LDS float offset | ci << 24
  // SPEC: every staging thread serves ONE channel of the chunk (32 threads per channel, elements m = tid % 32 + 32 i of its
  // 340-element patch), so the per-slot state is an image offset and an LDS offset, the "outside the image" test is one bit
  // of a mask, and the GroupNorm scale / shift of the thread are two registers per chunk.  (PMC, round 2: the generic slot
  // scheme below costs 2.3 VALU instructions per MFMA -- 13 % of the kernel's time, all of it taken from the matrix pipe.)
  const int s_ci = tid >> 5;
  // SPEC: slot i is zero padding (outside the image) -- its load goes to a safe offset and its value is replaced by 0.  One
  // bool per slot: the compiler keeps each as a lane mask in a scalar register pair, so the replacement is ONE v_cndmask per
  // element (packed into a bit mask they cost a shift, an and and a compare per element on top -- VALU issue time that the
  // fp32 MFMA stream cannot overlap).  Slots 0..9 of the 340-element patch exist for every lane, slot 10 for lanes < 20.
  constexpr int SPEC_SLOTS = 11;
  bool s_outb[SPEC_SLOTS];
  const bool s_last = (tid & 31) < 340 - 320;
  if constexpr (SPEC == 1) {
    static_assert(NXS >= SPEC_SLOTS, "the 3x3 patch needs 11 slots per thread");
    const int safe = vy0 * p.W + vx0;
#pragma unroll
    for (int i = 0; i < SPEC_SLOTS; ++i) {
      const int m = (tid & 31) + 32 * i;
      const int py = m / 34, px = m - py * 34;
      const int iy = iy0 + py, ix = ix0 + px;
      const bool in_patch = m < 340;
      const bool in_img = in_patch && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      x_soff[i] = 4 * ((in_img ? iy * p.W + ix : safe) + s_ci * (int)HW);     // BYTE offset inside the chunk (buffer loads)
      x_meta[i] = s_ci * 350 + py * 35 + px;
      s_outb[i] = !in_img;
    }
  } else {
    const int plane_elems = p.PH * p.PW;
    const int pw = p.PW;
#pragma unroll
    for (int i = 0; i < NXS; ++i) {
      const int e = tid + i * NTHREADS;
      const int ci = e / plane_elems;
      const int rem = e - ci * plane_elems;
      const int py = rem / pw, px = rem - py * pw;
      const int iy = iy0 + py, ix = ix0 + px;
      const bool in_slot = (i < p.nx) && (ci < CK);
      const bool in_img = in_slot && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      x_soff[i] = in_img ? (int)(ci * HW) + iy * p.W + ix : -1;
      x_meta[i] = in_slot ? ((ci * p.PLANE + py * p.PWp + px) | (ci << 24)) : -1;
    }
  }
  int w_goff[NWS];     // float offset of the slot's 16-byte vector inside the chunk's weight slab, or -1
#pragma unroll
  for (int i = 0; i < NWS; ++i) {
    const int idx = tid + i * NTHREADS;
    const int row = idx >> 4, seg = idx & 15;
    const int tap = row / CK, ci = row - tap * CK;
    w_goff[i] = (i < p.nw && tap < p.ntaps) ? (tap * p.CinPad + ci) * p.CoutPad + seg * 4 : -1;
  }
  if (has_gn) {
    for (int i = tid; i < 2 * p.Cin; i += NTHREADS)
      sG[i] = (i < p.Cin) ? p.gn_scale[b * p.Cin + i] : p.gn_shift[b * p.Cin + (i - p.Cin)];
  }
  if (tid < 2 * BM) {
    const int co = co0 + (tid & (BM - 1));
    const float* src = (tid < BM) ? p.bias : p.addvec;
    sE[tid] = (src != nullptr && co < p.Cout) ? src[(tid < BM ? 0 : (size_t)b * p.Cout) + co] : 0.f;
  }
  // tap offsets live across the lanes of one register: read back with v_readlane (no memory access in the MFMA loop)
  const int tapv = (lane < p.ntaps) ? p.tap_off[lane] : 0;

  int pixoff[WN];
#pragma unroll
  for (int nt = 0; nt < WN; ++nt) {
    const int pidx = (wave * WN + nt) * 32 + l31;
    // tile rows past the staged patch (only when the tile is taller than the whole image) alias its last row: never stored
    pixoff[nt] = (min(pidx >> p.tw_log2, p.rows - 1) * p.PWp + (pidx & TWm1)) * p.in_stride;
  }

  f32x16 acc[2][WN];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < WN; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  constexpr bool two_m = TWOM;

  float xr[NXS];
  float4 wr[NWS];
  // SPEC: buffer resources of this sample's input tensors (range = the sample's channels: reads past Cin return 0)
  const __amdgpu_buffer_rsrc_t rsrc0 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.x0 + (size_t)b * p.C0 * HW), 0, p.C0 * (int)HW * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc1 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.x1 ? p.x1 + (size_t)b * p.C1 * HW : p.x0), 0, p.x1 ? p.C1 * (int)HW * 4 : 0, 0x00020000);
  auto issue_loads = [&](int c0) {
    // a chunk never straddles the concat seam (C0 % CK == 0 is checked on the host)
    const float* xbase = (c0 < p.C0) ? p.x0 + ((size_t)b * p.C0 + c0) * HW : p.x1 + ((size_t)b * p.C1 + (c0 - p.C0)) * HW;
    const int c_left = p.Cin - c0;     // channels of this chunk that exist
    if constexpr (SPEC == 1) {
      // buffer loads: resource = this sample's x0 (or x1) tensor, vector offset = the slot's fixed byte offset, SCALAR offset =
      // the chunk -- no vector address arithmetic per load (a 64-bit multiply-add per slot with flat pointers).  A channel
      // past Cin (Cin % 8 != 0) lies beyond the resource and reads as 0; its GroupNorm scale / shift are zeroed below.
      const bool first = c0 < p.C0;
      const int soff = (first ? c0 : c0 - p.C0) * (int)HW * 4;
#pragma unroll
      for (int i = 0; i < SPEC_SLOTS; ++i)
        xr[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(first ? rsrc0 : rsrc1, x_soff[i], soff, 0));
    } else {
#pragma unroll
      for (int i = 0; i < NXS; ++i) {
        const bool ok = x_soff[i] >= 0 && (x_meta[i] >> 24) < c_left;
        xr[i] = ok ? xbase[x_soff[i]] : 0.f;
      }
    }
    const float* wbase = p.wp + (size_t)c0 * p.CoutPad + co0;
#pragma unroll
    for (int i = 0; i < NWS; ++i)
      wr[i] = (w_goff[i] >= 0) ? *reinterpret_cast<const float4*>(wbase + w_goff[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  auto store_staged = [&](int c0) {
    const int c_left = p.Cin - c0;
    if constexpr (SPEC == 1) {
      const bool live = s_ci < c_left;        // a channel past Cin was read as 0 (buffer range) and must stay 0 after Swish
      if (has_gn) {
        const float gsc = live ? sG[c0 + s_ci] : 0.f, gsh = live ? sG[p.Cin + c0 + s_ci] : 0.f;
#pragma unroll
        for (int i = 0; i < SPEC_SLOTS; ++i) {
          const float v = swish_fast(fmaf(xr[i], gsc, gsh));
          if (i < SPEC_SLOTS - 1 || s_last) sX[x_meta[i]] = s_outb[i] ? 0.f : v;     // the conv pads the ACTIVATED tensor with zeros
        }
      } else {
#pragma unroll
        for (int i = 0; i < SPEC_SLOTS; ++i)
          if (i < SPEC_SLOTS - 1 || s_last) sX[x_meta[i]] = s_outb[i] ? 0.f : xr[i];
      }
    } else {
#pragma unroll
      for (int i = 0; i < NXS; ++i) {
        if (x_meta[i] >= 0) {
          float v = xr[i];
          const int ci = x_meta[i] >> 24;
          if (has_gn && x_soff[i] >= 0 && ci < c_left) v = swish_fast(fmaf(v, sG[c0 + ci], sG[p.Cin + c0 + ci]));
          sX[x_meta[i] & 0xFFFFFF] = v;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NWS; ++i)
      if (w_goff[i] >= 0) reinterpret_cast<float4*>(sW)[tid + i * NTHREADS] = wr[i];
  };

  constexpr int KH2 = CK / 2;
  const int nsteps = p.ntaps * KH2;
  const float* sXh = sX + h * p.PLANE;
  const float* sWh = sW + h * BM + l31;
  const int plane2 = 2 * p.PLANE;

  issue_loads(c_begin);
  __syncthreads();     // sG visible
  for (int c0 = c_begin; c0 < c_end; c0 += CK) {
    store_staged(c0);
    __syncthreads();
    if (c0 + CK < c_end) issue_loads(c0 + CK);

    // ---- MFMA over (tap, k-pair) steps.  Lane half h supplies k = 2*k2 + h for both operands.
    auto mma = [&](float a0, float a1, const float (&bf)[WN]) {
#pragma unroll
      for (int nt = 0; nt < WN; ++nt) acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bf[nt], acc[0][nt], 0, 0, 0);
      if (two_m) {
#pragma unroll
        for (int nt = 0; nt < WN; ++nt) acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bf[nt], acc[1][nt], 0, 0, 0);
      }
    };
    if constexpr (SPEC == 1) {
      constexpr int PWP = 35, PL = 350;          // (8 + 2) x (32 + 2) patch, odd row stride
      // Every operand read below is ONE ds_read_b32 with an immediate offset from its own base register.  The bases are made
      // opaque to the compiler on purpose: when it can relate two of them it fuses the reads into ds_read2_b32, whose 8-bit
      // offsets do not reach across the slab, and pays one v_add per pair to rebase -- VALU issue time that the fp32 MFMA
      // stream cannot overlap (2-3 such adds per k-step in round 1's build).
      int wo0 = 0, wo1 = 32, xo_[WN];
      asm volatile("" : "+v"(wo0));
      asm volatile("" : "+v"(wo1));
#pragma unroll
      for (int nt = 0; nt < WN; ++nt) {
        xo_[nt] = pixoff[nt];
        asm volatile("" : "+v"(xo_[nt]));
      }
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
        for (int k2 = 0; k2 < CK / 2; ++k2) {
          const int xo = (tap / 3) * PWP + (tap % 3) + k2 * 2 * PL;
          const int wo = (tap * CK + 2 * k2) * BM;
          float bf[WN];
#pragma unroll
          for (int nt = 0; nt < WN; ++nt) bf[nt] = sXh[xo_[nt] + xo];
          mma(sWh[wo0 + wo], sWh[wo1 + wo], bf);
        }
      }
    } else {
      float a0A, a1A, a0B, a1B, bA[WN], bB[WN];
      auto frag = [&](int step, float& a0, float& a1, float (&bf)[WN]) {
        const int tap = step / KH2, k2 = step - tap * KH2;
        const float* xb = sXh + __builtin_amdgcn_readlane(tapv, tap) + k2 * plane2;
        const float* wb = sWh + (tap * CK + 2 * k2) * BM;
        a0 = wb[0];
        a1 = wb[32];
#pragma unroll
        for (int nt = 0; nt < WN; ++nt) bf[nt] = xb[pixoff[nt]];
      };
      frag(0, a0A, a1A, bA);
      for (int step = 0; step < nsteps; step += 2) {        // nsteps is even (CK/2 is 2 or 4)
        frag(step + 1, a0B, a1B, bB);
        mma(a0A, a1A, bA);
        frag(step + 2 < nsteps ? step + 2 : step, a0A, a1A, bA);
        mma(a0B, a1B, bB);
      }
    }
    __syncthreads();
  }

  // ---- epilogue: + bias + per-sample channel vector + residual, NCHW store (32 consecutive pixels per register row).
  // Address arithmetic is kept out of the per-element path: one 64-bit base per (pixel tile, tensor), then 32-bit element
  // offsets that advance by the plane size from one accumulator register to the next (the channel rows of a 32x32 tile are
  // 0,1,2,3, 8,9,10,11, ... + 4h).  Round 2's ISA count of the previous form (a 64-bit multiply-add chain and a bounds test per
  // element) was ~25 VALU instructions per output element -- 0.7 per MFMA of a 128-channel conv, none of it overlappable.
  const bool full_block = (co0 + BM <= p.Cout);      // every channel row of the block exists (all but the 3-channel tail conv)
  const int co_lim = p.Cout - co0;
  if constexpr (TWOM) {
    // Whole tile inside the output, whole channel block, one K slice (a workgroup-uniform test): no per-lane control flow, so
    // the residual / bias / vector loads of ALL of the wave's accumulator registers are in flight together before the first
    // add (the per-pixel-tile form below waits for one tile's loads, stores, and only then starts the next tile's loads).
    if (full_block && p.ksplit == 1 && vy0 + p.TH <= p.VH && vx0 + TWm1 < p.VW) {
      const int plane = p.OH * p.OW;
      const bool has_bias = p.bias != nullptr, has_vec = p.addvec != nullptr;
      size_t blk[WN];
#pragma unroll
      for (int nt = 0; nt < WN; ++nt) {
        const int pidx = (wave * WN + nt) * 32 + l31;
        const int vy = vy0 + (pidx >> p.tw_log2), vx = vx0 + (pidx & TWm1);
        blk[nt] = ((size_t)b * p.Cout + co0) * (size_t)plane + (vy * p.out_sy + p.out_oy) * p.OW + (vx * p.out_sx + p.out_ox);
      }
      float res[WN][2][16];
      if (p.residual) {
#pragma unroll
        for (int nt = 0; nt < WN; ++nt) {
          const float* rbase = p.residual + blk[nt];
          int rel = 4 * h * plane;
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              res[nt][mt][r] = rbase[rel];
              rel += ((r & 3) == 3) ? 5 * plane : plane;
            }
        }
      }
      // bias and vector of the block's channel rows from LDS (staged with the GroupNorm table): same order of additions as the
      // general form -- (acc + bias) + vector, then + residual
      const float* sEh = sE + 4 * h;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = mt * 32 + (r & 3) + 8 * (r >> 2);
          const float bv = sEh[row], vv = sEh[BM + row];
#pragma unroll
          for (int nt = 0; nt < WN; ++nt) {
            float v = acc[mt][nt][r];
            if (has_bias) v += bv;
            if (has_vec) v += vv;
            if (p.residual) v += res[nt][mt][r];
            acc[mt][nt][r] = v;
          }
        }
#pragma unroll
      for (int nt = 0; nt < WN; ++nt) {
        float* obase = p.out + blk[nt];
        int rel = 4 * h * plane;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            obase[rel] = acc[mt][nt][r];
            rel += ((r & 3) == 3) ? 5 * plane : plane;
          }
      }
      return;
    }
  }
#pragma unroll
  for (int nt = 0; nt < WN; ++nt) {
    const int pidx = (wave * WN + nt) * 32 + l31;
    const int vy = vy0 + (pidx >> p.tw_log2), vx = vx0 + (pidx & TWm1);
    if (vy >= p.VH || vx >= p.VW) continue;
    const bool split = p.ksplit > 1;
    // split-K: raw partial sums over this slice's channels, compact [kslice][b][co][virtual pixel]
    const int plane = split ? p.VH * p.VW : p.OH * p.OW;                    // < 2^25 (checked on the host): 64 rows fit 32 bits
    const int pix = split ? vy * p.VW + vx : (vy * p.out_sy + p.out_oy) * p.OW + (vx * p.out_sx + p.out_ox);
    const size_t blk = ((split ? (size_t)kslice * p.B + b : (size_t)b) * p.Cout + co0) * (size_t)plane + pix;
    float* obase = (split ? p.partial : p.out) + blk;
    const float* rbase = (!split && p.residual) ? p.residual + blk : nullptr;
    const float* bias_h = (!split && p.bias) ? p.bias + co0 + 4 * h : nullptr;
    const float* vec_h = (!split && p.addvec) ? p.addvec + (size_t)b * p.Cout + co0 + 4 * h : nullptr;
    int rel = 4 * h * plane;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      if (mt == 1 && !two_m) break;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = mt * 32 + (r & 3) + 8 * (r >> 2);                   // + 4h: the channel row inside the block
        if (full_block || row + 4 * h < co_lim) {
          float v = acc[mt][nt][r];
          if (bias_h) v += bias_h[row];
          if (vec_h) v += vec_h[row];
          if (rbase) v += rbase[rel];
          obase[rel] = v;
        }
        rel += ((r & 3) == 3) ? 5 * plane : plane;                           // next register: +1 row, or +5 rows after every fourth
      }
    }
  }
}

// Sum the split-K partials in slice order and apply the epilogue (bias, per-sample vector, residual).
__global__ void conv_splitk_reduce_kernel(const ConvK p) {
  const size_t vplane = (size_t)p.VH * p.VW;
  const size_t n = (size_t)p.B * p.Cout * vplane;
  const size_t slice = n;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int vpix = (int)(i % vplane);
    const size_t bc = i / vplane;
    const int co = (int)(bc % p.Cout), b = (int)(bc / p.Cout);
    float v = 0.f;
    for (int k = 0; k < p.ksplit; ++k) v += p.partial[k * slice + i];
    if (p.bias) v += p.bias[co];
    if (p.addvec) v += p.addvec[b * p.Cout + co];
    const int vy = vpix / p.VW, vx = vpix - vy * p.VW;
    const size_t o = (bc * p.OH + (size_t)(vy * p.out_sy + p.out_oy)) * p.OW + (vx * p.out_sx + p.out_ox);
    if (p.residual) v += p.residual[o];
    p.out[o] = v;
  }
}

struct PackK {
  int mode, Cout, Cin, KH, KW, ntaps, CinPad, CoutPad, accumulate;
  int ky[HDIFF_MAX_TAPS];
  int kx[HDIFF_MAX_TAPS];
};

__global__ void pack_conv_weight_kernel(const float* __restrict__ w, float* __restrict__ wp, const PackK p) {
  const size_t n = (size_t)p.ntaps * p.CinPad * p.CoutPad;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int co = (int)(i % p.CoutPad);
    const size_t r = i / p.CoutPad;
    const int ci = (int)(r % p.CinPad);
    const int tap = (int)(r / p.CinPad);
    float v = 0.f;
    const int ky = p.ky[tap], kx = p.kx[tap];
    if (co < p.Cout && ci < p.Cin && ky >= 0) {
      const size_t src = (p.mode == 0) ? (((size_t)co * p.Cin + ci) * p.KH + ky) * p.KW + kx
                                       : (((size_t)ci * p.Cout + co) * p.KH + ky) * p.KW + kx;
      v = w[src];
    }
    wp[i] = p.accumulate ? wp[i] + v : v;
  }
}

int ceil_log2(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

template <int WN, int CK, int NXS, int NWS, int SPEC, bool TWOM>
void launch_t(const ConvK& k, dim3 grid, size_t lds_bytes, hipStream_t stream) {
  static uint64_t attr_mask = 0;
  if (first_use_on_device(attr_mask)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<WN, CK, NXS, NWS, SPEC, TWOM>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  hipLaunchKernelGGL((conv_igemm_kernel<WN, CK, NXS, NWS, SPEC, TWOM>), grid, dim3(NTHREADS), lds_bytes, stream, k);
}

template <int WN, int CK, int NXS, int NWS, int SPEC>
int launch(const ConvK& k, int B, size_t lds_bytes, hipStream_t stream) {
  const int BN = 128 * WN;
  const int TW = 1 << k.tw_log2;
  const int tiles_y = cdiv(k.VH, BN / TW);
  dim3 grid(k.tiles_x * tiles_y, cdiv(k.Cout, BM), B * k.ksplit);
  (void)hipGetLastError();  // drop any stale error left by another HIP user in this thread
  if (k.Cout > 32) launch_t<WN, CK, NXS, NWS, SPEC, true>(k, grid, lds_bytes, stream);
  else launch_t<WN, CK, NXS, NWS, SPEC, false>(k, grid, lds_bytes, stream);
  if (k.ksplit > 1) {
    const size_t n = (size_t)B * k.Cout * k.VH * k.VW;
    const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, k);
  }
  HDIFF_CHECK_LAUNCH("conv_igemm_kernel");
  return HDIFF_OK;
}

}  // namespace

extern "C" int hdiff_pack_conv_weight(const float* w, float* wp, int mode, int Cout, int Cin, int KH, int KW, int ntaps,
                                      const int* tap_ky, const int* tap_kx, int CinPad, int CoutPad, int accumulate,
                                      hdiff_stream_t stream) {
  HDIFF_CHECK_ARG(w && wp && tap_ky && tap_kx, "pack_conv_weight: null pointer");
  HDIFF_CHECK_ARG(ntaps >= 1 && ntaps <= HDIFF_MAX_TAPS, "pack_conv_weight: ntaps %d out of range", ntaps);
  HDIFF_CHECK_ARG(CinPad >= Cin && CinPad % 8 == 0 && CoutPad >= Cout && CoutPad % 64 == 0,
                  "pack_conv_weight: CinPad %% 8 / CoutPad %% 64 violated (%d, %d)", CinPad, CoutPad);
  PackK p{};
  p.mode = mode; p.Cout = Cout; p.Cin = Cin; p.KH = KH; p.KW = KW; p.ntaps = ntaps;
  p.CinPad = CinPad; p.CoutPad = CoutPad; p.accumulate = accumulate;
  for (int t = 0; t < ntaps; ++t) {
    HDIFF_CHECK_ARG(tap_ky[t] < KH && tap_kx[t] < KW, "pack_conv_weight: tap %d outside the kernel", t);
    p.ky[t] = tap_ky[t];
    p.kx[t] = tap_kx[t];
  }
  const size_t n = (size_t)ntaps * CinPad * CoutPad;
  const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  (void)hipGetLastError();  // drop any stale error left by another HIP user in this thread
  hipLaunchKernelGGL(pack_conv_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, wp, p);
  HDIFF_CHECK_LAUNCH("pack_conv_weight_kernel");
  return HDIFF_OK;
}

namespace {

struct ConvCfg {
  ConvK k;
  int WN, CK;
  size_t lds;
  int64_t splitk_floats;   // workspace the split-K path wants (0: no split)
};

// Geometry, tile configuration and split-K policy of one launch (shared by the workspace query and the launcher).
int configure(const hdiff_conv_desc* d, ConvCfg& c) {
  HDIFF_CHECK_ARG(d && d->x0 && d->wp && d->out, "conv2d_fwd: null pointer");
  HDIFF_CHECK_ARG(d->C1 == 0 || d->x1, "conv2d_fwd: C1 > 0 without x1");
  HDIFF_CHECK_ARG(d->ntaps >= 1 && d->ntaps <= HDIFF_MAX_TAPS, "conv2d_fwd: ntaps %d out of range", d->ntaps);
  HDIFF_CHECK_ARG(d->CinPad >= d->C0 + d->C1 && d->CinPad % 8 == 0 && d->CoutPad >= d->Cout && d->CoutPad % 64 == 0,
                  "conv2d_fwd: padded channel counts invalid (Cin %d CinPad %d Cout %d CoutPad %d)", d->C0 + d->C1,
                  d->CinPad, d->Cout, d->CoutPad);
  HDIFF_CHECK_ARG(d->B > 0 && d->H > 0 && d->W > 0 && d->VH > 0 && d->VW > 0 && d->in_stride >= 1 && d->in_stride <= 2,
                  "conv2d_fwd: bad geometry");
  HDIFF_CHECK_ARG((d->VH - 1) * d->out_sy + d->out_oy < d->OH && (d->VW - 1) * d->out_sx + d->out_ox < d->OW,
                  "conv2d_fwd: virtual grid maps outside the output tensor");
  HDIFF_CHECK_ARG((d->gn_scale == nullptr) == (d->gn_shift == nullptr), "conv2d_fwd: gn_scale/gn_shift must come together");
  HDIFF_CHECK_ARG((long)d->OH * d->OW < (1L << 25) && (long)d->VH * d->VW < (1L << 25),
                  "conv2d_fwd: planes of 2^25 pixels or more are not supported (32-bit element offsets inside a channel block)");

  ConvK& k = c.k;
  k = ConvK{};
  k.x0 = d->x0; k.x1 = d->x1; k.C0 = d->C0; k.C1 = d->C1; k.Cin = d->C0 + d->C1; k.H = d->H; k.W = d->W;
  k.wp = d->wp; k.CinPad = d->CinPad; k.CoutPad = d->CoutPad; k.Cout = d->Cout;
  k.bias = d->bias; k.gn_scale = d->gn_scale; k.gn_shift = d->gn_shift; k.addvec = d->addvec; k.residual = d->residual;
  k.out = d->out; k.OH = d->OH; k.OW = d->OW; k.VH = d->VH; k.VW = d->VW; k.in_stride = d->in_stride;
  k.out_sy = d->out_sy; k.out_oy = d->out_oy; k.out_sx = d->out_sx; k.out_ox = d->out_ox; k.ntaps = d->ntaps;
  k.B = d->B;

  int dy_min = d->tap_dy[0], dy_max = d->tap_dy[0], dx_min = d->tap_dx[0], dx_max = d->tap_dx[0];
  for (int t = 1; t < d->ntaps; ++t) {
    dy_min = d->tap_dy[t] < dy_min ? d->tap_dy[t] : dy_min;
    dy_max = d->tap_dy[t] > dy_max ? d->tap_dy[t] : dy_max;
    dx_min = d->tap_dx[t] < dx_min ? d->tap_dx[t] : dx_min;
    dx_max = d->tap_dx[t] > dx_max ? d->tap_dx[t] : dx_max;
  }
  k.dy_min = dy_min; k.dx_min = dx_min;

  c.WN = ((long)d->VH * d->VW >= 1024) ? 2 : 1;
  const int BN = 128 * c.WN;
  int twl = ceil_log2(d->VW);
  if (twl > 5) twl = 5;
  k.tw_log2 = twl;
  const int TW = 1 << twl;
  k.TH = BN / TW;
  k.tiles_x = cdiv(d->VW, TW);
  k.rows = k.TH < d->VH ? k.TH : d->VH;
  k.PH = (k.rows - 1) * d->in_stride + (dy_max - dy_min + 1);
  k.PW = (TW - 1) * d->in_stride + (dx_max - dx_min + 1);
  k.PWp = k.PW | 1;
  k.PLANE = k.PH * k.PWp;
  for (int t = 0; t < d->ntaps; ++t) k.tap_off[t] = (d->tap_dy[t] - dy_min) * k.PWp + (d->tap_dx[t] - dx_min);

  // configuration A: CK = 8 channels per chunk, <= 12 activation slots and <= 5 weight vectors per thread
  // configuration B: CK = 4, <= 20 / <= 7 (5x5 stride-2 patches, very narrow images)
  const int Cin = d->C0 + d->C1;
  auto slots_x = [&](int ck) { return cdiv(ck * k.PH * k.PW, NTHREADS); };
  auto slots_w = [&](int ck) { return cdiv(d->ntaps * ck * (BM / 4), NTHREADS); };
  auto lds_for = [&](int ck) {
    const int xfl = (ck * k.PLANE + 3) & ~3;
    return (size_t)(xfl + d->ntaps * ck * BM + 2 * Cin + 2 * BM) * sizeof(float);
  };
  c.CK = 0;
  if (slots_x(8) <= 12 && slots_w(8) <= 5 && lds_for(8) <= 64 * 1024 && (d->C1 == 0 || d->C0 % 8 == 0)) c.CK = 8;
  else if (slots_x(4) <= 20 && slots_w(4) <= 7 && lds_for(4) <= 96 * 1024 && (d->C1 == 0 || d->C0 % 4 == 0)) c.CK = 4;
  HDIFF_CHECK_ARG(c.CK != 0, "conv2d_fwd: no kernel configuration fits (taps %d, patch %dx%d, C0 %d)", d->ntaps, k.PH, k.PW,
                  d->C0);
  c.lds = lds_for(c.CK);
  k.XFLOATS = (c.CK * k.PLANE + 3) & ~3;
  k.WFLOATS = d->ntaps * c.CK * BM;
  k.nx = slots_x(c.CK);
  k.nw = slots_w(c.CK);

  // split-K policy: a small grid with a long channel loop is cut into channel slices (more workgroups, shorter loops).
  // The decision looks at the whole launch (batch included): a large batch already fills the chip, and splitting it
  // would only add partial-sum traffic.  Consequence: a sample's last bits may depend on the batch it is computed in
  // (summation order), never on anything else -- for a fixed shape the result is bitwise reproducible.
  const int chunks = d->CinPad / c.CK;
  const long blocks = (long)k.tiles_x * cdiv(d->VH, k.TH) * cdiv(d->Cout, BM) * d->B;
  int ksplit = 1;
  if (blocks < 192 && chunks >= 8) {
    ksplit = (int)((512 + blocks - 1) / blocks);
    if (ksplit > chunks / 2) ksplit = chunks / 2;
    if (ksplit > 32) ksplit = 32;
  }
  k.chunks_per_split = cdiv(chunks, ksplit);
  k.ksplit = cdiv(chunks, k.chunks_per_split);
  c.splitk_floats = (k.ksplit > 1) ? (int64_t)k.ksplit * d->B * d->Cout * d->VH * d->VW : 0;
  return HDIFF_OK;
}

}  // namespace

namespace { bool is_direct_1x1(const hdiff_conv_desc* d); bool is_x3_conv(const hdiff_conv_desc* d); bool is_x3_1x1(const hdiff_conv_desc* d); }

extern "C" int hdiff_conv2d_fwd_workspace(const hdiff_conv_desc* d, int64_t* floats_out) {
  HDIFF_CHECK_ARG(floats_out, "conv2d_fwd_workspace: null pointer");
  ConvCfg c;
  const int rc = configure(d, c);
  if (rc != HDIFF_OK) return rc;
  *floats_out = (is_direct_1x1(d) || is_x3_conv(d)) ? 0 : c.splitk_floats;
  return HDIFF_OK;
}

namespace hdiff {
struct Conv1x1K {
  const float* x0;
  const float* x1;
  int C0, Cin;
  long HW;
  const float* wp;
  int CoutPad, Cout;
  const float* bias;
  const float* addvec;
  const float* residual;
  float* out;
};
void launch_conv1x1_direct(const Conv1x1K& k, int B, hipStream_t stream);   // conv1x1_direct.hip
}  // namespace hdiff

namespace {
// Split-bf16 mode: a stride-1 conv whose taps all lie in the 3x3 neighbourhood -- the plain 3x3 / pad-1 conv (9 taps, output
// grid = input grid) and the output-parity phases of ConvTranspose2d(5, stride 2) (9 / 6 / 6 / 4 taps, output pixel
// (2y + py, 2x + px)) -- with 16-channel-aligned inputs and a launch large enough to fill the chip (small ones keep the
// split-K path of the fp32 kernel).
// The fp16-pair form of that kernel (conv3x3_x3.hip, PAIR): the plain 3x3 conv whose input range the caller knows -- behind the
// GroupNorm + Swish prologue (or an activation tensor made from it), bounded by the GroupNorm weights (act_scale from
// hdiff_gn_act_scale, weights from hdiff_pack_conv_weight_h2).
bool is_h2_conv_shape(const hdiff_conv_desc* d, bool same) {
  return same && d->ntaps == 9 && d->wp_h2 != nullptr && d->act_scale != nullptr;
}
bool is_x3_conv(const hdiff_conv_desc* d) {
  if ((d->wp_x3 == nullptr && d->wp_h2 == nullptr) || hdiff::contraction_mode() != HDIFF_CONTRACT_BF16X3) return false;
  if ((d->ntaps != 9 && d->ntaps != 6 && d->ntaps != 4) || d->in_stride != 1) return false;
  if (d->VH != d->H || d->VW != d->W) return false;
  const bool same = d->out_sy == 1 && d->out_oy == 0 && d->out_sx == 1 && d->out_ox == 0 && d->OH == d->H && d->OW == d->W;
  const bool phase = d->out_sy == 2 && d->out_sx == 2 && (d->out_oy == 0 || d->out_oy == 1) && (d->out_ox == 0 || d->out_ox == 1) &&
                     d->OH == 2 * d->H && d->OW == 2 * d->W && d->residual == nullptr;
  if (!same && !phase) return false;
  unsigned seen = 0;
  for (int t = 0; t < d->ntaps; ++t) {
    if (d->tap_dy[t] < -1 || d->tap_dy[t] > 1 || d->tap_dx[t] < -1 || d->tap_dx[t] > 1) return false;
    const unsigned bit = 1u << ((d->tap_dy[t] + 1) * 3 + (d->tap_dx[t] + 1));
    if (seen & bit) return false;                 // a tap listed twice: not a launch any of the x3 packs describes
    seen |= bit;
  }
  // the plain 3x3 pack (hdiff_pack_conv_weight_x3, forward and mirrored / transposed) stores tap t = (t / 3, t % 3): a
  // descriptor that lists its nine taps in another order keeps the fp32 kernel, which reads the order from the descriptor.
  // (Tap LISTS -- the transposed-conv phases -- are packed from the caller's own list by hdiff_pack_conv_weight_x3_taps;
  // the descriptor must list the taps in that order, include/hdiff.h.)
  if (same)
    for (int t = 0; t < 9; ++t)
      if (d->ntaps != 9 || d->tap_dy[t] != t / 3 - 1 || d->tap_dx[t] != t % 3 - 1) return false;
  const int Cin = d->C0 + d->C1;
  if (Cin % 16 != 0 || (d->C1 != 0 && d->C0 % 16 != 0) || Cin > 4096) return false;
  const long blocks = (long)cdiv(d->W, 32) * cdiv(d->H, 8) * cdiv(d->Cout, 64) * d->B;
  if (blocks < 192) return false;
  return d->wp_x3 != nullptr || is_h2_conv_shape(d, same);
}

// A plain 1x1 / stride-1 conv over a full-size output with no GroupNorm prologue and enough pixels to fill the chip goes to
// the LDS-free GEMM kernel (small grids keep the split-K path of the implicit-GEMM kernel).
bool is_direct_1x1(const hdiff_conv_desc* d) {
  return d->ntaps == 1 && d->tap_dy[0] == 0 && d->tap_dx[0] == 0 && d->in_stride == 1 && d->gn_scale == nullptr &&
         d->out_sy == 1 && d->out_oy == 0 && d->out_sx == 1 && d->out_ox == 0 && d->VH == d->H && d->VW == d->W &&
         d->OH == d->H && d->OW == d->W && (long)d->B * d->H * d->W >= 32768 && ((long)d->H * d->W) % 128 == 0 &&
         d->C0 % 2 == 0 && (long)(d->C0 + d->C1) * d->H * d->W < (1L << 30) && d->CinPad * d->CoutPad < (1 << 30);
}
// ... and in the split-bf16 mode, with a one-tap bf16-triple pack (hdiff_pack_conv_weight_x3_taps, ntaps = 1) and 16-channel-aligned
// inputs, the same GEMM runs on the bf16 MFMA (conv1x1_x3.hip).
bool is_x3_1x1(const hdiff_conv_desc* d) {
  const int Cin = d->C0 + d->C1;
  return d->wp_x3 != nullptr && hdiff::contraction_mode() == HDIFF_CONTRACT_BF16X3 && Cin % 16 == 0 &&
         (d->C1 == 0 || d->C0 % 16 == 0) && ((long)d->H * d->W) % 256 == 0 && d->CoutPad % 64 == 0;
}
}  // namespace

extern "C" int hdiff_conv2d_fwd(const hdiff_conv_desc* d, hdiff_stream_t stream) {
  ConvCfg c;
  const int rc = configure(d, c);
  if (rc != HDIFF_OK) return rc;
  ConvK& k = c.k;
  if (is_x3_conv(d)) {
    hdiff::ConvX3K q{};
    q.x0 = d->x0; q.x1 = d->x1; q.C0 = d->C0; q.C1 = d->C1; q.Cin = d->C0 + d->C1; q.H = d->H; q.W = d->W;
    q.wp3 = (const unsigned*)d->wp_x3; q.CoutPad = d->CoutPad; q.Cout = d->Cout;
    {
      const bool same = d->out_sy == 1 && d->out_oy == 0 && d->out_sx == 1 && d->out_ox == 0 && d->OH == d->H && d->OW == d->W;
      if (is_h2_conv_shape(d, same)) {         // fp16 pairs (three products) instead of bf16 triples (six)
        q.wp3 = (const unsigned*)d->wp_h2;
        q.act_scale = d->act_scale;
        q.w_scale = reinterpret_cast<const float*>(q.wp3 + (size_t)(q.Cin / 16) * 9 * 2 * d->CoutPad * 8);
        q.one = 1.0f;
      }
    }
    q.bias = d->bias; q.gn_scale = d->gn_scale; q.gn_shift = d->gn_shift; q.addvec = d->addvec; q.residual = d->residual;
    q.out = d->out; q.tiles_x = cdiv(d->W, 32); q.ntaps = d->ntaps;
    for (int t = 0; t < d->ntaps; ++t) q.tap_off[t] = ((d->tap_dy[t] + 1) * 34 + (d->tap_dx[t] + 1)) * 4;
    q.OH = d->OH; q.OW = d->OW; q.out_sy = d->out_sy; q.out_oy = d->out_oy; q.out_sx = d->out_sx; q.out_ox = d->out_ox;
    (void)hipGetLastError();  // drop any stale error left by another HIP user in this thread
    hdiff::launch_conv3x3_x3(q, d->B, (hipStream_t)stream);
    HDIFF_CHECK_LAUNCH("conv3x3_x3_kernel");
    return HDIFF_OK;
  }
  if (is_direct_1x1(d) && is_x3_1x1(d)) {          // bf16x3 mode: the same GEMM on bf16 triples (conv1x1_x3.hip)
    hdiff::Conv1x1X3K q{d->x0, d->x1, d->C0, d->C0 + d->C1, (long)d->H * d->W, (const unsigned*)d->wp_x3, d->CoutPad, d->Cout,
                        d->bias, d->addvec, d->residual, d->out};
    (void)hipGetLastError();  // drop any stale error left by another HIP user in this thread
    hdiff::launch_conv1x1_x3(q, d->B, (hipStream_t)stream);
    HDIFF_CHECK_LAUNCH("conv1x1_x3_kernel");
    return HDIFF_OK;
  }
  if (is_direct_1x1(d)) {
    hdiff::Conv1x1K q{d->x0, d->x1, d->C0, d->C0 + d->C1, (long)d->H * d->W, d->wp, d->CoutPad, d->Cout, d->bias, d->addvec,
                      d->residual, d->out};
    (void)hipGetLastError();  // drop any stale error left by another HIP user in this thread
    hdiff::launch_conv1x1_direct(q, d->B, (hipStream_t)stream);
    HDIFF_CHECK_LAUNCH("conv1x1_direct_kernel");
    return HDIFF_OK;
  }
  if (k.ksplit > 1 && d->splitk_ws != nullptr && d->splitk_floats >= c.splitk_floats) {
    k.partial = d->splitk_ws;
  } else {            // no (or too small a) workspace: one slice, the plain epilogue
    k.ksplit = 1;
    k.chunks_per_split = d->CinPad / c.CK;
    k.partial = nullptr;
  }
  hipStream_t s = (hipStream_t)stream;
  // the specialised 3x3 path: standard taps in row-major order, stride 1, 8x32 tile (patch stride 35, plane 350)
  bool spec = c.WN == 2 && c.CK == 8 && d->ntaps == 9 && d->in_stride == 1 && k.tw_log2 == 5 && k.PWp == 35 && k.PLANE == 350;
  for (int t = 0; spec && t < 9; ++t) spec = k.tap_off[t] == (t / 3) * 35 + (t % 3);
  // the specialised kernel addresses a sample's input tensor through a buffer resource with 32-bit byte offsets
  const long long hw = (long long)d->H * d->W;
  if (4 * hw * (d->C0 > d->C1 ? d->C0 : d->C1) >= (1ll << 31)) spec = false;
  if (spec) return launch<2, 8, 12, 5, 1>(k, d->B, c.lds, s);
  if (c.WN == 2 && c.CK == 8) return launch<2, 8, 12, 5, 0>(k, d->B, c.lds, s);
  if (c.WN == 2 && c.CK == 4) return launch<2, 4, 20, 7, 0>(k, d->B, c.lds, s);
  if (c.WN == 1 && c.CK == 8) return launch<1, 8, 12, 5, 0>(k, d->B, c.lds, s);
  return launch<1, 4, 20, 7, 0>(k, d->B, c.lds, s);
}
