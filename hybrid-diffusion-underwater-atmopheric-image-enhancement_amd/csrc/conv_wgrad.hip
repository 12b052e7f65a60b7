// Weight gradient of the implicit-GEMM convolution (conv_igemm.hip), fp32 MFMA 32x32x2, gfx950.
//
// Reference: autograd of nn.Conv2d / nn.ConvTranspose2d weights in TrainCondition.py:60 for the call sites listed in
// conv_igemm.hip.  With the forward's geometry (virtual output grid v, input coordinate v*stride + tap, output
// coordinate v*out_s + out_o) the packed gradient is
//     dWp[tap][ci][co] = sum_{b, v}  dY[b][co][out(v)] * act(X[b][ci][in(v) + tap])
// GEMM roles: M = 64 output channels, N = (input channel, tap) columns of a CKW-channel chunk, K = pixels.
// A workgroup owns one (co tile, ci chunk) and walks a strided share of all (sample, 128-pixel tile) pairs, keeping the
// whole M x N accumulator in registers (one wave per SIMD, up to 160 accumulator VGPRs); its four waves split the pixels of
// every tile, and their partial accumulators are summed through LDS at the end.  The activation patch (with the forward's
// GroupNorm-affine + Swish prologue recomputed on the fly) and the dY tile of the NEXT tile are prefetched into registers
// while the MFMAs of the current one run.  Splits write separate partial slabs; hdiff_conv_wgrad_unpack sums them in a fixed
// order into the PyTorch weight layout -- no atomics, bitwise reproducible.
#include <stdlib.h>

#include "common.h"

using namespace hdiff;

namespace {

constexpr int BM = 64;
constexpr int BNP = 128;              // pixels per tile
constexpr int DYROW = BNP + 1;        // LDS row stride of the dY tile (odd: conflict-free column reads)
constexpr int NTHREADS = 256;
constexpr int MAXNT = 5;              // N tiles of 32 columns
constexpr int NXS = 18;               // activation staging slots per thread
constexpr int NYS = BM * BNP / NTHREADS;   // dY staging slots per thread (32)

struct WgradK {
  const float* x0;
  const float* x1;
  int C0, C1, Cin, H, W;
  const float* gn_scale;
  const float* gn_shift;
  const float* dy;
  int Cout, OH, OW;
  int VH, VW, in_stride, out_sy, out_oy, out_sx, out_ox;
  int ntaps, dy_min, dx_min, PH, PW, PWp, PLANE;
  int tw_log2, TH, tiles_x, tiles_per_image, total_tiles;
  int rows;                        // patch rows actually staged: min(TH, VH)
  int CKW, ncol, nt_used, nx;
  int CinPad, CoutPad, nsplit;
  float* dwp;
  int tap_off[HDIFF_MAX_TAPS];
};

__device__ __forceinline__ float swish_fast_w(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

// ROWS = input stride (1 or 2) when the tile is 4 rows x 32 pixels (each wave then owns one tile row and every LDS operand
// address in the pixel loop is base + immediate: no VALU between the MFMAs); ROWS = 0 is the generic tile shape.
template <int ROWS>
__global__ __launch_bounds__(NTHREADS) void conv_wgrad_kernel(const WgradK p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sY = smem;                          // [BM][DYROW]; reused for the cross-wave reduction
  float* sX = smem + BM * DYROW;             // [CKW][PLANE]
  float* sG = sX + p.CKW * p.PLANE;          // [2][CKW] GroupNorm scale | shift of the current sample's chunk

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int co0 = blockIdx.x * BM;
  const int ci0 = blockIdx.y * p.CKW;
  const int split = blockIdx.z;
  const bool has_gn = p.gn_scale != nullptr;
  const size_t HW = (size_t)p.H * p.W;
  const size_t OHW = (size_t)p.OH * p.OW;
  const int TWm1 = (1 << p.tw_log2) - 1;

  // activation staging slots: (ci_local, py, px) fixed per thread
  int x_pos[NXS];      // ci << 20 | py << 10 | px, or -1
  {
    const int plane_elems = p.PH * p.PW;
#pragma unroll
    for (int i = 0; i < NXS; ++i) {
      const int e = tid + i * NTHREADS;
      const int ci = e / plane_elems;
      const int rem = e - ci * plane_elems;
      const int py = rem / p.PW, px = rem - py * p.PW;
      const bool ok = (i < p.nx) && (ci < p.CKW);
      x_pos[i] = ok ? ((ci << 20) | (py << 10) | px) : -1;
    }
  }
  // this lane's column (ci_local, tap) in each N tile -> LDS offset of the column's patch origin
  int coloff[MAXNT];
#pragma unroll
  for (int nt = 0; nt < MAXNT; ++nt) {
    const int j = nt * 32 + l31;
    const int cil = j / p.ntaps, tap = j - cil * p.ntaps;
    coloff[nt] = (j < p.ncol) ? cil * p.PLANE + p.tap_off[tap] : 0;   // padded columns read column 0 (discarded)
  }

  f32x16 acc[2][MAXNT];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < MAXNT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  float xr[NXS], yr[NYS];
  unsigned xlive = 0;   // bit i: slot i holds a real input element (not zero padding / a missing channel)
  // dY staging: element e = i*256 + tid -> pixel e % 128 (fixed per thread: consecutive lanes read consecutive pixels,
  // coalesced), channel e / 128 = 2*i + (tid >> 7)
  const int ypix = tid & (BNP - 1);
  const int yco0 = tid >> 7;

  auto tile_origin = [&](int t, int& b, int& vy0, int& vx0) {
    b = t / p.tiles_per_image;
    const int ti = t - b * p.tiles_per_image;
    const int ty = ti / p.tiles_x, tx = ti - ty * p.tiles_x;
    vy0 = ty * p.TH;
    vx0 = tx << p.tw_log2;
  };
  auto issue_loads = [&](int t) {
    int b, vy0, vx0;
    tile_origin(t, b, vy0, vx0);
    const int iy0 = vy0 * p.in_stride + p.dy_min, ix0 = vx0 * p.in_stride + p.dx_min;
    xlive = 0;
#pragma unroll
    for (int i = 0; i < NXS; ++i) {
      float v = 0.f;
      if (x_pos[i] >= 0) {
        const int c = ci0 + (x_pos[i] >> 20);
        const int iy = iy0 + ((x_pos[i] >> 10) & 1023), ix = ix0 + (x_pos[i] & 1023);
        if (c < p.Cin && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) {
          const float* src = (c < p.C0) ? p.x0 + ((size_t)b * p.C0 + c) * HW : p.x1 + ((size_t)b * p.C1 + (c - p.C0)) * HW;
          v = src[(size_t)iy * p.W + ix];
          xlive |= 1u << i;
        }
      }
      xr[i] = v;
    }
    const int vy = vy0 + (ypix >> p.tw_log2), vx = vx0 + (ypix & TWm1);
    const bool pix_ok = vy < p.VH && vx < p.VW;
    const float* ysrc = p.dy + ((size_t)b * p.Cout + co0 + yco0) * OHW +
                        (pix_ok ? (size_t)(vy * p.out_sy + p.out_oy) * p.OW + (vx * p.out_sx + p.out_ox) : 0);
    const int co_left = p.Cout - co0 - yco0;      // channel 2*i + yco0 exists iff 2*i < co_left
#pragma unroll
    for (int i = 0; i < NYS; ++i) yr[i] = (pix_ok && 2 * i < co_left) ? ysrc[(size_t)(2 * i) * OHW] : 0.f;
  };
  auto store_staged = [&](int t) {
    const int b = t / p.tiles_per_image;
    if (has_gn && tid < 2 * p.CKW) {
      const int ci = tid < p.CKW ? tid : tid - p.CKW;
      const int c = ci0 + ci;
      const float* tab = tid < p.CKW ? p.gn_scale : p.gn_shift;
      sG[tid] = (c < p.Cin) ? tab[b * p.Cin + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NXS; ++i) {
      if (x_pos[i] >= 0) {
        float v = xr[i];
        if (has_gn && ((xlive >> i) & 1u)) v = swish_fast_w(fmaf(v, sG[x_pos[i] >> 20], sG[p.CKW + (x_pos[i] >> 20)]));
        sX[(x_pos[i] >> 20) * p.PLANE + ((x_pos[i] >> 10) & 1023) * p.PWp + (x_pos[i] & 1023)] = v;
      }
    }
#pragma unroll
    for (int i = 0; i < NYS; ++i) sY[(2 * i + yco0) * DYROW + ypix] = yr[i];
  };

  int t = split;
  if (t < p.total_tiles) issue_loads(t);
  for (; t < p.total_tiles; t += p.nsplit) {
    __syncthreads();                 // previous tile's LDS reads are done
    store_staged(t);
    __syncthreads();
    if (t + p.nsplit < p.total_tiles) issue_loads(t + p.nsplit);

    // each wave takes a quarter of the tile's pixels; lane half h supplies pixel k = 2*kk + h
    const int kbeg = wave * (BNP / 4);
    if constexpr (ROWS > 0) {
      const float* yb = sY + l31 * DYROW + kbeg + h;
      const float* xb = sX + (wave * p.PWp + h) * ROWS;          // tile row `wave`, pixel h
#pragma unroll
      for (int kk = 0; kk < BNP / 8; ++kk) {
        const float a0 = yb[2 * kk];
        const float a1 = yb[32 * DYROW + 2 * kk];
#pragma unroll
        for (int nt = 0; nt < MAXNT; ++nt) {
          if (nt < p.nt_used) {
            const float bv = xb[coloff[nt] + 2 * kk * ROWS];
            acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv, acc[0][nt], 0, 0, 0);
            acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv, acc[1][nt], 0, 0, 0);
          }
        }
      }
    } else {
#pragma unroll 4
      for (int kk = 0; kk < BNP / 8; ++kk) {
        const int k = kbeg + 2 * kk + h;
        // rows past the staged patch (tile taller than the image) alias its last row: their dY is zero
        const int pixoff = (min(k >> p.tw_log2, p.rows - 1) * p.PWp + (k & TWm1)) * p.in_stride;
        const float a0 = sY[l31 * DYROW + k];
        const float a1 = sY[(32 + l31) * DYROW + k];
#pragma unroll
        for (int nt = 0; nt < MAXNT; ++nt) {
          if (nt < p.nt_used) {
            const float bv = sX[coloff[nt] + pixoff];
            acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv, acc[0][nt], 0, 0, 0);
            acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv, acc[1][nt], 0, 0, 0);
          }
        }
      }
    }
  }

  // ---- cross-wave sum through LDS, then the packed partial slab of this split
  float* red = sY;                                   // [4 waves][32 rows][33]
  float* slab = p.dwp + (size_t)split * p.ntaps * p.CinPad * p.CoutPad;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
    for (int nt = 0; nt < MAXNT; ++nt) {
      if (nt >= p.nt_used) continue;
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        red[(wave * 32 + row) * 33 + l31] = acc[mt][nt][r];
      }
      __syncthreads();
      for (int idx = tid; idx < 1024; idx += NTHREADS) {
        const int row = idx & 31, col = idx >> 5;    // consecutive threads -> consecutive output channels
        const float v = (red[(0 * 32 + row) * 33 + col] + red[(1 * 32 + row) * 33 + col]) +
                        (red[(2 * 32 + row) * 33 + col] + red[(3 * 32 + row) * 33 + col]);
        const int j = nt * 32 + col;
        if (j < p.ncol) {
          const int cil = j / p.ntaps, tap = j - cil * p.ntaps;
          const int ci = ci0 + cil, co = co0 + mt * 32 + row;
          if (ci < p.CinPad && co < p.CoutPad) slab[((size_t)tap * p.CinPad + ci) * p.CoutPad + co] = v;
        }
      }
    }
  }
}

struct UnpackK {
  int mode, Cout, Cin, KH, KW, ntaps, CinPad, CoutPad, nsplit, accumulate;
  int ky[HDIFF_MAX_TAPS];
  int kx[HDIFF_MAX_TAPS];
};

// dw (+)= sum over splits of the packed slabs, scattered back to the PyTorch layout (inverse of pack_conv_weight)
__global__ void conv_wgrad_unpack_kernel(const float* __restrict__ dwp, float* __restrict__ dw, const UnpackK p) {
  const size_t slab = (size_t)p.ntaps * p.CinPad * p.CoutPad;
  const size_t n = (size_t)p.ntaps * p.Cin * p.Cout;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int co = (int)(i % p.Cout);
    const size_t r = i / p.Cout;
    const int ci = (int)(r % p.Cin);
    const int tap = (int)(r / p.Cin);
    const int ky = p.ky[tap], kx = p.kx[tap];
    if (ky < 0) continue;
    const size_t src = ((size_t)tap * p.CinPad + ci) * p.CoutPad + co;
    float s = 0.f;
    for (int sp = 0; sp < p.nsplit; ++sp) s += dwp[sp * slab + src];
    const size_t dst = (p.mode == 0) ? (((size_t)co * p.Cin + ci) * p.KH + ky) * p.KW + kx
                                     : (((size_t)ci * p.Cout + co) * p.KH + ky) * p.KW + kx;
    dw[dst] = p.accumulate ? dw[dst] + s : s;
  }
}

int ceil_log2w(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

}  // namespace

extern "C" int hdiff_conv2d_wgrad_workspace(const hdiff_conv_wgrad_desc* d, int* nsplit_out, int64_t* floats_out) {
  HDIFF_CHECK_ARG(d && nsplit_out && floats_out, "conv2d_wgrad_workspace: null pointer");
  if (wgrad1x1_applicable(d)) {
    *nsplit_out = wgrad1x1_nsplit(d);
    *floats_out = (int64_t)*nsplit_out * d->ntaps * d->CinPad * d->CoutPad;
    return HDIFF_OK;
  }
  if (wgrad3x3_applicable(d)) {
    *nsplit_out = wgrad3x3_nsplit(d);
    *floats_out = (int64_t)*nsplit_out * d->ntaps * d->CinPad * d->CoutPad;
    return HDIFF_OK;
  }
  const int ckw = d->ntaps == 1 ? 32 : (d->ntaps > 9 ? 4 : 16);
  const int base = cdiv(d->Cout, BM) * cdiv(d->CinPad, ckw);
  int twl = ceil_log2w(d->VW);
  if (twl > 5) twl = 5;
  const int TW = 1 << twl, TH = BNP / TW;
  const int total = d->B * cdiv(d->VW, TW) * cdiv(d->VH, TH);
  int ns = cdiv(768, base);
  if (ns > total) ns = total;
  if (ns < 1) ns = 1;
  *nsplit_out = ns;
  *floats_out = (int64_t)ns * d->ntaps * d->CinPad * d->CoutPad;
  return HDIFF_OK;
}

extern "C" int hdiff_conv2d_wgrad(const hdiff_conv_wgrad_desc* d, float* dwp, int nsplit, hdiff_stream_t stream) {
  HDIFF_CHECK_ARG(d && d->x0 && d->dy && dwp, "conv2d_wgrad: null pointer");
  HDIFF_CHECK_ARG(d->C1 == 0 || d->x1, "conv2d_wgrad: C1 > 0 without x1");
  HDIFF_CHECK_ARG(d->ntaps >= 1 && d->ntaps <= HDIFF_MAX_TAPS, "conv2d_wgrad: ntaps %d out of range", d->ntaps);
  HDIFF_CHECK_ARG(d->CinPad >= d->C0 + d->C1 && d->CinPad % 8 == 0 && d->CoutPad >= d->Cout && d->CoutPad % 64 == 0,
                  "conv2d_wgrad: padded channel counts invalid");
  HDIFF_CHECK_ARG(d->B > 0 && d->VH > 0 && d->VW > 0 && d->in_stride >= 1 && d->in_stride <= 2 && nsplit >= 1,
                  "conv2d_wgrad: bad geometry");
  HDIFF_CHECK_ARG((d->gn_scale == nullptr) == (d->gn_shift == nullptr), "conv2d_wgrad: gn_scale/gn_shift must come together");
  if (wgrad1x1_applicable(d)) return launch_wgrad1x1(d, dwp, nsplit, (hipStream_t)stream);
  if (wgrad3x3_applicable(d)) return launch_wgrad3x3(d, dwp, nsplit, (hipStream_t)stream);

  WgradK k{};
  k.x0 = d->x0; k.x1 = d->x1; k.C0 = d->C0; k.C1 = d->C1; k.Cin = d->C0 + d->C1; k.H = d->H; k.W = d->W;
  k.gn_scale = d->gn_scale; k.gn_shift = d->gn_shift; k.dy = d->dy; k.Cout = d->Cout; k.OH = d->OH; k.OW = d->OW;
  k.VH = d->VH; k.VW = d->VW; k.in_stride = d->in_stride;
  k.out_sy = d->out_sy; k.out_oy = d->out_oy; k.out_sx = d->out_sx; k.out_ox = d->out_ox;
  k.ntaps = d->ntaps; k.CinPad = d->CinPad; k.CoutPad = d->CoutPad; k.nsplit = nsplit; k.dwp = dwp;

  int dy_min = d->tap_dy[0], dy_max = d->tap_dy[0], dx_min = d->tap_dx[0], dx_max = d->tap_dx[0];
  for (int t = 1; t < d->ntaps; ++t) {
    dy_min = d->tap_dy[t] < dy_min ? d->tap_dy[t] : dy_min;
    dy_max = d->tap_dy[t] > dy_max ? d->tap_dy[t] : dy_max;
    dx_min = d->tap_dx[t] < dx_min ? d->tap_dx[t] : dx_min;
    dx_max = d->tap_dx[t] > dx_max ? d->tap_dx[t] : dx_max;
  }
  k.dy_min = dy_min; k.dx_min = dx_min;
  int twl = ceil_log2w(d->VW);
  if (twl > 5) twl = 5;
  k.tw_log2 = twl;
  const int TW = 1 << twl;
  k.TH = BNP / TW;
  k.tiles_x = cdiv(d->VW, TW);
  k.tiles_per_image = k.tiles_x * cdiv(d->VH, k.TH);
  k.total_tiles = d->B * k.tiles_per_image;
  k.rows = k.TH < d->VH ? k.TH : d->VH;
  k.PH = (k.rows - 1) * d->in_stride + (dy_max - dy_min + 1);
  k.PW = (TW - 1) * d->in_stride + (dx_max - dx_min + 1);
  k.PWp = k.PW | 1;
  k.PLANE = k.PH * k.PWp;
  // Bank-conflict-free B-operand reads: lane j of an N tile reads column (channel j / ntaps, tap j % ntaps) at
  // channel * PLANE + tap_off.  With PWp = 3 (mod 32) the nine taps of a 3x3 stencil fall on nine consecutive banks, and with
  // PLANE = ntaps (mod 32) consecutive channels continue the run: 32 consecutive columns hit 32 different banks
  // (the unpadded 210-float plane put channels 0 and 2 on overlapping banks: 1.5 conflict cycles per read, round 1 PMC).
  if (d->ntaps == 9 && (k.PWp & 31) == 3 && dx_max - dx_min == 2 && dy_max - dy_min == 2)
    k.PLANE += ((9 - k.PLANE) % 32 + 32) % 32;
  for (int t = 0; t < d->ntaps; ++t) k.tap_off[t] = (d->tap_dy[t] - dy_min) * k.PWp + (d->tap_dx[t] - dx_min);
  k.CKW = d->ntaps == 1 ? 32 : (d->ntaps > 9 ? 4 : 16);
  k.ncol = k.CKW * d->ntaps;
  k.nt_used = cdiv(k.ncol, 32);
  k.nx = cdiv(k.CKW * k.PH * k.PW, NTHREADS);
  HDIFF_CHECK_ARG(k.nt_used <= MAXNT && k.nx <= NXS && k.PH < 1024 && k.PW < 1024,
                  "conv2d_wgrad: configuration does not fit (columns %d, slots %d)", k.ncol, k.nx);
  HDIFF_CHECK_ARG(nsplit <= k.total_tiles, "conv2d_wgrad: nsplit %d exceeds the %d tiles", nsplit, k.total_tiles);
  const size_t lds = (size_t)(BM * DYROW + k.CKW * k.PLANE + 2 * k.CKW) * sizeof(float);
  HDIFF_CHECK_ARG(lds <= 160 * 1024, "conv2d_wgrad: tile needs %zu bytes of LDS", lds);

  static uint64_t attr_mask = 0;
  if (first_use_on_device(attr_mask)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  dim3 grid(cdiv(d->Cout, BM), cdiv(d->CinPad, k.CKW), nsplit);
  (void)hipGetLastError();  // drop any stale error left by another HIP user in this thread
  if (k.tw_log2 == 5 && d->in_stride == 1)
    hipLaunchKernelGGL(conv_wgrad_kernel<1>, grid, dim3(NTHREADS), lds, (hipStream_t)stream, k);
  else if (k.tw_log2 == 5 && d->in_stride == 2)
    hipLaunchKernelGGL(conv_wgrad_kernel<2>, grid, dim3(NTHREADS), lds, (hipStream_t)stream, k);
  else
    hipLaunchKernelGGL(conv_wgrad_kernel<0>, grid, dim3(NTHREADS), lds, (hipStream_t)stream, k);
  HDIFF_CHECK_LAUNCH("conv_wgrad_kernel");
  return HDIFF_OK;
}

extern "C" int hdiff_conv_wgrad_unpack(const float* dwp, int nsplit, float* dw, int mode, int Cout, int Cin, int KH, int KW,
                                       int ntaps, const int* tap_ky, const int* tap_kx, int CinPad, int CoutPad,
                                       int accumulate, hdiff_stream_t stream) {
  HDIFF_CHECK_ARG(dwp && dw && tap_ky && tap_kx, "conv_wgrad_unpack: null pointer");
  HDIFF_CHECK_ARG(ntaps >= 1 && ntaps <= HDIFF_MAX_TAPS && nsplit >= 1, "conv_wgrad_unpack: bad sizes");
  UnpackK p{};
  p.mode = mode; p.Cout = Cout; p.Cin = Cin; p.KH = KH; p.KW = KW; p.ntaps = ntaps; p.CinPad = CinPad; p.CoutPad = CoutPad;
  p.nsplit = nsplit; p.accumulate = accumulate;
  for (int t = 0; t < ntaps; ++t) {
    p.ky[t] = tap_ky[t];
    p.kx[t] = tap_kx[t];
  }
  const size_t n = (size_t)ntaps * Cin * Cout;
  const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  (void)hipGetLastError();  // drop any stale error left by another HIP user in this thread
  hipLaunchKernelGGL(conv_wgrad_unpack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dwp, dw, p);
  HDIFF_CHECK_LAUNCH("conv_wgrad_unpack_kernel");
  return HDIFF_OK;
}
