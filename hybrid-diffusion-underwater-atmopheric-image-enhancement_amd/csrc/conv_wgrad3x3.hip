// Weight gradient of the 3x3 / stride-1 convolutions of the U-Net body (all 30 of them at the default config; 93 % of the
// conv-wgrad FLOPs of a 256x256 training step), fp32 MFMA 32x32x2, gfx950.  conv_wgrad.hip keeps every other geometry
// (strided 5x5, transposed-conv phases, 1x1, ragged widths, narrow head / tail convs) and dispatches here when
// hdiff::wgrad3x3_applicable() holds.
//
// Reference: autograd of nn.Conv2d(.., 3, stride=1, padding=1).weight (ModelCondition.py:171,186,82,217,250) in
// TrainCondition.py:60:   dW[co][ci][ky][kx] = sum_{b,y,x} dY[b][co][y][x] * act(X)[b][ci][y+ky-1][x+kx-1]
// with act = Swish(GroupNorm(.)) recomputed on the fly from the per-(sample, channel) scale / shift of the forward.
//
// GEMM roles: M = 128 output channels (4 waves x 32 rows), N = 32 input channels x 9 taps, K = pixels.
//   * one N tile per TAP: the 32 columns of a tile are 32 input channels (lane = channel, LDS plane stride odd: every
//     B read hits 32 different banks) and the tap is an IMMEDIATE offset into the staged activation patch -- no column
//     tables, no padding columns (the generic kernel spends 10 % of its MFMAs on them), no VALU in the MFMA loop;
//   * a wave keeps its 32 x 288 accumulator (144 VGPRs) for ALL pixels of every tile the workgroup walks: the waves split M,
//     not K, so there is no cross-wave reduction; the B operands are read by all four waves (LDS bandwidth is not the limit);
//   * tiles are 2 rows x 32 pixels; dY [64 px][128 co] and the activation patch [32 ci][4 rows][34] are DOUBLE-buffered in
//     LDS: the global loads of tile t+1 are issued before the 288 MFMAs of tile t, stored behind them, ONE barrier per tile;
//   * dY is stored pixel-major ([pixel][co], stride 129): the A operand of lane co at pixel p is conflict-free, and the
//     staging threads' stores (4 pixels of one channel each) are at most 2-way conflicted;
//   * every staging thread serves ONE input channel (17 patch elements): its GroupNorm scale / shift live in two registers.
// Workgroups: (Cout / 128) x (Cin / 32) x nsplit, each walking tiles split, split + nsplit, ...; partial slabs
// dwp[split][tap][ci][co] are summed in split order by hdiff_conv_wgrad_unpack -- no atomics, bitwise reproducible.
#include "common.h"

using namespace hdiff;

namespace {

constexpr int FT = 256;              // threads
constexpr int FM = 128;              // output channels per workgroup
constexpr int FC = 32;               // input channels per workgroup
constexpr int FP = 64;               // pixels per tile: 2 rows x 32
constexpr int YS = FM + 1;           // dY row stride (floats) in the pixel-major tile
constexpr int PR = 4, PC = 34;       // activation patch rows / columns per channel
constexpr int XP = PR * PC + 1;      // plane stride (137, odd)
constexpr int NXE = PR * PC / 8;     // patch elements per staging thread (17): 8 threads per channel
constexpr int NY4 = FM * FP / 4 / FT;  // dY float4 loads per thread (8)
constexpr int SY_FLOATS = FP * YS;   // 8256
constexpr int SX_FLOATS = FC * XP;   // 4384
constexpr int BUF_FLOATS = SY_FLOATS + SX_FLOATS;

struct WgradF {
  const float* x;            // the source tensor holding this launch's channels is chosen per workgroup (x0 or x1)
  const float* x1;
  int C0, C1, Cin, H, W;
  const float* gn_scale;
  const float* gn_shift;
  const float* dy;
  int Cout, tiles_x, tiles_per_image, total_tiles, nsplit;
  int CinPad, CoutPad;
  float* dwp;
};

__device__ __forceinline__ float swish_w3(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

template <bool GN>
__global__ __launch_bounds__(FT) void conv_wgrad3x3_kernel(const WgradF p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int co0 = blockIdx.x * FM;
  const int ci0 = blockIdx.y * FC;
  const int split = blockIdx.z;
  const size_t HW = (size_t)p.H * p.W;

  // ---- staging roles -------------------------------------------------------------------------------------------------
  // activation patch: thread serves channel xci = tid / 8, elements m = (tid & 7) + 8 i  ->  (row m / 34, column m % 34)
  const int xci = tid >> 3;
  const int cg = ci0 + xci;                                   // global input channel
  const float* xsrc = (cg < p.C0) ? p.x + (size_t)cg * HW : p.x1 + (size_t)(cg - p.C0) * HW;
  const size_t xbstride = (size_t)((cg < p.C0) ? p.C0 : p.C1) * HW;
  int xoff[NXE];            // offset of the element from the patch origin (row -1, column -1 of the tile) in the image plane
  int xlds[NXE];
  unsigned m_top = 0, m_bot = 0, m_left = 0, m_right = 0;     // bit i: element i lies in patch row 0 / row 3 / column 0 / column 33
#pragma unroll
  for (int i = 0; i < NXE; ++i) {
    const int m = (tid & 7) + 8 * i;
    const int r = m / PC, c = m - r * PC;
    xoff[i] = r * p.W + c;
    xlds[i] = xci * XP + r * PC + c;
    m_top |= (r == 0 ? 1u : 0u) << i;
    m_bot |= (r == PR - 1 ? 1u : 0u) << i;
    m_left |= (c == 0 ? 1u : 0u) << i;
    m_right |= (c == PC - 1 ? 1u : 0u) << i;
  }
  const int xsafe = p.W + 1;                                  // patch element (1, 1): inside the image for every tile
  // dY: float4 number idx = tid + 256 i  ->  channel idx / 16, row (idx % 16) / 8, columns 4 (idx % 8) ..+3
  // (all eight share the thread's (row, column) and step the channel by 16)
  const int yrow = (tid & 15) >> 3, yc4 = tid & 7, yco = tid >> 4;
  const size_t OHW = HW;                                       // stride 1, same-size output

  // ---- accumulators: acc[tap] is the 32 x 32 tile (rows = this wave's output channels, columns = input channels)
  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  float xr[NXE];
  float4 yr[NY4];
  float gsc = 1.f, gsh = 0.f;
  unsigned xbad = 0;        // bit i: element i of the tile in flight lies outside the image (zero padding)

  auto tile_origin = [&](int t, int& b, int& y0, int& x0) {
    b = t / p.tiles_per_image;
    const int ti = t - b * p.tiles_per_image;
    const int ty = ti / p.tiles_x, tx = ti - ty * p.tiles_x;
    y0 = ty * 2;
    x0 = tx * 32;
  };
  auto issue_loads = [&](int t) {
    int b, y0, x0;
    tile_origin(t, b, y0, x0);
    const bool top = (y0 == 0), bottom = (y0 + 2 == p.H), left = (x0 == 0), right = (x0 + 32 == p.W);
    const float* xs = xsrc + (size_t)b * xbstride + (ptrdiff_t)((y0 - 1) * p.W + (x0 - 1));
    xbad = (top ? m_top : 0u) | (bottom ? m_bot : 0u) | (left ? m_left : 0u) | (right ? m_right : 0u);
#pragma unroll
    for (int i = 0; i < NXE; ++i) xr[i] = xs[((xbad >> i) & 1u) ? xsafe : xoff[i]];     // branch-free; zeroed when stored
    if (GN) {
      gsc = p.gn_scale[(size_t)b * p.Cin + cg];
      gsh = p.gn_shift[(size_t)b * p.Cin + cg];
    }
    const float* ys = p.dy + ((size_t)b * p.Cout + co0 + yco) * OHW + (size_t)(y0 + yrow) * p.W + x0 + yc4 * 4;
#pragma unroll
    for (int i = 0; i < NY4; ++i) yr[i] = *reinterpret_cast<const float4*>(ys + (size_t)(16 * i) * OHW);
  };
  auto store_staged = [&](float* buf) {
    float* sY = buf;
    float* sX = buf + SY_FLOATS;
#pragma unroll
    for (int i = 0; i < NXE; ++i) {
      float v = xr[i];
      if (GN) v = swish_w3(fmaf(v, gsc, gsh));
      sX[xlds[i]] = ((xbad >> i) & 1u) ? 0.f : v;             // the conv pads the ACTIVATED tensor with zeros
    }
    const int pix = yrow * 32 + yc4 * 4;
#pragma unroll
    for (int i = 0; i < NY4; ++i) {
      float* d = sY + pix * YS + yco + 16 * i;
      d[0] = yr[i].x;
      d[YS] = yr[i].y;
      d[2 * YS] = yr[i].z;
      d[3 * YS] = yr[i].w;
    }
  };

  int t = split;
  if (t < p.total_tiles) {
    issue_loads(t);
    store_staged(smem);
  }
  __syncthreads();
  int cur = 0;
  for (; t < p.total_tiles; t += p.nsplit) {
    const bool more = (t + p.nsplit < p.total_tiles);
    if (more) issue_loads(t + p.nsplit);

    const float* sY = smem + cur * BUF_FLOATS;
    const float* sX = sY + SY_FLOATS;
    const float* ya = sY + h * YS + wave * 32 + l31;          // A: dY[pixel 2 kk + h][co]
    const float* xb = sX + l31 * XP + h;                       // B: patch[ci][row + ky][column 2 kk' + h + kx]
    // one base register per tap column, opaque to the compiler: each read stays a ds_read_b32 with an immediate offset
    // (related bases get fused into ds_read2_b32 whose short offsets cost a v_add per pair -- VALU time the MFMAs cannot hide)
    int bo[3] = {0, 1, 2};
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) asm volatile("" : "+v"(bo[kx]));
#pragma unroll
    for (int kk = 0; kk < FP / 2; ++kk) {
      const int row = kk >> 4, col = 2 * (kk & 15);
      const float a = ya[2 * kk * YS];
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int ky = tap / 3, kx = tap - 3 * ky;
        const float bv = xb[bo[kx] + (row + ky) * PC + col];
        acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[tap], 0, 0, 0);
      }
    }
    if (more) store_staged(smem + (cur ^ 1) * BUF_FLOATS);
    __syncthreads();
    cur ^= 1;
  }

  // ---- this split's packed partial slab: dwp[split][tap][ci][co], four consecutive output channels per store
  float* slab = p.dwp + (size_t)split * 9 * p.CinPad * p.CoutPad;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    float* dst = slab + ((size_t)tap * p.CinPad + ci0 + l31) * p.CoutPad + co0 + wave * 32 + 4 * h;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *reinterpret_cast<float4*>(dst + 8 * q) =
          make_float4(acc[tap][4 * q + 0], acc[tap][4 * q + 1], acc[tap][4 * q + 2], acc[tap][4 * q + 3]);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// 1x1 / stride-1 weight gradients (the attention in / out projections and the ResBlock shortcuts: a quarter of the conv FLOPs
// at the attention levels; the generic kernel ran them at 6 VALU instructions per MFMA, 40 % MFMA busy):
//     dW[co][ci] = sum_{b, p} dY[b][co][p] * X[b][ci][p]                      -- a plain GEMM with K = pixels
// Same schedule as the 3x3 kernel above with channel GROUPS in place of taps: M = 128 output channels (4 waves x 32 rows),
// N = 128 input channels = four 32-channel N tiles, K = 64 consecutive pixels of the flattened plane per tile.  dY and X tiles
// have the same shape and are staged by the same code (16-byte loads, pixel-major [64 px][129] in LDS: A and B operands of lane
// l31 at pixel p are conflict-free), double-buffered, one barrier per tile.  No GroupNorm prologue exists on these convs.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int P1 = 64;                 // pixels per tile
constexpr int T1_FLOATS = P1 * YS;     // one staged tile: [64 px][129]
constexpr int N1 = 8;                  // float4 loads per thread per tile and tensor (128 rows x 16 segments / 256)

struct Wgrad1 {
  const float* x0;
  const float* x1;
  int C0, C1, Cin;
  const float* dy;
  int Cout, HW, tiles_per_image, total_tiles, nsplit, CinPad, CoutPad;
  float* dwp;
};

__global__ __launch_bounds__(FT) void conv_wgrad1x1_kernel(const Wgrad1 p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int co0 = blockIdx.x * FM;
  const int ci0 = blockIdx.y * 128;
  const int split = blockIdx.z;
  const size_t HW = (size_t)p.HW;

  // staging: float4 number idx = tid + 256 i -> row (channel) idx / 16 = tid / 16 + 16 i, pixels 4 (tid % 16) ..+3
  const int srow = tid >> 4, sc4 = tid & 15;
  const float* xrow[N1];       // this thread's eight input-channel rows (sample 0), nullptr-free: missing channels read row 0
  size_t xbs[N1];              // their batch strides (x0 and x1 have different channel counts)
  unsigned xdead = 0;          // bit i: channel ci0 + row does not exist (Cin not a multiple of 128): staged as zero
#pragma unroll
  for (int i = 0; i < N1; ++i) {
    const int c = ci0 + srow + 16 * i;
    const bool live = c < p.Cin;
    const int cc = live ? c : 0;
    xrow[i] = (cc < p.C0) ? p.x0 + (size_t)cc * HW : p.x1 + (size_t)(cc - p.C0) * HW;
    xbs[i] = (size_t)((cc < p.C0) ? p.C0 : p.C1) * HW;
    xdead |= (live ? 0u : 1u) << i;
  }
  const float* yrow0 = p.dy + (size_t)(co0 + srow) * HW;

  f32x16 acc[4];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;

  float4 xr[N1], yr[N1];
  auto issue_loads = [&](int t) {
    const int b = t / p.tiles_per_image;
    const size_t p0 = (size_t)(t - b * p.tiles_per_image) * P1 + sc4 * 4;
#pragma unroll
    for (int i = 0; i < N1; ++i) {
      xr[i] = *reinterpret_cast<const float4*>(xrow[i] + (size_t)b * xbs[i] + p0);
      yr[i] = *reinterpret_cast<const float4*>(yrow0 + ((size_t)b * p.Cout + 16 * i) * HW + p0);
    }
  };
  auto store_staged = [&](float* buf) {
    float* sY = buf;
    float* sX = buf + T1_FLOATS;
    const int base = (sc4 * 4) * YS + srow;
#pragma unroll
    for (int i = 0; i < N1; ++i) {
      float* dy_ = sY + base + 16 * i;
      dy_[0] = yr[i].x; dy_[YS] = yr[i].y; dy_[2 * YS] = yr[i].z; dy_[3 * YS] = yr[i].w;
      const bool dead = (xdead >> i) & 1u;
      float* dx_ = sX + base + 16 * i;
      dx_[0] = dead ? 0.f : xr[i].x; dx_[YS] = dead ? 0.f : xr[i].y; dx_[2 * YS] = dead ? 0.f : xr[i].z; dx_[3 * YS] = dead ? 0.f : xr[i].w;
    }
  };

  int t = split;
  if (t < p.total_tiles) {
    issue_loads(t);
    store_staged(smem);
  }
  __syncthreads();
  int cur = 0;
  for (; t < p.total_tiles; t += p.nsplit) {
    const bool more = (t + p.nsplit < p.total_tiles);
    if (more) issue_loads(t + p.nsplit);
    const float* sY = smem + cur * 2 * T1_FLOATS;
    const float* ya = sY + h * YS + wave * 32 + l31;                  // A: dY[pixel 2 kk + h][co]
    const float* xb = sY + T1_FLOATS + h * YS + l31;                  // B: X[pixel 2 kk + h][ci group g]
    int go[4] = {0, 32, 64, 96};
#pragma unroll
    for (int g = 0; g < 4; ++g) asm volatile("" : "+v"(go[g]));       // opaque bases: one ds_read_b32 + immediate per operand
#pragma unroll
    for (int kk = 0; kk < P1 / 2; ++kk) {
      const float a = ya[2 * kk * YS];
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xb[go[g] + 2 * kk * YS], acc[g], 0, 0, 0);
    }
    if (more) store_staged(smem + (cur ^ 1) * 2 * T1_FLOATS);
    __syncthreads();
    cur ^= 1;
  }

  // ---- this split's packed partial slab dwp[split][0][ci][co]
  float* slab = p.dwp + (size_t)split * p.CinPad * p.CoutPad;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int ci = ci0 + g * 32 + l31;
    if (ci < p.Cin) {
      float* dst = slab + (size_t)ci * p.CoutPad + co0 + wave * 32 + 4 * h;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4*>(dst + 8 * q) = make_float4(acc[g][4 * q + 0], acc[g][4 * q + 1], acc[g][4 * q + 2], acc[g][4 * q + 3]);
    }
  }
}

}  // namespace

namespace hdiff {

bool wgrad1x1_applicable(const hdiff_conv_wgrad_desc* d) {
  if (d->ntaps != 1 || d->tap_dy[0] != 0 || d->tap_dx[0] != 0 || d->in_stride != 1 || d->gn_scale != nullptr) return false;
  if (d->out_sy != 1 || d->out_oy != 0 || d->out_sx != 1 || d->out_ox != 0) return false;
  if (d->VH != d->H || d->VW != d->W || d->OH != d->H || d->OW != d->W) return false;
  const long HW = (long)d->H * d->W;
  const int Cin = d->C0 + d->C1;
  if (HW % P1 != 0 || HW >= (1L << 30)) return false;
  if (d->Cout % FM != 0 || Cin % 32 != 0 || d->CinPad != Cin || d->CoutPad != d->Cout) return false;
  return true;
}

int wgrad1x1_nsplit(const hdiff_conv_wgrad_desc* d) {
  const int base = (d->Cout / FM) * cdiv(d->C0 + d->C1, 128);
  const long total = (long)d->B * ((long)d->H * d->W / P1);
  long ns = 512 / base;                     // one workgroup per CU (132 KB of LDS): at most two full rounds of the 256 CUs
  if (ns > total) ns = total;
  return ns < 1 ? 1 : (int)ns;
}

int launch_wgrad1x1(const hdiff_conv_wgrad_desc* d, float* dwp, int nsplit, hipStream_t stream) {
  Wgrad1 k{};
  k.x0 = d->x0; k.x1 = d->x1; k.C0 = d->C0; k.C1 = d->C1; k.Cin = d->C0 + d->C1; k.dy = d->dy; k.Cout = d->Cout;
  k.HW = d->H * d->W;
  k.tiles_per_image = k.HW / P1;
  k.total_tiles = d->B * k.tiles_per_image;
  k.nsplit = nsplit; k.CinPad = d->CinPad; k.CoutPad = d->CoutPad; k.dwp = dwp;
  HDIFF_CHECK_ARG(nsplit >= 1 && nsplit <= k.total_tiles, "conv2d_wgrad (1x1): nsplit %d not in [1, %d]", nsplit, k.total_tiles);
  const size_t lds = (size_t)4 * T1_FLOATS * sizeof(float);
  static uint64_t attr_mask = 0;
  if (first_use_on_device(attr_mask))
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad1x1_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  dim3 grid(d->Cout / FM, cdiv(k.Cin, 128), nsplit);
  (void)hipGetLastError();  // drop any stale error left by another HIP user in this thread
  hipLaunchKernelGGL(conv_wgrad1x1_kernel, grid, dim3(FT), lds, stream, k);
  HDIFF_CHECK_LAUNCH("conv_wgrad1x1_kernel");
  return HDIFF_OK;
}

bool wgrad3x3_applicable(const hdiff_conv_wgrad_desc* d) {
  if (d->ntaps != 9 || d->in_stride != 1) return false;
  for (int t = 0; t < 9; ++t)
    if (d->tap_dy[t] != t / 3 - 1 || d->tap_dx[t] != t % 3 - 1) return false;
  if (d->out_sy != 1 || d->out_oy != 0 || d->out_sx != 1 || d->out_ox != 0) return false;
  if (d->VH != d->H || d->VW != d->W || d->OH != d->H || d->OW != d->W) return false;
  if (d->W % 32 != 0 || d->H % 2 != 0) return false;
  if (d->Cout % FM != 0 || d->C0 % FC != 0 || d->C1 % FC != 0) return false;
  if (d->CinPad != d->C0 + d->C1 || d->CoutPad != d->Cout) return false;
  return true;
}

int wgrad3x3_nsplit(const hdiff_conv_wgrad_desc* d) {
  const int base = (d->Cout / FM) * ((d->C0 + d->C1) / FC);
  const int total = d->B * (d->H / 2) * (d->W / 32);
  int ns = 512 / base;                      // one workgroup per CU (100 KB of LDS): at most two FULL rounds of the 256 CUs
  if (ns > total) ns = total;               // (rounding up instead left a third, nearly empty round: 384 -> 128 ran at 76 TFLOP/s)
  return ns < 1 ? 1 : ns;
}

int launch_wgrad3x3(const hdiff_conv_wgrad_desc* d, float* dwp, int nsplit, hipStream_t stream) {
  WgradF k{};
  k.x = d->x0; k.x1 = d->x1; k.C0 = d->C0; k.C1 = d->C1; k.Cin = d->C0 + d->C1; k.H = d->H; k.W = d->W;
  k.gn_scale = d->gn_scale; k.gn_shift = d->gn_shift; k.dy = d->dy; k.Cout = d->Cout;
  k.tiles_x = d->W / 32;
  k.tiles_per_image = k.tiles_x * (d->H / 2);
  k.total_tiles = d->B * k.tiles_per_image;
  k.nsplit = nsplit; k.CinPad = d->CinPad; k.CoutPad = d->CoutPad; k.dwp = dwp;
  HDIFF_CHECK_ARG(nsplit >= 1 && nsplit <= k.total_tiles, "conv2d_wgrad (3x3): nsplit %d not in [1, %d]", nsplit, k.total_tiles);
  const size_t lds = (size_t)2 * BUF_FLOATS * sizeof(float);
  static uint64_t attr_mask = 0;
  if (first_use_on_device(attr_mask)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad3x3_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad3x3_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  dim3 grid(d->Cout / FM, k.Cin / FC, nsplit);
  (void)hipGetLastError();  // drop any stale error left by another HIP user in this thread
  if (d->gn_scale)
    hipLaunchKernelGGL(conv_wgrad3x3_kernel<true>, grid, dim3(FT), lds, stream, k);
  else
    hipLaunchKernelGGL(conv_wgrad3x3_kernel<false>, grid, dim3(FT), lds, stream, k);
  HDIFF_CHECK_LAUNCH("conv_wgrad3x3_kernel");
  return HDIFF_OK;
}

}  // namespace hdiff
