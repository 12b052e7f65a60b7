// 1x1 stride-1 convolution as a plain GEMM on the fp32-input MFMA, with NO LDS staging and no barriers.
//
// Call sites in the reference: the packed in-/out-projections of nn.MultiheadAttention viewed as 1x1 convs
// (ModelCondition.py:189, diffusion/Model.py:291) and the ResBlock shortcut (ModelCondition.py:192, Model.py:294).
//
//     D[co][p] = sum_ci Wp[ci][co] * X[ci][p]            p = flat pixel index (NCHW: contiguous per channel)
// With v_mfma_f32_32x32x2_f32 lane (l31, h) supplies A[m = l31][k = h] and B[k = h][n = l31]; for a 1x1 conv both are
// 32 consecutive floats of a row of Wp ([ci][co]) resp. of X ([ci][pixel]) -- already coalesced 128-byte segments in
// global memory, so the operands go global -> register -> MFMA.  The implicit-GEMM kernel's LDS round trip pays off when
// each staged element feeds 9-25 taps; for one tap it is pure overhead (measured 36 TFLOP/s, barrier bound).
// Operand loads for the next group of k-steps are issued before the MFMAs of the current one.  The channel-block index is the fastest grid dimension so that the blocks sharing a pixel
// range run together and X is fetched from HBM once.
#include <stdlib.h>

#include "common.h"

using namespace hdiff;

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
}  // namespace hdiff

namespace {

constexpr int THREADS = 256;
constexpr int EC = 8;     // channel rows per epilogue group
constexpr int U = 2;      // k-steps per software-pipeline stage (4 measured no faster)
typedef float f32x2 __attribute__((ext_vector_type(2)));

// A wave owns 64 output channels x 128 pixels as 2 x 4 accumulators of 32x32.  The tile <-> index maps are chosen so that
// one vector load feeds all tiles of an operand: lane l31 holds pixels 4*l31 .. 4*l31+3 (one 16-byte load of X per k: the
// HBM stream moves in 16-byte pieces, which is what keeps enough bytes in flight -- with 4-byte loads the kernel sat at
// 0.8 TB/s, latency bound) and channels 2*l31, 2*l31+1 (one 8-byte load of the L2-resident weights).  So N-tile nt is the
// pixel set {4*l + nt} and M-tile mt the channel set {2*m + mt}; the epilogue stores 16 bytes per lane.
// Preconditions (checked by the dispatcher in conv_igemm.hip): HW % 128 == 0, C0 % 2 == 0 (a k-pair never straddles the
// concat seam: the activation base pointer is wave-uniform), CoutPad % 64 == 0, weight rows [Cin, CinPad) zero
// (hdiff_pack_conv_weight).  Every load is unconditional: an odd Cin's phantom channel is clamped onto the last real one
// and meets a zero weight row.
__global__ __launch_bounds__(THREADS, 2) void conv1x1_direct_kernel(const Conv1x1K p) {
  constexpr int MT = 2, WN = 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int co0 = blockIdx.x * 64;
  const long px0 = ((long)blockIdx.y * 4 + __builtin_amdgcn_readfirstlane(wave)) * 128;      // wave-uniform, in SGPRs
  const int b = blockIdx.z;
  if (px0 >= p.HW) return;                      // whole wave out of range (no barriers in this kernel)
  const int C1 = p.Cin - p.C0;
  const unsigned hw = (unsigned)p.HW;
  const float* x0b = p.x0 + (size_t)b * p.C0 * p.HW + px0;
  const float* x1b = p.x1 ? p.x1 + (size_t)b * C1 * p.HW + px0 - (size_t)p.C0 * p.HW : x0b;
  const float* wb = p.wp + co0 + 2 * l31 + (size_t)h * p.CoutPad;
  const unsigned wstep = 2u * (unsigned)p.CoutPad;

  f32x16 acc[MT][WN];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < WN; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  struct Frag { f32x2 a; f32x4 b; };
  const int K2 = (p.Cin + 1) / 2;
  auto load = [&](Frag& f, int k2) {
    k2 = min(k2, K2 - 1);                                    // uniform; the pipeline's overshoot re-reads the last pair
    const float* xb = (2 * k2 < p.C0) ? x0b : x1b;             // uniform select
    const unsigned kx = (unsigned)min(2 * k2 + h, p.Cin - 1);
    f.b = *reinterpret_cast<const f32x4*>(xb + (size_t)(kx * hw + 4u * l31));
    f.a = *reinterpret_cast<const f32x2*>(wb + (size_t)((unsigned)k2 * wstep));
  };
  auto mma = [&](const Frag& f) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < WN; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[mt], f.b[nt], acc[mt][nt], 0, 0, 0);
  };

  Frag fa[U], fb[U];
#pragma unroll
  for (int u = 0; u < U; ++u) load(fa[u], u);
  for (int k2 = 0; k2 < K2; k2 += 2 * U) {
#pragma unroll
    for (int u = 0; u < U; ++u) load(fb[u], k2 + U + u);
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (k2 + u < K2) mma(fa[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) load(fa[u], k2 + 2 * U + u);
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (k2 + U + u < K2) mma(fb[u]);
  }

  // ---- epilogue.  A full 64-channel block (a wave-uniform test: everything but a tail block) takes the branch-free form:
  // the bias / vector / residual loads of EC channel rows are in flight together before the first add -- with the
  // per-row channel test every row waited for its own three loads in turn (32 dependent round trips per wave).
  if (co0 + 64 <= p.Cout) {
    // addresses = wave-uniform base (scalar registers) + one 32-bit lane offset: channel row 8h of the block, pixel 4*l31
    const size_t blk = ((size_t)b * p.Cout + co0) * p.HW + px0;
    const unsigned lane_row = 8u * h, lane_off = lane_row * hw + 4u * l31;
    const float* biasu = p.bias ? p.bias + co0 : nullptr;
    const float* vecu = p.addvec ? p.addvec + (size_t)b * p.Cout + co0 : nullptr;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r0 = 0; r0 < 16; r0 += EC) {
        f32x4 res[EC];
        float add[EC];
#pragma unroll
        for (int j = 0; j < EC; ++j) {
          const int r = r0 + j;
          const int rowc = 2 * (r & 3) + 16 * (r >> 2) + mt;             // compile-time part of the channel row
          float a = 0.f;
          if (biasu) a += biasu[rowc + lane_row];
          if (vecu) a += vecu[rowc + lane_row];
          add[j] = a;
          if (p.residual) res[j] = *reinterpret_cast<const f32x4*>(p.residual + blk + (size_t)rowc * p.HW + lane_off);
        }
#pragma unroll
        for (int j = 0; j < EC; ++j) {
          const int r = r0 + j;
          const int rowc = 2 * (r & 3) + 16 * (r >> 2) + mt;
          f32x4 v = {acc[mt][0][r], acc[mt][1][r], acc[mt][2][r], acc[mt][3][r]};
          v += add[j];
          if (p.residual) v += res[j];
          *reinterpret_cast<f32x4*>(p.out + blk + (size_t)rowc * p.HW + lane_off) = v;
        }
        __builtin_amdgcn_sched_barrier(0);     // keep the next group's loads below: the register budget is three waves per SIMD
      }
    return;
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * h) + mt;
      if (co < p.Cout) {
        f32x4 v = {acc[mt][0][r], acc[mt][1][r], acc[mt][2][r], acc[mt][3][r]};
        float add = 0.f;
        if (p.bias) add += p.bias[co];
        if (p.addvec) add += p.addvec[b * p.Cout + co];
        v += add;
        const size_t o = ((size_t)b * p.Cout + co) * p.HW + px0 + 4 * l31;
        if (p.residual) v += *reinterpret_cast<const f32x4*>(p.residual + o);
        *reinterpret_cast<f32x4*>(p.out + o) = v;
      }
    }
}

}  // namespace

namespace hdiff {

void launch_conv1x1_direct(const Conv1x1K& k, int B, hipStream_t stream) {
  dim3 grid(cdiv(k.Cout, 64), (unsigned)((k.HW + 511) / 512), B);
  hipLaunchKernelGGL(conv1x1_direct_kernel, grid, dim3(THREADS), 0, stream, k);
}

}  // namespace hdiff
