// Shared helpers for the gfx950 kernels of the CFG-DDPM hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/hdiff.h"

namespace hdiff {

void set_error(const char* fmt, ...);

#define HDIFF_CHECK_ARG(cond, ...)            \
  do {                                        \
    if (!(cond)) {                            \
      hdiff::set_error(__VA_ARGS__);          \
      return HDIFF_ERR_INVALID;               \
    }                                         \
  } while (0)

#define HDIFF_CHECK_LAUNCH(what)                                                   \
  do {                                                                             \
    hipError_t e__ = hipGetLastError();                                            \
    if (e__ != hipSuccess) {                                                       \
      hdiff::set_error("%s: %s", what, hipGetErrorString(e__));                    \
      return HDIFF_ERR_LAUNCH;                                                     \
    }                                                                              \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float swishf(float v) { return v / (1.0f + __expf(-v)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// hipFuncSetAttribute is per DEVICE: true the first time the calling site (one `static uint64_t` mask each) runs with the
// current device, so that a process using several GPUs raises the dynamic-LDS limit on each of them.
static inline bool first_use_on_device(uint64_t& mask) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
  const uint64_t bit = 1ull << dev;
  if (mask & bit) return false;
  mask |= bit;
  return true;
}

// XCD-aware tile order for grids of (blocks along the sequence, heads, batch).  Workgroups are handed to the 8 XCDs round
// robin by linear id, and each XCD has its own L2: with the plain order the 8 XCDs all stream the K/V (or Q/dO) of every
// head (measured on the forward: 4.5x the algorithmic HBM bytes).  This bijection gives each (head, sample) pair to ONE XCD
// -- pair p runs on XCD p % 8 -- so its operands are fetched into one L2 only.  Needs heads * batch % 8 == 0 (heads = 8 in
// this model); otherwise the identity.
struct TileId { int x, head, b; };
__device__ __forceinline__ TileId xcd_tile() {
  const unsigned gx = gridDim.x, pairs = gridDim.y * gridDim.z;
  if (pairs % 8u != 0u) return TileId{(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z};
  const unsigned lin = blockIdx.x + gx * (blockIdx.y + gridDim.y * blockIdx.z);
  const unsigned xcd = lin & 7u, idx = lin >> 3;
  const unsigned pair = (idx / gx) * 8u + xcd, x = idx % gx;
  return TileId{(int)x, (int)(pair % gridDim.y), (int)(pair / gridDim.y)};
}

// conv_wgrad3x3.hip: the 3x3 / stride-1 weight-gradient kernel of the U-Net body (dispatched from conv_wgrad.hip)
bool wgrad3x3_applicable(const hdiff_conv_wgrad_desc* d);
int wgrad3x3_nsplit(const hdiff_conv_wgrad_desc* d);
int launch_wgrad3x3(const hdiff_conv_wgrad_desc* d, float* dwp, int nsplit, hipStream_t stream);
bool wgrad1x1_applicable(const hdiff_conv_wgrad_desc* d);     // same file: the 1x1 / stride-1 convs without a prologue
int wgrad1x1_nsplit(const hdiff_conv_wgrad_desc* d);
int launch_wgrad1x1(const hdiff_conv_wgrad_desc* d, float* dwp, int nsplit, hipStream_t stream);

// conv3x3_x3.hip: split-bf16 convolution over the 3x3 neighbourhood (plain 3x3 / stride-1 convs and the four output-parity
// phases of the transposed 5x5 / stride-2 conv: every tap offset lies in [-1, 1]^2)
struct ConvX3K {
  const float* x0;
  const float* x1;
  int C0, C1, Cin, H, W;
  const unsigned* wp3;             // [Cin/16][ntaps][3][CoutPad][8] packed bf16 pairs
  int CoutPad, Cout;
  const float* bias;
  const float* gn_scale;
  const float* gn_shift;
  const float* addvec;
  const float* residual;
  float* out;
  int tiles_x;
  int ntaps;                       // 9, 6 or 4
  int tap_off[9];                  // LDS word offset of the tap inside the staged patch: ((dy + 1) * 34 + (dx + 1)) * 4
  int OH, OW, out_sy, out_oy, out_sx, out_ox;     // output pixel (vy * out_sy + out_oy, vx * out_sx + out_ox) of an OH x OW plane
  // fp16-pair form (plain 3x3 behind GroupNorm + Swish): wp3 then holds [Cin/16][9][2][CoutPad][8] words of w 2^t
  const float* act_scale;          // device {2^s, 2^-s} of the staged activations (hdiff_gn_act_scale), NULL = bf16 triples
  const float* w_scale;            // the pack's tail {bits of max |w|, 2^-t, 2^t, 0} (hdiff_pack_conv_weight_h2)
  float one;                       // 1.0f, opaque to the compiler
};
void launch_conv3x3_x3(const ConvX3K& k, int B, hipStream_t stream);

// conv1x1_x3.hip: the 1x1 / stride-1 convolution on bf16 triples (no LDS; operands split in registers)
struct Conv1x1X3K {
  const float* x0;
  const float* x1;
  int C0, Cin;
  long HW;
  const unsigned* wp3;             // [Cin/16][1][3][CoutPad][8] packed bf16 pairs (hdiff_pack_conv_weight_x3_taps, one tap)
  int CoutPad, Cout;
  const float* bias;
  const float* addvec;
  const float* residual;
  float* out;
};
void launch_conv1x1_x3(const Conv1x1X3K& k, int B, hipStream_t stream);

int contraction_mode();   // HDIFF_CONTRACT_*

// Mutation switch of the parity suite's own sensitivity tests (tests/test_gpu_mutation.py): a library built with
// -DHDIFF_MUTANT=<mask> silently damages ONE low-order piece product per bit, at the 2^-16 / 2^-17 level of the product --
//   bit 0 (1)   the split-bf16 3x3 and 1x1 convolutions: the term w0 x2 dropped (the 3x3's fp16-pair form: the low five bits of
//               every activation's second piece masked)
//   bit 1 (2)   the d_head 32 attention forward (attention_x3p.hip): the low five bits of the second Q piece of the scores masked
//   bit 2 (4)   the d_head 16 attention forward (attention_h2.hip): the same in its score product
//   bit 3 (8)   the d_head 16 attention forward: the low five bits of every second piece of P masked (the P V product)
//   bit 4 (16)  the attention backward (attention_bwd_h2.hip): the cross product o0 v1 of dP = dO V^T dropped
//   bit 5 (32)  the attention backward: the product o1 p0 of dV^T = dO^T P dropped
//   bit 6 (64)  the attention backward: the low five bits of the second fp16 piece of dS masked (2^-17 of dS: dK^T and dQ^T)
// `make mutant` builds bits 0, 1, 2, 4, 5 into build/libhdiff_mutant.so, `make mutant2` bits 3 and 6 into build/libhdiff_mutant2.so (bits 2
// and 3 both end in the d_head 16 forward's output, bits 4 and 6 both in dK / dQ: one library could not tell which of them a red test
// has seen).  The float64 error-class tests must FAIL on them.
#ifndef HDIFF_MUTANT
#define HDIFF_MUTANT 0
#endif

// attention_bwd_h2.hip: the attention backward at d_head 16 / 32 in the split-operand mode (dispatched from attention_bwd.hip)
long long mha_bwd_slab_cap_bytes();                          // upper bound on the dQ partial slabs (HDIFF_BWD_SLAB_GIB)
bool mha_bwd_x3_shape_ok(int B, int C, int heads, int L);     // the shape alone (mode-independent)
bool mha_bwd_x3_applicable(int B, int C, int heads, int L);   // shape AND the bf16x3 contraction mode
int64_t mha_bwd_x3_workspace_floats(int B, int C, int heads, int L);   // dQ slabs + piece tensors + maxima
bool launch_mha_bwd_h2(const float* qkv, const float* d_o, const float* lse2, const float* delta, float* dqkv, float* ws, int B,
                       int C, int heads, int L, hipStream_t stream);
// Attention forward in the split-operand mode.  attention_x3p.hip: d_head 32 on operands split ONCE into a workspace (fp16 pairs;
// 0 bytes = shape not covered); attention_h2.hip: d_head 16 likewise, and the split passes of both; attention_x3.hip: the kernel that
// splits in its loop (bf16 triples), for calls without a workspace.  Each returns false when the shape is not covered / no workspace.
int64_t mha_fwd_x3p_workspace(int B, int C, int heads, int L);
int64_t mha_fwd_h2_tail_bytes(int B, int C);      // bytes behind the pairs of the workspace (Q / K row maxima)
bool launch_mha_fwd_x3p(const float* qkv, float* o, float* lse2, int B, int C, int heads, int L, float qscale, void* ws,
                        int64_t ws_bytes, hipStream_t stream);
bool launch_mha_fwd_h2(const float* qkv, float* o, float* lse2, int B, int C, int heads, int L, float qscale, void* ws,
                       int64_t ws_bytes, hipStream_t stream);
void launch_qk_split_h2(const float* qkv, void* ws, int B, int C, int heads, int L, float qscale, hipStream_t stream);   // Q, K as fp16 score operands
void launch_v_split_h2(const float* qkv, void* ws, int B, int C, int heads, int L, hipStream_t stream);                  // V as fp16 pairs, d_head 16 / 32
bool launch_mha_fwd_x3(const float* qkv, float* o, float* lse2, int B, int C, int heads, int L, float qscale, hipStream_t stream);

}  // namespace hdiff
