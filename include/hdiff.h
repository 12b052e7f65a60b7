/*
 * hdiff.h -- C ABI of the MI355X (gfx950) CFG-DDPM hot path.
 *
 * One shared library (libhdiff.so, built from hybrid-diffusion-underwater-atmopheric-image-enhancement_amd/csrc)
 * exports every operator the reference's hot path calls through PyTorch.  The reference
 * has no FFI of its own for this path: its "operator interface" is the set of ATen call sites listed beside each entry
 * point below (paths relative to the reference checkout).  All entry points
 *   - take plain device pointers, sizes and a HIP stream (void* = hipStream_t); no torch types,
 *   - are stream-ordered and stateless (no allocation, no synchronisation: safe under hipGraph capture),
 *   - return 0 on success or a negative hdiff_status; hdiff_last_error() gives the text,
 *   - compute in fp32 (fp32-input MFMA for every contraction); tensors are fp32 NCHW contiguous, indices int64.
 */
#ifndef HDIFF_H_
#define HDIFF_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* hdiff_stream_t; /* hipStream_t */

enum hdiff_status {
  HDIFF_OK = 0,
  HDIFF_ERR_INVALID = -1,  /* bad shape / unsupported configuration */
  HDIFF_ERR_LAUNCH = -2,   /* HIP launch error */
  HDIFF_ERR_NO_DEVICE = -3
};

int hdiff_abi_version(void);
const char* hdiff_last_error(void);
/* Number of HIP devices visible (0 without a GPU); never initialises a context. */
int hdiff_device_count(void);
/* How the fp32 matrix contractions of the attention core and of the 3x3 convolutions are carried out (process-wide, read
 * at launch time):
 *   HDIFF_CONTRACT_F32    (0, default)  fp32-input MFMA (v_mfma_f32_16x16x4_f32 / 32x32x2)
 *   HDIFF_CONTRACT_BF16X3 (1)           each fp32 operand as three bf16 pieces, the six products with i+j<=2 on the bf16
 *                                       MFMA with fp32 accumulation: fp32-class error (about 2^-22 relative per product),
 *                                       about 1.5x faster; shapes it does not cover silently use the fp32 kernels
 *                                       (attention: d_head 16/32, L % 64 == 0; conv: 3x3 stride 1 with Cin % 16 == 0 and a
 *                                       descriptor that carries wp_x3).
 * The initial value comes from the environment variable HDIFF_CONTRACT ("f32" | "bf16x3"). */
enum { HDIFF_CONTRACT_F32 = 0, HDIFF_CONTRACT_BF16X3 = 1 };
int hdiff_set_contraction_mode(int mode);
int hdiff_get_contraction_mode(void);

/* ------------------------------------------------------------------------------------------------------------------
 * Convolutions (K1-K4).  Replaces nn.Conv2d / nn.ConvTranspose2d call sites
 *   DiffusionFreeGuidence/ModelCondition.py:71-75 (DownSample), :82-88 (UpSample), :172,:186,:192 (ResBlock),
 *   :219 (head), :251 (tail) and the packed in/out projections of nn.MultiheadAttention (:189) viewed as 1x1 convs.
 *
 * Weights are consumed in a packed layout wp[tap][CinPad][CoutPad] (CoutPad % 64 == 0, CinPad % 8 == 0, zero padded),
 * produced on the device by hdiff_pack_conv_weight from the PyTorch layouts.
 * ------------------------------------------------------------------------------------------------------------------ */

#define HDIFF_MAX_TAPS 25

/* mode 0: Conv2d weight [Cout][Cin][KH][KW]; mode 1: ConvTranspose2d weight [Cin][Cout][KH][KW].
 * tap t of the packed tensor takes kernel element (tap_ky[t], tap_kx[t]).  accumulate != 0 adds into wp (used to fold
 * DownSample's 3x3 into the centre of its 5x5: ModelCondition.py:75 computes c1(x)+c2(x) on the same input). */
int hdiff_pack_conv_weight(const float* w, float* wp, int mode, int Cout, int Cin, int KH, int KW, int ntaps,
                           const int* tap_ky, const int* tap_kx, int CinPad, int CoutPad, int accumulate,
                           hdiff_stream_t stream);

/* Weights of a standard 3x3 conv ([Cout][Cin][3][3], Cin % 16 == 0) as three bf16 pieces per value, laid out
 * [Cin/16][tap][piece][CoutPad][16] for conv3x3_x3.hip; wp3 holds (Cin/16)*9*3*CoutPad*8 32-bit words.
 * transposed != 0: w is [Cin][Cout][3][3] and is read with mirrored taps -- the weight of the input-gradient convolution
 * of a 3x3 / stride-1 conv (Cout / Cin are those of the gradient conv: its outputs are the forward's inputs). */
int hdiff_pack_conv_weight_x3(const float* w, void* wp3, int Cout, int Cin, int CoutPad, int transposed,
                              hdiff_stream_t stream);
/* The same pack for any launch of the split-bf16 convolution kernel (taps within the 3x3 neighbourhood): tap t reads kernel
 * element (tap_ky[t], tap_kx[t]) of a KH x KW kernel stored [Cout][Cin][KH][KW] (mode 0) or [Cin][Cout][KH][KW] (mode 1,
 * nn.ConvTranspose2d) -- the four output-parity phases of ConvTranspose2d(5, stride 2) are packed this way (9 / 6 / 6 / 4 taps);
 * wp3 holds (Cin/16)*ntaps*3*CoutPad*8 32-bit words. */
int hdiff_pack_conv_weight_x3_taps(const float* w, void* wp3, int mode, int Cout, int Cin, int KH, int KW, int ntaps,
                                   const int* tap_ky, const int* tap_kx, int CoutPad, hdiff_stream_t stream);
/* (ABI 5) Weights of a standard 3x3 conv ([Cout][Cin][3][3], Cin % 16 == 0) as TWO fp16 pieces of w * 2^t, t chosen on the
 * device so that max |w| 2^t lies in [2^14, 2^15): [Cin/16][tap][piece][CoutPad][16] followed by a 4-word tail
 * {scratch, 2^-t, 2^t, 0}; hdiff_pack_conv_weight_h2_words gives the size in 32-bit words.  For the fp16-pair form of the
 * split-operand 3x3 kernel (hdiff_conv_desc.wp_h2; replaces the weight operand of F.conv2d at ModelCondition.py:172, 186). */
int hdiff_pack_conv_weight_h2_words(int Cout, int Cin, int CoutPad, int64_t* words_out);
int hdiff_pack_conv_weight_h2(const float* w, void* wp2, int Cout, int Cin, int CoutPad, hdiff_stream_t stream);
/* (ABI 5) Range of a GroupNorm + Swish output from the GroupNorm weights alone: |swish(gamma * xhat + beta)| <=
 * sqrt(n - 1) * max |gamma| + max |beta| =: A for groups of n = group_elems elements (a normalised value cannot leave
 * [-sqrt(n - 1), sqrt(n - 1)]).  out2[0] = 2^s, out2[1] = 2^-s with gain * A * 2^s in [2^13, 2^14): the power of two by which
 * the fp16-pair conv stages its activations (hdiff_conv_desc.act_scale); gain = 1, or 1 / keep when a dropout mask scaled by
 * 1 / keep sits between the activation and the conv (ModelCondition.py:185).  nn.GroupNorm at ModelCondition.py:169, 182. */
int hdiff_gn_act_scale(const float* gamma, const float* beta, int C, int64_t group_elems, float gain, float* out2,
                       hdiff_stream_t stream);

typedef struct hdiff_conv_desc {
  /* input: virtual channel-concat of x0 [B][C0][H][W] and x1 [B][C1][H][W] (x1 may be NULL with C1 = 0);
   * replaces torch.cat([h, hs.pop()], dim=1) at ModelCondition.py:271 */
  const float* x0;
  const float* x1;
  int C0, C1;
  int B, H, W;
  /* packed weights and optional bias [Cout] */
  const float* wp;
  const float* bias;
  int Cout, CinPad, CoutPad;
  /* optional fused prologue: y = swish(x * gn_scale[b][c] + gn_shift[b][c])  (GroupNorm affine + Swish folded to a
   * per-(sample,channel) scale/shift by hdiff_gn_finalize; zero padding is applied AFTER the activation) */
  const float* gn_scale;
  const float* gn_shift;
  /* optional fused epilogue: + addvec[b][co] (temb/cemb projections, ModelCondition.py:198-200), + residual (h + shortcut(x), :202) */
  const float* addvec;
  const float* residual;
  float* out;      /* [B][Cout][OH][OW] */
  int OH, OW;
  /* geometry: the kernel iterates a virtual output grid VH x VW; input coord = v*in_stride + tap_d; output coord = v*out_s + out_o */
  int VH, VW;
  int in_stride;
  int out_sy, out_oy, out_sx, out_ox;
  int ntaps;
  int tap_dy[HDIFF_MAX_TAPS];
  int tap_dx[HDIFF_MAX_TAPS];
  /* optional split-K workspace (hdiff_conv2d_fwd_workspace floats): small grids with long channel loops are cut into
   * channel slices whose partial sums are reduced in a fixed order; NULL = never split */
  float* splitk_ws;
  int64_t splitk_floats;
  /* optional (ABI 2): the same weights as three bf16 pieces (hdiff_pack_conv_weight_x3).  Used instead of wp when the
   * contraction mode is HDIFF_CONTRACT_BF16X3 and the launch is a stride-1 conv with Cin % 16 == 0 whose taps lie in the 3x3
   * neighbourhood (the plain 3x3 / pad-1 conv; a transposed-conv phase with its (2, py, 2, px) output map), or (ABI 5) a
   * full-resolution 1x1 / stride-1 conv with a ONE-tap pack from hdiff_pack_conv_weight_x3_taps (Cin % 16 == 0, H * W % 256 == 0);
   * NULL = always the fp32-input MFMA.  Tap order: hdiff_pack_conv_weight_x3 stores tap t = (t / 3, t % 3), so a plain 3x3
   * launch must list tap_dy[t] = t / 3 - 1, tap_dx[t] = t % 3 - 1 (any other order of the nine taps, and any list with a
   * repeated tap, runs the fp32 kernel on wp instead); a pack made by hdiff_pack_conv_weight_x3_taps holds the taps in
   * the order of the list it was given, and the descriptor has to list them in that same order. */
  const void* wp_x3;
  /* optional (ABI 5): the fp16-pair form of the same kernel for the plain 3x3 / pad-1 conv WITH the GroupNorm + Swish prologue
   * (three fp16 piece products instead of six bf16 ones).  wp_h2 = hdiff_pack_conv_weight_h2 of the same weights, act_scale =
   * the two floats of hdiff_gn_act_scale for the GroupNorm whose statistics gn_scale / gn_shift carry.  Preferred over wp_x3
   * when both are set, the mode is HDIFF_CONTRACT_BF16X3 and the launch qualifies; either may be NULL.  act_scale is the
   * caller's statement that every staged activation (after the prologue, if any) is below 2^15 / act_scale[0] in magnitude:
   * with gn_scale / gn_shift that are not GroupNorm statistics of x0 / x1 a value can exceed it, the fp16 conversion then gives
   * inf and the output NaN (never silently wrong). */
  const void* wp_h2;
  const float* act_scale;
} hdiff_conv_desc;

int hdiff_conv2d_fwd_workspace(const hdiff_conv_desc* d, int64_t* floats_out);
int hdiff_conv2d_fwd(const hdiff_conv_desc* d, hdiff_stream_t stream);

/* Weight gradient of hdiff_conv2d_fwd (autograd of the conv weights, TrainCondition.py:60).  Same geometry fields as the
 * forward descriptor; dy is the gradient of the forward's `out`.  The kernel writes `nsplit` packed partial slabs
 * dwp[nsplit][ntaps][CinPad][CoutPad] (size from hdiff_conv2d_wgrad_workspace); hdiff_conv_wgrad_unpack sums them in a
 * fixed order into the PyTorch layout (mode / taps as in hdiff_pack_conv_weight; tap_ky < 0 skips a tap). */
typedef struct hdiff_conv_wgrad_desc {
  const float* x0;
  const float* x1;
  int C0, C1;
  int B, H, W;
  const float* gn_scale;
  const float* gn_shift;
  const float* dy;      /* [B][Cout][OH][OW] */
  int Cout, CinPad, CoutPad;
  int OH, OW;
  int VH, VW;
  int in_stride;
  int out_sy, out_oy, out_sx, out_ox;
  int ntaps;
  int tap_dy[HDIFF_MAX_TAPS];
  int tap_dx[HDIFF_MAX_TAPS];
} hdiff_conv_wgrad_desc;
int hdiff_conv2d_wgrad_workspace(const hdiff_conv_wgrad_desc* d, int* nsplit_out, int64_t* floats_out);
int hdiff_conv2d_wgrad(const hdiff_conv_wgrad_desc* d, float* dwp, int nsplit, hdiff_stream_t stream);
int hdiff_conv_wgrad_unpack(const float* dwp, int nsplit, float* dw, int mode, int Cout, int Cin, int KH, int KW, int ntaps,
                            const int* tap_ky, const int* tap_kx, int CinPad, int CoutPad, int accumulate,
                            hdiff_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * GroupNorm statistics (K5).  Replaces nn.GroupNorm(32, C) at ModelCondition.py:170,184,249 (the normalisation itself
 * and the Swish at :22-24 are applied inside the consuming convolution's prologue).
 *   hdiff_gn_stats:    partial (count, mean, M2) per (sample, group, split) over the virtual concat input
 *   hdiff_gn_finalize: combines the partials (Chan) and writes scale[b][c] = rstd*gamma[c], shift[b][c] = beta[c]-mean*scale
 * ws must hold B*G*nsplit*3 floats.
 * ------------------------------------------------------------------------------------------------------------------ */
int hdiff_gn_stats(const float* x0, const float* x1, int C0, int C1, int B, int HW, int G, int nsplit, float* ws,
                   hdiff_stream_t stream);
int hdiff_gn_finalize(const float* ws, int B, int C, int G, int nsplit, const float* gamma, const float* beta, float eps,
                      float* scale, float* shift, float* mean_out /*[B][G] or NULL*/, float* rstd_out /*[B][G] or NULL*/,
                      hdiff_stream_t stream);
/* hdiff_gn_stats + hdiff_gn_finalize behind one entry point.  nsplit == 1 (planes up to 64x64: every (sample, group) is
 * one workgroup) -> ONE launch, the workgroup folds its own statistics into scale / shift; nsplit > 1 -> the streaming
 * pass and the small merge kernel, as two launches of the same call.  Results are bit-identical to the two-call form. */
int hdiff_gn_scale_shift(const float* x0, const float* x1, int C0, int C1, int B, int HW, int G, int nsplit, float* ws,
                         const float* gamma, const float* beta, float eps, float* scale, float* shift,
                         hdiff_stream_t stream);
/* Backward of GroupNorm + Swish (the conv prologue): dA is the gradient w.r.t. the activated tensor [B][C0+C1][HW];
 * mean/rstd [B][G] come from hdiff_gn_finalize.  Writes dx0 [B][C0][HW], dx1 [B][C1][HW], dgamma [C], dbeta [C].
 * ws: 2*B*C + 2*B*G floats. */
int hdiff_gn_swish_bwd(const float* x0, const float* x1, int C0, int C1, int B, int HW, int G, const float* dA,
                       const float* mean, const float* rstd, const float* gamma, const float* beta, float* ws, float* dx0,
                       float* dx1, float* dgamma, float* dbeta, hdiff_stream_t stream);
/* The same for GroupNorm WITHOUT Swish (AttnBlock's pre-norm, ModelCondition.py:103): dY is the gradient w.r.t. the
 * normalised tensor [B][C][HW].  ws: 2*B*C + 2*B*G floats. */
int hdiff_gn_affine_bwd(const float* x, int C, int B, int HW, int G, const float* dY, const float* mean, const float* rstd,
                        const float* gamma, const float* beta, float* ws, float* dx, float* dgamma, float* dbeta,
                        hdiff_stream_t stream);
/* dvec[b][c] = sum_hw dy[b][c][:] (gradient of the per-sample channel vector) and dbias[c] = sum_b dvec[b][c]; either may be NULL */
int hdiff_bias_addvec_grad(const float* dy, int B, int C, int HW, float* dvec, float* dbias, hdiff_stream_t stream);
/* y = x*scale[b][c] + shift[b][c]: GroupNorm without Swish (AttnBlock, ModelCondition.py:103) */
int hdiff_gn_affine_apply(const float* x, const float* scale, const float* shift, float* y, int B, int C, int HW,
                          hdiff_stream_t stream);
/* Stand-alone y = swish(x*scale+shift) (used by tests and by the training path). */
int hdiff_gn_swish_apply(const float* x, const float* scale, const float* shift, float* y, int B, int C, int HW,
                         hdiff_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Multi-head self-attention core (K9).  Replaces the softmax(QK^T/sqrt(d))V of nn.MultiheadAttention(C, 8) called as
 * attn(h,h,h) at ModelCondition.py:189,204-208.  qkv is the output of the packed in-projection viewed as a 1x1 conv:
 * [B][3C][L] with rows [Q | K | V], head h owning rows h*d..h*d+d-1 of each third.  o is [B][C][L].
 * Flash style: the L x L score matrix is never materialised.  d = C/heads must be one of 4, 8, 12, 16, 24, 32, 48, 64 (48 / 64: slower one-tile-per-wave kernels; not shapes of the default model).
 * ------------------------------------------------------------------------------------------------------------------ */
int hdiff_mha_flash_fwd(const float* qkv, float* o, float* lse2 /*[B][heads][L] or NULL*/, int B, int C, int heads, int L,
                        hdiff_stream_t stream);
/* The same call with a caller-provided scratch buffer.  In the bf16x3 contraction mode (hdiff_set_contraction_mode) the
 * operands are then split ONCE into three bf16 pieces each (a streaming pass into `ws`) and the attention kernel runs on the
 * pre-split tensors (attention_x3p.hip); without a workspace -- or for a shape the pre-split kernel does not cover, for
 * which the query returns 0 bytes -- the kernels split inside their loop.  In the fp32 mode `ws` is not touched.  The size
 * is a function of the shape only (B * 3C * L * 6 bytes when covered: d_head 16 / 32, L a multiple of 256, L >= 512). */
int hdiff_mha_flash_fwd_workspace(int B, int C, int heads, int L, int64_t* bytes_out);
int hdiff_mha_flash_fwd_ws(const float* qkv, float* o, float* lse2 /*[B][heads][L] or NULL*/, int B, int C, int heads, int L,
                           void* ws, int64_t ws_bytes, hdiff_stream_t stream);
/* Single-head attention with a head wider than 64 channels: softmax(q k^T * C^-1/2) v with d_head = C, the core of the
 * reference's AttnBlock (ModelCondition.py:109-116; dead code there, built for completeness: one workgroup per query row,
 * L + C floats of LDS).  qkv [B][3C][L] rows [q | k | v], o [B][C][L].  Heads of width <= 64: hdiff_mha_flash_fwd, heads = 1. */
int hdiff_mha_wide_fwd(const float* qkv, float* o, int B, int C, int L, hdiff_stream_t stream);
/* Backward of the single-head core of any width (autograd through AttnBlock.forward, ModelCondition.py:109-116): dqkv
 * [B][3C][L] from dO [B][C][L]; probabilities are recomputed.  ws: 2*B*L floats (log-sum-exp and delta per query). */
int hdiff_mha_wide_bwd(const float* qkv, const float* d_o, float* dqkv, float* ws, int B, int C, int L, hdiff_stream_t stream);
/* Backward of the core (autograd of nn.MultiheadAttention, TrainCondition.py:60): dqkv [B][3C][L] from dO [B][C][L].
 * lse2 is the forward's log2-domain log-sum-exp; delta is a [B][heads][L] workspace (rowsum(dO o O), written here).
 * P is recomputed, never stored; five MFMA products per tile in ONE kernel: a workgroup owns a key range (dK, dV in
 * registers) and adds its dQ tiles to the partial slab of that range in ws, summed in range order afterwards -- every slab
 * word is only ever touched by one thread, in program order: bitwise reproducible.  In the bf16x3 contraction mode at
 * d_head 16 or 32 (L a multiple of 256, >= 512) the five products run on the bf16 matrix core (attention_bwd_x3.hip); ws then
 * also holds the five bf16 piece tensors of Q, K, K^T, V, dO (30 bytes per element of a [B][C][L] tensor) and the slab
 * words after a range's first key block are accumulated by in-order L2 float adds.  hdiff_mha_flash_bwd_workspace gives
 * the size of ws in floats: a function of the shape only, large enough for either contraction mode (0: ws may be NULL). */
int hdiff_mha_flash_bwd_workspace(int B, int C, int heads, int L, int64_t* n_floats);
int hdiff_mha_flash_bwd(const float* qkv, const float* o, const float* d_o, const float* lse2, float* delta, float* dqkv,
                        float* ws, int B, int C, int heads, int L, hdiff_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Small dense layers (K6, K10).  y[b][j] (+)= bias[j] + sum_k W[j][k] * f(x_row(b)[k]),  f = identity or Swish.
 * If idx != NULL the input row is x[idx[b]] (nn.Embedding gather, ModelCondition.py:38,56) with idx clamped to
 * [0, n_rows) so a bad index can never fault the GPU (the host layer raises IndexError like nn.Embedding); otherwise x[b].
 * Replaces nn.Linear / nn.Embedding at ModelCondition.py:38-43, 56-61, 174-181.
 * ------------------------------------------------------------------------------------------------------------------ */
int hdiff_linear_rows(const float* x, const int64_t* idx, int n_rows, const float* W, const float* bias, float* y, int B,
                      int K, int N, int swish_input, int accumulate, hdiff_stream_t stream);
/* Every per-block projection of the time and label embeddings in one launch (ResBlock.forward, ModelCondition.py:199-200):
 *   y_j[b][n] = (sum_k swish(x0[b][k]) w0_j[n][k] + b0_j[n]) + (sum_k swish(x1[b][k]) w1_j[n][k] + b1_j[n]),  j < njobs
 * bit-identical to hdiff_linear_rows(x0 -> y_j, swish) followed by hdiff_linear_rows(x1 -> y_j, swish, accumulate).
 * `jobs` lives in DEVICE memory (pointers to [n][K] weights, [n] biases, the [B][n] output); jobs[j].first = sum of the n
 * of the jobs before j; total_n = sum of all n; w1 / b1 may be NULL (no second term). */
typedef struct hdiff_linear_job {
  const float* w0; const float* b0; const float* w1; const float* b1; float* y;
  int n; int first;
} hdiff_linear_job;
int hdiff_linear_rows_multi(const float* x0 /*[B][K]*/, const float* x1 /*[B][K] or NULL*/, const hdiff_linear_job* jobs,
                            int njobs, int total_n, int B, int K, hdiff_stream_t stream);

/* Backward of hdiff_linear_rows: dW [N][K] (+)= dy^T f(x), db [N] (+)= sum_b dy, and, if dx != NULL, dx = (dy W) o f'(x).
 * With idx != NULL, dx is the [n_rows][K] gradient of the gathered table and is ACCUMULATED into (zero it first); row
 * pad_row (>= 0) gets no gradient (nn.Embedding(padding_idx=0), ModelCondition.py:57). */
int hdiff_linear_rows_bwd(const float* x, const int64_t* idx, int n_rows, const float* W, const float* dy, float* dx,
                          float* dW, float* db, int B, int K, int N, int swish_input, int accumulate, int pad_row,
                          hdiff_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Diffusion process (K11-K13), DiffusionFreeGuidence/DiffusionCondition.py.
 * ------------------------------------------------------------------------------------------------------------------ */
/* q_sample (:43-44): x_t = sa[t[b]]*x0 + sb[t[b]]*noise ; sa/sb are the fp32 casts of the float64 schedule buffers, T their
 * length: t[b] is clamped to [0, T) in the kernel (a bad index never faults the GPU; the host raises like extract's gather). */
int hdiff_q_sample(const float* x0, const float* noise, const int64_t* t, const float* sqrt_ab, const float* sqrt_1mab,
                   float* xt, int B, int per_sample, int T, hdiff_stream_t stream);
/* unreduced squared error (:45): loss = (eps_hat - noise)^2 */
int hdiff_sq_err(const float* a, const float* b, float* out, int64_t n, hdiff_stream_t stream);
/* its backward: da = 2*(a-b)*dloss */
int hdiff_sq_err_bwd(const float* a, const float* b, const float* dloss, float* da, int64_t n, hdiff_stream_t stream);
/* One ancestral step (:74-80, :89-96):
 *   eps = (1+w)*eps_c - w*eps_u  (w is the Python float of the reference; (1+w) is formed in double, then cast) ; mean = coeff1[t]*x - coeff2[t]*eps ; x_next = mean + sigma[t]*z  (z = 0 when t == 0)
 * step_ptr points at the device-resident current time step (int32) so the launch is hipGraph-replayable; when
 * noise == NULL z is drawn in-kernel (Philox4x32-10 + Box-Muller, counter = (seed, step, element)).  nan_flag (int32)
 * is OR-ed with 1 if any output is NaN (the reference's per-step assert, :96, evaluated once after the loop). */
int hdiff_ddpm_step(const float* x, const float* eps_c, const float* eps_u, const float* noise, float* x_next,
                    const float* coeff1, const float* coeff2, const float* sigma, const int32_t* step_ptr,
                    int T /* length of the three tables: the step read from step_ptr is clamped into [0, T) */, double w,
                    uint64_t seed, int32_t* nan_flag, int64_t n, hdiff_stream_t stream);
/* The same update with the loop's bookkeeping folded in, so that one captured denoising step is the UNet launches plus
 * this ONE kernel (DiffusionCondition.py:87-96).  After the update the workgroup that finishes last
 *   - writes the next time step, *step_ptr - 1, to *step_ptr and (clamped at 0) to t_next[0 .. t_count): the time vector
 *     `t = x_t.new_ones([B]) * time_step` (:89) of the next replay;
 * and every workgroup also stores x_next to x_dup0 / x_dup1 when they are given (the two halves of the 2B-batched
 * cond + uncond UNet input of the next step, :76-77).  done_counter: one uint32 in device memory, zero before the first
 * launch; it counts finished workgroups and wraps back to zero by itself (no reset between replays). */
typedef struct hdiff_ddpm_loop_desc {
  const float* x; const float* eps_c; const float* eps_u;
  const float* noise;                 /* NULL: in-kernel Philox noise */
  float* x_next;
  const float* coeff1; const float* coeff2; const float* sigma;   /* [T] fp32 */
  int32_t* step_ptr;
  int T;
  double w;
  uint64_t seed;
  int32_t* nan_flag;
  int64_t n;
  float* x_dup0; float* x_dup1;       /* optional */
  int64_t* t_next; int t_count;       /* optional (t_count = 0) */
  uint32_t* done_counter;
} hdiff_ddpm_loop_desc;
int hdiff_ddpm_step_loop(const hdiff_ddpm_loop_desc* d, hdiff_stream_t stream);
/* step bookkeeping for the captured loop: t[b] = *step for all b (int64 vector for the embedding gather) */
int hdiff_fill_t(int64_t* t, const int32_t* step_ptr, int B, hdiff_stream_t stream);
int hdiff_step_decrement(int32_t* step_ptr, hdiff_stream_t stream);
/* ------------------------------------------------------------------------------------------------------------------
 * Image-conditioned sampler of the reference's second tree (diffusion/Diffusion.py:182-269, diffusion/Model.py).
 *   hdiff_ddim_step        one deterministic DDIM update (Diffusion.py:259-263, eta = 0):
 *                          y0 = (y - eps*tab[k][0]) / tab[k][1] ; y_next = tab[k][2]*y0 + tab[k][3]*eps, k = *step_ptr;
 *                          tab[k] = {sqrt(1-at), sqrt(at), sqrt(at_next), sqrt(1-at_next)} in fp32 (host builds it with the
 *                          reference's own tensor ops); nan_flag is OR-ed with 1 on a NaN output
 *   hdiff_fill_from_table  dst[i] = table[*idx] for i < n: the DDIM time-step vector t of step k (Diffusion.py:249)
 *   hdiff_resize_nearest   F.interpolate(mode="nearest") of [BC][H][W] to [BC][OH][OW] (skip tensors, Model.py:503-504)
 *   hdiff_avgpool_global   nn.AdaptiveAvgPool2d((1,1)) of [BC][HW] -> [BC] (ConditionalEmbedding, Model.py:124,150)
 *   hdiff_concat2          out[b] = [a[b] (n0 floats) | b[b] (n1 floats)]: torch.cat([input_image, y_t], dim=1) of
 *                          Diffusion.py:229,252 (3 + 3 channels: too narrow for the conv's two-pointer input)
 * ------------------------------------------------------------------------------------------------------------------ */
int hdiff_ddim_step(const float* y, const float* eps, float* y_next, const float* tab, const int32_t* step_ptr,
                    int nsteps /* rows of tab: k = *step_ptr is clamped into [0, nsteps) */, int32_t* nan_flag, int64_t n,
                    hdiff_stream_t stream);
int hdiff_fill_from_table(int64_t* dst, const int32_t* table, const int32_t* idx, int table_len /* *idx is clamped */, int n,
                          hdiff_stream_t stream);
int hdiff_resize_nearest(const float* x, float* y, int BC, int H, int W, int OH, int OW, hdiff_stream_t stream);
int hdiff_avgpool_global(const float* x, float* y, int BC, int HW, hdiff_stream_t stream);
int hdiff_concat2(const float* a, const float* b, float* out, int B, int64_t n0, int64_t n1, hdiff_stream_t stream);
/* final clip (:98) */
int hdiff_clip(const float* x, float* y, float lo, float hi, int64_t n, hdiff_stream_t stream);
/* out = a*x + b*y (y may be NULL): bias merges and other weight-preparation arithmetic */
int hdiff_axpby(float a, const float* x, float b, const float* y, float* out, int64_t n, hdiff_stream_t stream);

/* (ABI 6) The tail of an optimizer step over a LIST of tensors, replacing the caller's torch.nn.utils.clip_grad_norm_(params, grad_clip) and
 * torch.optim.AdamW.step() (reference call sites: DiffusionFreeGuidence/TrainCondition.py:61-63, :39-40).  `table` (device) holds one entry per
 * tensor; `chunks` (device, 2 ints per chunk: tensor index, chunk index inside the tensor) cuts the work into pieces of hdiff_opt_chunk()
 * elements -- both built once by the host.  Pointers need 4-byte alignment only.
 *   hdiff_grad_norm_clip_coef: partial[nchunks] scratch; norm_coef[0] = sqrt(sum g^2) over all tensors (fixed summation order, float64 final
 *                              sum), norm_coef[1] = min(1, max_norm / (norm + 1e-6)).
 *   hdiff_adamw_step:          g *= norm_coef[1] (in place, skipped when norm_coef is NULL); p *= 1 - lr wd; m += (g - m)(1 - beta1);
 *                              v = beta2 v + (1 - beta2) g^2; p -= lr / (1 - beta1^step) * m / (sqrt(v) / sqrt(1 - beta2^step) + eps)
 *                              (torch.optim.AdamW's single-tensor formulas in its order of operations; step counts from 1). */
typedef struct { float* p; float* g; float* m; float* v; long long n; } hdiff_opt_tensor;
int hdiff_opt_chunk(void);
int hdiff_grad_norm_clip_coef(const hdiff_opt_tensor* table, const int* chunks, int nchunks, float* partial, float max_norm,
                              float* norm_coef, hdiff_stream_t stream);
int hdiff_adamw_step(const hdiff_opt_tensor* table, const int* chunks, int nchunks, const float* norm_coef, double lr, double beta1,
                     double beta2, double eps, double weight_decay, int64_t step, hdiff_stream_t stream);
/* nn.Dropout (train mode, ModelCondition.py:185): keep-mask scaled by 1/keep from the Philox stream, and out = a*b */
int hdiff_dropout_mask(float* out, int64_t n, float keep, uint64_t seed, uint64_t offset, hdiff_stream_t stream);
int hdiff_mul(const float* a, const float* b, float* out, int64_t n, hdiff_stream_t stream);
/* standard-normal fill with the same Philox stream (used for in-graph noise and for tests of the generator) */
int hdiff_randn(float* out, int64_t n, uint64_t seed, uint64_t offset, hdiff_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * hipGraph helpers: capture everything enqueued on `stream` between begin/end, replay it later.
 * ------------------------------------------------------------------------------------------------------------------ */
int hdiff_graph_begin(hdiff_stream_t stream);
int hdiff_graph_end(hdiff_stream_t stream, void** graph_exec_out);
int hdiff_graph_launch(void* graph_exec, hdiff_stream_t stream);
int hdiff_graph_destroy(void* graph_exec);

/* HIP event timing on the launch stream (bench.py's roofline leg) */
int hdiff_event_create(void** ev);
int hdiff_event_record(void* ev, hdiff_stream_t stream);
int hdiff_event_elapsed_ms(void* start, void* stop, float* ms); /* synchronises on stop */
int hdiff_event_destroy(void* ev);

#ifdef __cplusplus
}
#endif
#endif /* HDIFF_H_ */
