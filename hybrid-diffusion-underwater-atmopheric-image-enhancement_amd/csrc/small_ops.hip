// Small dense layers, the DDPM elementwise steps and the in-graph normal generator (gfx950).
// All HBM- or latency-bound; none of them shows up next to attention/conv at the target sizes, so they are written
// for clarity: one wave per output of a GEMV, 16-byte grid-stride loops for the elementwise passes.
#include "common.h"

// The DDPM elementwise steps must round exactly like the reference's separate mul / sub / add tensor ops:
// fma contraction is forbidden in this file (hipcc defaults to -ffp-contract=fast for device code, and HIP's __fmul_rn
// is a plain, contractible '*'): the Makefile compiles it with -ffp-contract=off and the pragma repeats it.
#pragma clang fp contract(off)

using namespace hdiff;

namespace {

// ---------------------------------------------------------------------------------------------------------------------
// y[b][j] (+)= bias[j] + sum_k W[j][k] * f(x_row(b)[k]); one wave per (b, j).
// Reference: nn.Embedding gather + nn.Linear (+Swish on the input) at ModelCondition.py:38-43, 56-61, 174-181.
// ---------------------------------------------------------------------------------------------------------------------
__global__ void linear_rows_kernel(const float* __restrict__ x, const int64_t* __restrict__ idx,
                                   const float* __restrict__ W, const float* __restrict__ bias, float* __restrict__ y,
                                   int B, int K, int N, int n_rows, int swish_input, int accumulate) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (wave >= B * N) return;
  const int b = wave / N, j = wave - b * N;
  int64_t row = idx ? idx[b] : (int64_t)b;
  if (idx && n_rows > 0) row = row < 0 ? 0 : (row >= n_rows ? n_rows - 1 : row);   // never fault on a bad index; the host validates
  const float* xr = x + (size_t)row * K;
  const float* wr = W + (size_t)j * K;
  float s = 0.f;
  for (int k = lane; k < K; k += 64) {
    float v = xr[k];
    if (swish_input) v = swishf(v);
    s = fmaf(wr[k], v, s);
  }
  s = wave_sum(s);
  if (lane == 0) {
    if (bias) s += bias[j];
    y[(size_t)b * N + j] = accumulate ? y[(size_t)b * N + j] + s : s;
  }
}

// All per-block projections of the time / label embeddings in ONE launch (ResBlock.forward, ModelCondition.py:199-200:
// h += temb_proj(temb)[:, :, None, None]; h += cond_proj(cemb)[:, :, None, None], each Swish -> Linear): job j computes
//   y_j[b][n] = (sum_k swish(x0[b][k]) w0_j[n][k] + b0_j[n]) + (sum_k swish(x1[b][k]) w1_j[n][k] + b1_j[n])
// with the SAME roundings as the two accumulating linear_rows launches it replaces (each dot product reduced on its own,
// bias added, then the two results added).  One wave per output; the job of a wave is found from the prefix table.
__global__ void linear_rows_multi_kernel(const float* __restrict__ x0, const float* __restrict__ x1,
                                         const hdiff_linear_job* __restrict__ jobs, int njobs, int total_n, int B, int K) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (wave >= B * total_n) return;
  const int b = wave / total_n, gn = wave - b * total_n;
  int j = 0;
  while (j + 1 < njobs && jobs[j + 1].first <= gn) ++j;          // wave-uniform, at most a few dozen jobs
  const hdiff_linear_job job = jobs[j];
  const int n = gn - job.first;
  const float* xr0 = x0 + (size_t)b * K;
  const float* wr0 = job.w0 + (size_t)n * K;
  float s = 0.f;
  for (int k = lane; k < K; k += 64) s = fmaf(wr0[k], swishf(xr0[k]), s);
  s = wave_sum(s);
  float t = 0.f;
  if (job.w1 != nullptr) {
    const float* xr1 = x1 + (size_t)b * K;
    const float* wr1 = job.w1 + (size_t)n * K;
    for (int k = lane; k < K; k += 64) t = fmaf(wr1[k], swishf(xr1[k]), t);
    t = wave_sum(t);
  }
  if (lane == 0) {
    if (job.b0) s += job.b0[n];
    if (job.w1 != nullptr) {
      if (job.b1) t += job.b1[n];
      s = s + t;
    }
    job.y[(size_t)b * job.n + n] = s;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Philox4x32-10 counter-based generator + Box-Muller: 4 normals per counter.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
  const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
  const uint32_t n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
  const uint32_t n3 = (uint32_t)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

__device__ __forceinline__ void philox4x32_10(uint64_t seed, uint64_t ctr_lo, uint64_t ctr_hi, uint32_t (&out)[4]) {
  uint32_t c[4] = {(uint32_t)ctr_lo, (uint32_t)(ctr_lo >> 32), (uint32_t)ctr_hi, (uint32_t)(ctr_hi >> 32)};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}

__device__ __forceinline__ float4 normal4(uint64_t seed, uint64_t ctr_lo, uint64_t ctr_hi) {
  uint32_t r[4];
  philox4x32_10(seed, ctr_lo, ctr_hi, r);
  // u in (0,1]: (r + 1) * 2^-32 ; Box-Muller on two pairs
  const float u0 = ((float)(r[0] >> 8) + 1.0f) * (1.0f / 16777216.0f);
  const float u1 = ((float)(r[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float u2 = ((float)(r[2] >> 8) + 1.0f) * (1.0f / 16777216.0f);
  const float u3 = ((float)(r[3] >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float ra = sqrtf(-2.0f * logf(u0)), rb = sqrtf(-2.0f * logf(u2));
  float s0, c0, s1, c1;
  sincosf(6.283185307179586f * u1, &s0, &c0);
  sincosf(6.283185307179586f * u3, &s1, &c1);
  return make_float4(ra * c0, ra * s0, rb * c1, rb * s1);
}

__global__ void randn_kernel(float* __restrict__ out, int64_t n, uint64_t seed, uint64_t offset) {
  const int64_t nq = (n + 3) >> 2;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (int64_t)gridDim.x * blockDim.x) {
    const float4 z = normal4(seed, (uint64_t)q, offset);
    const int64_t i = q << 2;
    if (i + 3 < n) {
      *reinterpret_cast<float4*>(out + i) = z;
    } else {
      const float zz[4] = {z.x, z.y, z.z, z.w};
      for (int e = 0; e < 4 && i + e < n; ++e) out[i + e] = zz[e];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Diffusion elementwise steps (DiffusionCondition.py)
// ---------------------------------------------------------------------------------------------------------------------
__global__ void q_sample_kernel(const float* __restrict__ x0, const float* __restrict__ noise,
                                const int64_t* __restrict__ t, const float* __restrict__ sa,
                                const float* __restrict__ sb, float* __restrict__ xt, int per_sample, int T) {
  const int b = blockIdx.y;
  int64_t tb = t[b];
  tb = tb < 0 ? 0 : (tb >= T ? T - 1 : tb);     // never fault on a bad index; the host validates (extract raises)
  const float a = sa[tb], c = sb[tb];
  const size_t base = (size_t)b * per_sample;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < per_sample; i += gridDim.x * blockDim.x)
    xt[base + i] = a * x0[base + i] + c * noise[base + i];   // mul, mul, add: the reference's roundings (contraction is off)
}

__global__ void sq_err_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                              int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float d = a[i] - b[i];
    out[i] = d * d;
  }
}

// x_next = coeff1[t]*x - coeff2[t]*((1+w)*eps_c - w*eps_u) + sigma[t]*z   (DiffusionCondition.py:78-79, 95)
// Written with separate multiplies/adds in the reference's order (no fma contraction across terms) so that the CPU
// oracle and this kernel round identically given identical eps.
// x and x_next may be the same buffer (the sampler updates in place): neither is __restrict__.
struct DdpmStepK {
  const float* x; const float* eps_c; const float* eps_u; const float* noise; float* x_next;
  const float* coeff1; const float* coeff2; const float* sigma;
  int32_t* step_ptr; int T; float w1, w; uint64_t seed; int32_t* nan_flag; int64_t n;
  float* x_dup0; float* x_dup1; int64_t* t_next; int t_count; unsigned* done_counter;     // loop bookkeeping (all optional)
};
__global__ void ddpm_step_kernel(const DdpmStepK p) {
  int step = *p.step_ptr;
  step = step < 0 ? 0 : (step >= p.T ? p.T - 1 : step);        // never index outside the schedule tables, whatever the counter holds
  const float c1 = p.coeff1[step], c2 = p.coeff2[step], sg = p.sigma[step];
  const bool add_noise = step > 0;
  const float* x = p.x;
  float* x_next = p.x_next;
  bool bad = false;
  const int64_t n = p.n, nq = (n + 3) >> 2;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i0 = q << 2;
    float z[4] = {0.f, 0.f, 0.f, 0.f};
    if (add_noise && p.noise == nullptr) {
      const float4 zz = normal4(p.seed, (uint64_t)q, (uint64_t)(uint32_t)step);
      z[0] = zz.x; z[1] = zz.y; z[2] = zz.z; z[3] = zz.w;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t i = i0 + e;
      if (i < n) {
        if (add_noise && p.noise != nullptr) z[e] = p.noise[i];
        const float eps = p.w1 * p.eps_c[i] - p.w * p.eps_u[i];
        const float mean = c1 * x[i] - c2 * eps;
        const float v = add_noise ? mean + sg * z[e] : mean;
        bad |= (v != v);
        x_next[i] = v;
        if (p.x_dup0) p.x_dup0[i] = v;
        if (p.x_dup1) p.x_dup1[i] = v;
      }
    }
  }
  if (__any(bad)) {
    if ((threadIdx.x & 63) == 0) atomicOr(p.nan_flag, 1);
  }
  // Loop bookkeeping of the captured sampler step (DiffusionCondition.py:87-89: `for time_step in reversed(range(T))`,
  // `t = x_t.new_ones([B]) * time_step`): the workgroup that finishes LAST -- every other one has read *step_ptr by then --
  // decrements the device-resident step and writes the next step's time vector.  The counter wraps back to 0 by itself.
  if (p.done_counter != nullptr) {
    __shared__ int is_last;
    __syncthreads();
    if (threadIdx.x == 0) is_last = atomicInc(p.done_counter, gridDim.x - 1) == gridDim.x - 1;
    __syncthreads();
    if (is_last) {
      const int next = *p.step_ptr - 1;
      for (int i = threadIdx.x; i < p.t_count; i += blockDim.x) p.t_next[i] = (int64_t)(next < 0 ? 0 : next);
      if (threadIdx.x == 0) *p.step_ptr = next;
    }
  }
}

__global__ void fill_t_kernel(int64_t* t, const int32_t* step_ptr, int B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) t[i] = (int64_t)(*step_ptr);
}
__global__ void step_decrement_kernel(int32_t* step_ptr) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *step_ptr = *step_ptr - 1;
}
__global__ void clip_kernel(const float* __restrict__ x, float* __restrict__ y, float lo, float hi, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = fminf(fmaxf(x[i], lo), hi);
}
__global__ void axpby_kernel(float a, const float* __restrict__ x, float b, const float* __restrict__ y,
                             float* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = y ? a * x[i] + b * y[i] : a * x[i];   // separate roundings (mul, mul, add): contraction is off in this file
}


// ---------------------------------------------------------------------------------------------------------------------
// Backward of linear_rows (autograd of nn.Linear / nn.Embedding, TrainCondition.py:60).  Tiny shapes (B <= 128,
// K, N <= 512): one thread per output, samples walked in order so the sums are reproducible.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float dswishf(float v) {
  const float sg = 1.0f / (1.0f + __expf(-v));
  return sg * (1.0f + v * (1.0f - sg));
}

__global__ void linear_rows_bwd_w_kernel(const float* __restrict__ x, const int64_t* __restrict__ idx, int n_rows,
                                         const float* __restrict__ dy, float* __restrict__ dW, float* __restrict__ db, int B,
                                         int K, int N, int swish_input, int accumulate) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * K) return;
  const int n = i / K, k = i - n * K;
  float s = 0.f, sb = 0.f;
  for (int b = 0; b < B; ++b) {
    int64_t row = idx ? idx[b] : (int64_t)b;
    if (idx && n_rows > 0) row = row < 0 ? 0 : (row >= n_rows ? n_rows - 1 : row);
    float v = x[(size_t)row * K + k];
    if (swish_input) v = swishf(v);
    const float g = dy[(size_t)b * N + n];
    s = fmaf(g, v, s);
    sb += g;
  }
  dW[i] = accumulate ? dW[i] + s : s;
  if (k == 0 && db) db[n] = accumulate ? db[n] + sb : sb;
}

// g = sum_n dy[n] * W[n][k]: eight independent partial sums so that eight loads are in flight per lane (a single fma chain
// waits one L2 round trip per term: 86 us for a 4 x 512 x 256 layer, 8 workgroups on the whole chip); fixed combination order.
__device__ __forceinline__ float dot_rows(const float* __restrict__ dyb, const float* __restrict__ Wk, int N, int K) {
  float g[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int n = 0;
  for (; n + 8 <= N; n += 8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] = fmaf(dyb[n + j], Wk[(size_t)(n + j) * K], g[j]);
  }
  for (; n < N; ++n) g[0] = fmaf(dyb[n], Wk[(size_t)n * K], g[0]);
  return ((g[0] + g[1]) + (g[2] + g[3])) + ((g[4] + g[5]) + (g[6] + g[7]));
}

__global__ void linear_rows_bwd_x_kernel(const float* __restrict__ x, const int64_t* __restrict__ idx, int n_rows,
                                         const float* __restrict__ W, const float* __restrict__ dy, float* __restrict__ dx,
                                         int B, int K, int N, int swish_input, int pad_row) {
  if (idx) {
    // gradient of the gathered table rows: one thread per column walks the samples in order (no atomics)
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    for (int b = 0; b < B; ++b) {
      int64_t row = idx[b];
      if (n_rows > 0) row = row < 0 ? 0 : (row >= n_rows ? n_rows - 1 : row);
      if (row == pad_row) continue;     // nn.Embedding(padding_idx): the padding row never receives a gradient
      float g = dot_rows(dy + (size_t)b * N, W + k, N, K);
      if (swish_input) g *= dswishf(x[(size_t)row * K + k]);
      dx[(size_t)row * K + k] += g;
    }
    return;
  }
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * K) return;
  const int b = i / K, k = i - b * K;
  float g = dot_rows(dy + (size_t)b * N, W + k, N, K);
  if (swish_input) g *= dswishf(x[i]);
  dx[i] = g;
}

// d eps_hat of the unreduced squared error: 2 * (a - b) * dloss  (DiffusionCondition.py:45)
__global__ void sq_err_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ dl,
                                  float* __restrict__ da, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    da[i] = 2.0f * (a[i] - b[i]) * dl[i];
}

// Bernoulli keep-mask scaled by 1/keep (nn.Dropout in train mode, ModelCondition.py:185), same Philox stream as randn
__global__ void dropout_mask_kernel(float* __restrict__ out, int64_t n, float keep, uint64_t seed, uint64_t offset) {
  const float scale = 1.0f / keep;
  const int64_t nq = (n + 3) >> 2;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (int64_t)gridDim.x * blockDim.x) {
    uint32_t r[4];
    philox4x32_10(seed, (uint64_t)q, offset, r);
    for (int e = 0; e < 4; ++e) {
      const int64_t i = (q << 2) + e;
      if (i < n) out[i] = ((float)(r[e] >> 8) * (1.0f / 16777216.0f) < keep) ? scale : 0.f;
    }
  }
}
__global__ void mul_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = a[i] * b[i];
}

inline int grid_for(int64_t n, int per_thread = 1) {
  int64_t blocks = (n / per_thread + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  return (int)blocks;
}

}  // namespace

extern "C" {

int hdiff_linear_rows(const float* x, const int64_t* idx, int n_rows, const float* W, const float* bias, float* y, int B,
                      int K, int N, int swish_input, int accumulate, hdiff_stream_t stream) {
  HDIFF_CHECK_ARG(x && W && y, "linear_rows: null pointer");
  HDIFF_CHECK_ARG(B > 0 && K > 0 && N > 0, "linear_rows: bad sizes");
  const int waves = B * N;
  (void)hipGetLastError();  // drop any stale error left by another HIP user in this thread
  hipLaunchKernelGGL(linear_rows_kernel, dim3(cdiv(waves, 4)), dim3(256), 0, (hipStream_t)stream, x, idx, W, bias, y, B,
                     K, N, n_rows, swish_input, accumulate);
  HDIFF_CHECK_LAUNCH("linear_rows_kernel");
  return HDIFF_OK;
}

int hdiff_q_sample(const float* x0, const float* noise, const int64_t* t, const float* sqrt_ab, const float* sqrt_1mab,
                   float* xt, int B, int per_sample, int T, hdiff_stream_t stream) {
  HDIFF_CHECK_ARG(x0 && noise && t && sqrt_ab && sqrt_1mab && xt, "q_sample: null pointer");
  HDIFF_CHECK_ARG(B > 0 && B <= 65535 && per_sample > 0 && T > 0, "q_sample: bad sizes B=%d per_sample=%d T=%d", B, per_sample, T);
  const int bx = cdiv(per_sample, 256) < 256 ? cdiv(per_sample, 256) : 256;
  (void)hipGetLastError();  // drop any stale error left by another HIP user in this thread
  hipLaunchKernelGGL(q_sample_kernel, dim3(bx, B), dim3(256), 0, (hipStream_t)stream, x0, noise, t, sqrt_ab, sqrt_1mab,
                     xt, per_sample, T);
  HDIFF_CHECK_LAUNCH("q_sample_kernel");
  return HDIFF_OK;
}

int hdiff_sq_err(const float* a, const float* b, float* out, int64_t n, hdiff_stream_t stream) {
  HDIFF_CHECK_ARG(a && b && out, "sq_err: null pointer");
  (void)hipGetLastError();  // drop any stale error left by another HIP user in this thread
  hipLaunchKernelGGL(sq_err_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, a, b, out, n);
  HDIFF_CHECK_LAUNCH("sq_err_kernel");
  return HDIFF_OK;
}

int hdiff_ddpm_step(const float* x, const float* eps_c, const float* eps_u, const float* noise, float* x_next,
                    const float* coeff1, const float* coeff2, const float* sigma, const int32_t* step_ptr, int T, double w,
                    uint64_t seed, int32_t* nan_flag, int64_t n, hdiff_stream_t stream) {
  HDIFF_CHECK_ARG(x && eps_c && eps_u && x_next && coeff1 && coeff2 && sigma && step_ptr && nan_flag,
                  "ddpm_step: null pointer");
  HDIFF_CHECK_ARG(T > 0 && n > 0, "ddpm_step: bad sizes T=%d n=%lld", T, (long long)n);
  (void)hipGetLastError();  // drop any stale error left by another HIP user in this thread
  DdpmStepK k{x, eps_c, eps_u, noise, x_next, coeff1, coeff2, sigma, const_cast<int32_t*>(step_ptr), T, (float)(1.0 + w),
              (float)w, seed, nan_flag, n, nullptr, nullptr, nullptr, 0, nullptr};
  hipLaunchKernelGGL(ddpm_step_kernel, dim3(grid_for(n, 4)), dim3(256), 0, (hipStream_t)stream, k);
  HDIFF_CHECK_LAUNCH("ddpm_step_kernel");
  return HDIFF_OK;
}

int hdiff_ddpm_step_loop(const hdiff_ddpm_loop_desc* d, hdiff_stream_t stream) {
  HDIFF_CHECK_ARG(d && d->x && d->eps_c && d->eps_u && d->x_next && d->coeff1 && d->coeff2 && d->sigma && d->step_ptr &&
                      d->nan_flag && d->done_counter,
                  "ddpm_step_loop: null pointer");
  HDIFF_CHECK_ARG(d->T > 0 && d->n > 0 && d->t_count >= 0 && (d->t_count == 0 || d->t_next),
                  "ddpm_step_loop: bad sizes T=%d n=%lld t_count=%d", d->T, (long long)d->n, d->t_count);
  (void)hipGetLastError();  // drop any stale error left by another HIP user in this thread
  DdpmStepK k{d->x, d->eps_c, d->eps_u, d->noise, d->x_next, d->coeff1, d->coeff2, d->sigma, d->step_ptr, d->T,
              (float)(1.0 + d->w), (float)d->w, d->seed, d->nan_flag, d->n, d->x_dup0, d->x_dup1, d->t_next, d->t_count,
              d->done_counter};
  hipLaunchKernelGGL(ddpm_step_kernel, dim3(grid_for(d->n, 4)), dim3(256), 0, (hipStream_t)stream, k);
  HDIFF_CHECK_LAUNCH("ddpm_step_kernel");
  return HDIFF_OK;
}

int hdiff_linear_rows_multi(const float* x0, const float* x1, const hdiff_linear_job* jobs, int njobs, int total_n, int B,
                            int K, hdiff_stream_t stream) {
  HDIFF_CHECK_ARG(x0 && jobs && njobs > 0 && total_n > 0 && B > 0 && K > 0, "linear_rows_multi: bad arguments");
  (void)hipGetLastError();  // drop any stale error left by another HIP user in this thread
  const int64_t waves = (int64_t)B * total_n;
  hipLaunchKernelGGL(linear_rows_multi_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x0, x1,
                     jobs, njobs, total_n, B, K);
  HDIFF_CHECK_LAUNCH("linear_rows_multi_kernel");
  return HDIFF_OK;
}

int hdiff_fill_t(int64_t* t, const int32_t* step_ptr, int B, hdiff_stream_t stream) {
  HDIFF_CHECK_ARG(t && step_ptr && B > 0, "fill_t: bad arguments");
  (void)hipGetLastError();  // drop any stale error left by another HIP user in this thread
  hipLaunchKernelGGL(fill_t_kernel, dim3(cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, t, step_ptr, B);
  HDIFF_CHECK_LAUNCH("fill_t_kernel");
  return HDIFF_OK;
}

int hdiff_step_decrement(int32_t* step_ptr, hdiff_stream_t stream) {
  HDIFF_CHECK_ARG(step_ptr, "step_decrement: null pointer");
  (void)hipGetLastError();  // drop any stale error left by another HIP user in this thread
  hipLaunchKernelGGL(step_decrement_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, step_ptr);
  HDIFF_CHECK_LAUNCH("step_decrement_kernel");
  return HDIFF_OK;
}

int hdiff_clip(const float* x, float* y, float lo, float hi, int64_t n, hdiff_stream_t stream) {
  HDIFF_CHECK_ARG(x && y, "clip: null pointer");
  (void)hipGetLastError();  // drop any stale error left by another HIP user in this thread
  hipLaunchKernelGGL(clip_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, y, lo, hi, n);
  HDIFF_CHECK_LAUNCH("clip_kernel");
  return HDIFF_OK;
}

int hdiff_randn(float* out, int64_t n, uint64_t seed, uint64_t offset, hdiff_stream_t stream) {
  HDIFF_CHECK_ARG(out, "randn: null pointer");
  (void)hipGetLastError();  // drop any stale error left by another HIP user in this thread
  hipLaunchKernelGGL(randn_kernel, dim3(grid_for(n, 4)), dim3(256), 0, (hipStream_t)stream, out, n, seed, offset);
  HDIFF_CHECK_LAUNCH("randn_kernel");
  return HDIFF_OK;
}

int hdiff_axpby(float a, const float* x, float b, const float* y, float* out, int64_t n, hdiff_stream_t stream) {
  HDIFF_CHECK_ARG(x && out, "axpby: null pointer");
  (void)hipGetLastError();  // drop any stale error left by another HIP user in this thread
  hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, a, x, b, y, out, n);
  HDIFF_CHECK_LAUNCH("axpby_kernel");
  return HDIFF_OK;
}

int hdiff_linear_rows_bwd(const float* x, const int64_t* idx, int n_rows, const float* W, const float* dy, float* dx,
                          float* dW, float* db, int B, int K, int N, int swish_input, int accumulate, int pad_row,
                          hdiff_stream_t stream) {
  HDIFF_CHECK_ARG(x && W && dy, "linear_rows_bwd: null pointer");
  HDIFF_CHECK_ARG(B > 0 && K > 0 && N > 0, "linear_rows_bwd: bad sizes");
  (void)hipGetLastError();  // drop any stale error left by another HIP user in this thread
  if (dW)
    hipLaunchKernelGGL(linear_rows_bwd_w_kernel, dim3(cdiv(N * K, 256)), dim3(256), 0, (hipStream_t)stream, x, idx, n_rows, dy,
                       dW, db, B, K, N, swish_input, accumulate);
  if (dx)
    hipLaunchKernelGGL(linear_rows_bwd_x_kernel, dim3(cdiv(idx ? K : B * K, 256)), dim3(256), 0, (hipStream_t)stream, x, idx,
                       n_rows, W, dy, dx, B, K, N, swish_input, pad_row);
  HDIFF_CHECK_LAUNCH("linear_rows_bwd kernels");
  return HDIFF_OK;
}

int hdiff_sq_err_bwd(const float* a, const float* b, const float* dloss, float* da, int64_t n, hdiff_stream_t stream) {
  HDIFF_CHECK_ARG(a && b && dloss && da, "sq_err_bwd: null pointer");
  (void)hipGetLastError();  // drop any stale error left by another HIP user in this thread
  hipLaunchKernelGGL(sq_err_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, a, b, dloss, da, n);
  HDIFF_CHECK_LAUNCH("sq_err_bwd_kernel");
  return HDIFF_OK;
}

int hdiff_dropout_mask(float* out, int64_t n, float keep, uint64_t seed, uint64_t offset, hdiff_stream_t stream) {
  HDIFF_CHECK_ARG(out && keep > 0.f && keep <= 1.f, "dropout_mask: bad arguments");
  (void)hipGetLastError();  // drop any stale error left by another HIP user in this thread
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for(n, 4)), dim3(256), 0, (hipStream_t)stream, out, n, keep, seed, offset);
  HDIFF_CHECK_LAUNCH("dropout_mask_kernel");
  return HDIFF_OK;
}

int hdiff_mul(const float* a, const float* b, float* out, int64_t n, hdiff_stream_t stream) {
  HDIFF_CHECK_ARG(a && b && out, "mul: null pointer");
  (void)hipGetLastError();  // drop any stale error left by another HIP user in this thread
  hipLaunchKernelGGL(mul_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, a, b, out, n);
  HDIFF_CHECK_LAUNCH("mul_kernel");
  return HDIFF_OK;
}

}  // extern "C"
