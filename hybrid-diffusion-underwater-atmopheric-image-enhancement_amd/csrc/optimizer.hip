// The tail of an optimizer step as three launches over a LIST of tensors: the global gradient norm, the clip coefficient and one fused
// clip + AdamW update (reference: TrainCondition.py:61-63 -- torch.nn.utils.clip_grad_norm_(params, grad_clip); optimizer.step() with
// torch.optim.AdamW(lr, weight_decay = 1e-4), :39-40).  Through round 5 this tail was torch-native (foreach kernels: ~25 launches over 366
// tensors); SURVEY section 7.6 allowed that "at first".  HBM-bound elementwise work: p, g, m, v read, p, g, m, v written once -- 8 x 4 bytes
// per parameter (1.5 GB for the default model).
//
// Tensors are described by a device table {p, g, m, v, n}; the work is cut into chunks of CHUNK elements, chunk c -> (tensor, offset) by a
// second device table, both built once by the host (hdiff_amd/optim.py) and reused while the pointers stay the same.  Pointers need only
// 4-byte alignment (gradient views into a flat exchange buffer start anywhere).
// Deterministic: every partial sum is formed in a fixed order (no float atomics), the final sum in float64.
#include "common.h"

using namespace hdiff;

namespace {

constexpr int CHUNK = 4096;       // elements per workgroup
constexpr int OT = 256;

struct OptTensor {
  float* p;
  float* g;
  float* m;
  float* v;
  long long n;
};
static_assert(sizeof(OptTensor) == sizeof(hdiff_opt_tensor), "host and device table layouts");

__global__ __launch_bounds__(OT) void opt_sq_norm_kernel(const OptTensor* __restrict__ tab, const int2* __restrict__ chunks,
                                                         float* __restrict__ partial) {
  const int2 c = chunks[blockIdx.x];
  const OptTensor t = tab[c.x];
  const long long base = (long long)c.y * CHUNK;
  float s = 0.f;
#pragma unroll 4
  for (int i = threadIdx.x; i < CHUNK; i += OT) {
    const long long e = base + i;
    if (e < t.n) { const float gv = t.g[e]; s = __builtin_fmaf(gv, gv, s); }
  }
  s = wave_sum(s);
  __shared__ float red[OT / 64];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

// out[0] = total norm, out[1] = clip coefficient min(1, max_norm / (norm + 1e-6)) (torch.nn.utils.clip_grad_norm_'s), one workgroup
__global__ __launch_bounds__(OT) void opt_clip_coef_kernel(const float* __restrict__ partial, int n, float max_norm, float* __restrict__ out) {
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += OT) s += (double)partial[i];
  __shared__ double red[OT];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = OT / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float norm = (float)sqrt(red[0]);
    out[0] = norm;
    const float c = max_norm / (norm + 1e-6f);
    out[1] = c < 1.0f ? c : 1.0f;
  }
}

struct AdamArgs {
  float decay, one_minus_beta1, beta2, one_minus_beta2, eps, step_size, bias2_sqrt;      // the scalars torch forms in Python doubles: 1 - lr wd, lr / (1 - beta1^step), sqrt(1 - beta2^step)
};

// torch.optim.AdamW's single-tensor formulas, in its order of operations:
//   p *= 1 - lr wd;  m += (g - m)(1 - beta1);  v = beta2 v + (1 - beta2) g g;  p -= (lr / bias1) m / (sqrt(v) / bias2_sqrt + eps)
// with g the CLIPPED gradient g coef, written back (clip_grad_norm_ scales the gradients in place)
__global__ __launch_bounds__(OT) void opt_adamw_kernel(const OptTensor* __restrict__ tab, const int2* __restrict__ chunks,
                                                       const float* __restrict__ coef, const AdamArgs a) {
  const int2 c = chunks[blockIdx.x];
  const OptTensor t = tab[c.x];
  const long long base = (long long)c.y * CHUNK;
  const float k = coef ? coef[1] : 1.0f;
#pragma unroll 4
  for (int i = threadIdx.x; i < CHUNK; i += OT) {
    const long long e = base + i;
    if (e < t.n) {
      float g = t.g[e];
      if (coef) { g *= k; t.g[e] = g; }
      float p = t.p[e] * a.decay;
      float m = t.m[e];
      m = m + (g - m) * a.one_minus_beta1;
      const float v = t.v[e] * a.beta2 + a.one_minus_beta2 * g * g;
      const float denom = sqrtf(v) / a.bias2_sqrt + a.eps;
      p = p - a.step_size * (m / denom);
      t.p[e] = p; t.m[e] = m; t.v[e] = v;
    }
  }
}

}  // namespace

extern "C" int hdiff_opt_chunk(void) { return CHUNK; }

extern "C" int hdiff_grad_norm_clip_coef(const hdiff_opt_tensor* table, const int* chunks, int nchunks, float* partial, float max_norm,
                                         float* norm_coef, hdiff_stream_t stream) {
  HDIFF_CHECK_ARG(table && chunks && partial && norm_coef, "grad_norm_clip_coef: null pointer");
  HDIFF_CHECK_ARG(nchunks > 0 && max_norm > 0.f, "grad_norm_clip_coef: bad sizes (nchunks %d, max_norm %g)", nchunks, (double)max_norm);
  (void)hipGetLastError();
  hipLaunchKernelGGL(opt_sq_norm_kernel, dim3(nchunks), dim3(OT), 0, (hipStream_t)stream, reinterpret_cast<const OptTensor*>(table),
                     reinterpret_cast<const int2*>(chunks), partial);
  hipLaunchKernelGGL(opt_clip_coef_kernel, dim3(1), dim3(OT), 0, (hipStream_t)stream, partial, nchunks, max_norm, norm_coef);
  HDIFF_CHECK_LAUNCH("grad_norm_clip_coef kernels");
  return HDIFF_OK;
}

extern "C" int hdiff_adamw_step(const hdiff_opt_tensor* table, const int* chunks, int nchunks, const float* norm_coef, double lr, double beta1,
                                double beta2, double eps, double weight_decay, int64_t step, hdiff_stream_t stream) {
  HDIFF_CHECK_ARG(table && chunks, "adamw_step: null pointer");
  HDIFF_CHECK_ARG(nchunks > 0 && step >= 1, "adamw_step: bad sizes (nchunks %d, step %lld)", nchunks, (long long)step);
  HDIFF_CHECK_ARG(beta1 >= 0. && beta1 < 1. && beta2 >= 0. && beta2 < 1. && eps >= 0., "adamw_step: bad hyper-parameters");
  AdamArgs a;
  a.decay = (float)(1.0 - lr * weight_decay);
  a.one_minus_beta1 = (float)(1.0 - beta1); a.beta2 = (float)beta2; a.one_minus_beta2 = (float)(1.0 - beta2); a.eps = (float)eps;
  a.step_size = (float)(lr / (1.0 - pow(beta1, (double)step)));
  a.bias2_sqrt = (float)sqrt(1.0 - pow(beta2, (double)step));
  (void)hipGetLastError();
  hipLaunchKernelGGL(opt_adamw_kernel, dim3(nchunks), dim3(OT), 0, (hipStream_t)stream, reinterpret_cast<const OptTensor*>(table),
                     reinterpret_cast<const int2*>(chunks), norm_coef, a);
  HDIFF_CHECK_LAUNCH("adamw kernel");
  return HDIFF_OK;
}
