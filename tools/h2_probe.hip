// Dev tool (round 4): everything the fp16-piece form of P needs to know about gfx950 before a kernel is built on it.
//   hipcc --offload-arch=gfx950 -O3 tools/h2_probe.hip -o h2_probe && ./h2_probe
// Part A (function): v_cvt_pk_f16_f32 rounds to nearest-even and produces fp16 subnormals; v_fma_mix_f32 reads them;
//                    v_mfma_f32_16x16x32_f16 multiplies subnormal inputs instead of flushing them; inf inputs poison.
// Part B (rates):    issue cost of v_cvt_pk_f16_f32 / v_cvt_pkrtz_f16_f32 / v_fma_mix_f32, alone and beside one 16x16x32 f16 MFMA
//                    (as tools/valu_rates.hip does for the bf16 forms).
// Part C (mix):      the per-tile instruction mixes, old and candidate, beside their MFMAs:
//                      old  : 24 MFMA + 16 exp + 16 add + 88 (v_and, v_sub, v_perm 4:4:3)       (bf16 x 3 pieces of P)
//                      h2   : 18 MFMA + 16 exp + 16 add + 16 v_cvt_pk + 16 v_fma_mix              (fp16 x 2 pieces of P, 3 P.V terms)
//                      h2q4 : 14 MFMA + the same vector stream                                    (and 4 QK^T terms)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <string.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// ------------------------------------------------------------------------------------------------ part A
__global__ void func_kernel(const float* in, float* out, int n) {
  const int i = threadIdx.x;
  if (i >= n) return;
  const float a = in[2 * i], b = in[2 * i + 1];
  unsigned p0, p1;
  asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(p0) : "v"(a), "v"(b));
  float ra, rb;
  asm volatile("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(ra) : "v"(a), "v"(p0));
  asm volatile("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(rb) : "v"(b), "v"(p0));
  asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(p1) : "v"(ra), "v"(rb));
  out[6 * i + 0] = __builtin_bit_cast(float, p0);
  out[6 * i + 1] = __builtin_bit_cast(float, p1);
  out[6 * i + 2] = ra;
  out[6 * i + 3] = rb;
  // back to fp32 through the mix instruction: a0 + a1
  float sa, sb;
  asm volatile("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(sa) : "v"(p0), "v"(p1));
  asm volatile("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(sb) : "v"(p0), "v"(p1));
  out[6 * i + 4] = sa;
  out[6 * i + 5] = sb;
}

// one wave: D = A(16x32) * B(32x16) with A[m][k] = av (all m, k), B[k][n] = bv: every D element = 32 * av * bv
__global__ void mfma_denorm_kernel(float* out, float av, float bv) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)av; b[i] = (_Float16)bv; }
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = c[0];
}

// ------------------------------------------------------------------------------------------------ part B
template <int KIND>
__device__ __forceinline__ void one(float& x, float& y, unsigned& u) {
  if (KIND == 0) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u) : "v"(x), "v"(y));
  if (KIND == 1) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(u) : "v"(x), "v"(y));
  if (KIND == 2) asm volatile("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(x) : "v"(y), "v"(u));
  if (KIND == 3) asm volatile("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(x) : "v"(y), "v"(u));
  if (KIND == 4) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u) : "v"(x), "v"(y));
  if (KIND == 5) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(y));
}

template <int KIND, int NV, int MF>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters, float seed) {
  float x[8], y[8];
  unsigned u[8];
  for (int i = 0; i < 8; ++i) { x[i] = seed * (i + 1); y[i] = 1.0f + seed * 1e-6f * i; u[i] = 0x3c003c00u + i; }
  f16x8 a8, b8;
  for (int i = 0; i < 8; ++i) { a8[i] = (_Float16)(seed + i); b8[i] = (_Float16)(seed - i); }
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      if (MF) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(a8), "v"(b8));
#pragma unroll
      for (int i = 0; i < NV; ++i) one<KIND>(x[(m * NV + i) & 7], y[(m * NV + i) & 7], u[(m * NV + i) & 7]);
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += x[i] + y[i] + (float)u[i] + acc[i][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
double time_cycles(F launch, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  launch(100);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  launch(iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e-3 * 2.4e9 / iters;
}

template <int KIND, int NV, int MF>
double run_rate(int w, float* out) {
  return time_cycles([&](int it) { rate_kernel<KIND, NV, MF><<<256 * w, 256>>>(out, it, 0.3f); }, 20000) / 8 / w;
}

template <int KIND>
void kind(const char* name, float* out) {
  for (int w = 1; w <= 2; ++w) {
    const double alone = run_rate<KIND, 8, 0>(w, out) / 8;
    const double m0 = run_rate<KIND, 0, 1>(w, out), m2 = run_rate<KIND, 2, 1>(w, out), m4 = run_rate<KIND, 4, 1>(w, out),
                 m8 = run_rate<KIND, 8, 1>(w, out);
    printf("waves/SIMD=%d  %-22s alone %5.2f cycles/instr | beside one 16x16x32 f16 MFMA (alone %5.1f): +2 -> %5.1f  +4 -> %5.1f  +8 -> %5.1f\n",
           w, name, alone, m0, m2, m4, m8);
  }
}

// ------------------------------------------------------------------------------------------------ part C
// One "tile" = NM MFMAs with the vector stream of 16 scores per lane spread evenly between them.
//  MIX 0: old (exp, add, and/sub/perm)   MIX 1: h2 (exp, add, cvt_pk_f16, 2 fma_mix, cvt_pk_f16)   MIX 2: no vector work
template <int MIX, int NM>
__global__ __launch_bounds__(256, 2) void mix_kernel(float* out, int iters, float seed) {
  float x[16];
  unsigned u[16];
  float sum = 0.f;
  for (int i = 0; i < 16; ++i) { x[i] = seed * (i + 1) * 0.01f; u[i] = i; }
  f16x8 a8, b8;
  for (int i = 0; i < 8; ++i) { a8[i] = (_Float16)(seed + i); b8[i] = (_Float16)(seed - i); }
  f32x4 acc[6];
  for (int i = 0; i < 6; ++i) acc[i] = f32x4{0, 0, 0, 0};
  unsigned msk = 0xffff0000u, sel = 0x07060302u;
  asm volatile("" : "+s"(msk), "+s"(sel));
  constexpr int NVEC = (MIX == 0) ? 120 : (MIX == 1) ? 64 : 0;
  for (int it = 0; it < iters; ++it) {
    int v = 0;          // vector instructions issued so far in this tile
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[m % 6]) : "v"(a8), "v"(b8));
      const int upto = NVEC * (m + 1) / NM;
#pragma unroll
      for (; v < upto; ++v) {
        if (MIX == 0) {
          // per pair of scores (15 instructions): 2 exp, 2 add, then split3: perm, 2 and, 2 sub, perm, 2 and, 2 sub, perm
          const int pr = (v / 15) & 7, j = v % 15;
          float& a = x[2 * pr];
          float& b = x[2 * pr + 1];
          unsigned& ua = u[2 * pr];
          unsigned& ub = u[2 * pr + 1];
          if (j == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(a));
          if (j == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(b));
          if (j == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(sum) : "v"(a));
          if (j == 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(sum) : "v"(b));
          if (j == 4 || j == 9 || j == 14) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(ua) : "v"(a), "v"(b), "s"(sel));
          if (j == 5 || j == 10) asm volatile("v_and_b32 %0, %1, %2" : "=v"(ua) : "s"(msk), "v"(a));
          if (j == 6 || j == 11) asm volatile("v_and_b32 %0, %1, %2" : "=v"(ub) : "s"(msk), "v"(b));
          if (j == 7 || j == 12) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a) : "v"(ua));
          if (j == 8 || j == 13) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(b) : "v"(ub));
        }
        if (MIX == 1) {
          // per pair of scores (8 instructions): 2 exp, 2 add, cvt_pk, 2 fma_mix, cvt_pk
          const int pr = (v / 8) & 7, j = v % 8;
          float& a = x[2 * pr];
          float& b = x[2 * pr + 1];
          unsigned& ua = u[2 * pr];
          unsigned& ub = u[2 * pr + 1];
          if (j == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(a));
          if (j == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(b));
          if (j == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(sum) : "v"(a));
          if (j == 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(sum) : "v"(b));
          if (j == 4) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(ua) : "v"(a), "v"(b));
          if (j == 5) asm volatile("v_fma_mix_f32 %0, %0, 1.0, -%1 op_sel_hi:[0,0,1]" : "+v"(a) : "v"(ua));
          if (j == 6) asm volatile("v_fma_mix_f32 %0, %0, 1.0, -%1 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(b) : "v"(ua));
          if (j == 7) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(ub) : "v"(a), "v"(b));
        }
      }
    }
  }
  float s = sum;
  for (int i = 0; i < 16; ++i) s += x[i] + (float)u[i];
  for (int i = 0; i < 6; ++i) s += acc[i][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MIX, int NM>
void run_mix(const char* name, int w, float* out) {
  const double cyc = time_cycles([&](int it) { mix_kernel<MIX, NM><<<256 * w, 256>>>(out, it, 0.3f); }, 4000) / w;
  printf("waves/SIMD=%d  %-34s %7.1f SIMD cycles per 16x64 score tile of one wave (MFMA issue alone would be %d)\n", w, name, cyc,
         NM * 16);
}

int main() {
  float* out;
  (void)hipMalloc(&out, 256 * 4 * 256 * 4);
  // ---- A
  {
    const int n = 12;
    float h[2 * n] = {1.0f, 1.0f + 1.0f / 2048,        // tie at the 11-bit boundary: RTN-even keeps 1.0, RTZ too
                      1.0f + 3.0f / 2048, 0.7853982f,   // tie, odd -> RTN-even rounds UP (RTZ would not)
                      3.0e-6f, 5.0e-8f,                 // fp16 subnormal range (min normal 6.1e-5, min subnormal 5.96e-8)
                      1.0e-7f, 2.9e-8f,                 // near the bottom of the subnormal range
                      65504.f, 65519.f, 65520.f, 1e6f,  // overflow edge: 65520 rounds to inf
                      0.33333334f, 123.456f, 1e-3f, 2.5e-4f, 6.1e-5f, 6.0e-5f, 0.f, -0.75f, 1e-5f, 7e-6f, 255.99f, 256.01f};
    float *din, *dout;
    (void)hipMalloc(&din, sizeof(h));
    (void)hipMalloc(&dout, 6 * n * 4);
    (void)hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
    func_kernel<<<1, 64>>>(din, dout, n);
    float o[6 * n];
    (void)hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    printf("A: x -> p0 = cvt_pk_f16(x) [raw halves], r = x - p0 (fma_mix), p1 = cvt_pk_f16(r), back = p0 + p1 (fma_mix); err = x - back\n");
    for (int i = 0; i < n; ++i) {
      unsigned p0, p1;
      memcpy(&p0, &o[6 * i], 4);
      memcpy(&p1, &o[6 * i + 1], 4);
      printf("   a=%-14.9g p0.lo=0x%04x p1.lo=0x%04x r=%-13.6g back=%-14.9g err=%-11.3g | b=%-14.9g p0.hi=0x%04x p1.hi=0x%04x r=%-13.6g back=%-14.9g err=%.3g\n",
             h[2 * i], p0 & 0xffff, p1 & 0xffff, o[6 * i + 2], o[6 * i + 4], h[2 * i] - o[6 * i + 4], h[2 * i + 1], p0 >> 16,
             p1 >> 16, o[6 * i + 3], o[6 * i + 5], h[2 * i + 1] - o[6 * i + 5]);
    }
    float* d1;
    (void)hipMalloc(&d1, 4);
    const float pairs[][2] = {{1e-6f, 1.0f}, {1e-6f, 1e-6f}, {6e-8f, 1024.f}, {65504.f, 1.0f}, {INFINITY, 1.0f}, {INFINITY, 0.f}};
    for (auto& p : pairs) {
      mfma_denorm_kernel<<<1, 64>>>(d1, p[0], p[1]);
      float r;
      (void)hipMemcpy(&r, d1, 4, hipMemcpyDeviceToHost);
      const double expect = 32.0 * (double)(float)(_Float16)p[0] * (double)(float)(_Float16)p[1];
      printf("A: mfma_f32_16x16x32_f16, all A = %g, all B = %g: D = %.9g (32 a b with fp16-rounded inputs = %.9g)\n", p[0], p[1], r, expect);
    }
  }
  // ---- B
  kind<0>("v_cvt_pk_f16_f32", out);
  kind<1>("v_cvt_pkrtz_f16_f32", out);
  kind<2>("v_fma_mix_f32 (lo)", out);
  kind<3>("v_fma_mix_f32 (hi)", out);
  kind<4>("v_cvt_pk_bf16_f32", out);
  kind<5>("v_add_f32", out);
  // ---- C
  for (int w = 1; w <= 2; ++w) {
    run_mix<2, 24>("24 MFMA, no vector work", w, out);
    run_mix<2, 18>("18 MFMA, no vector work", w, out);
    run_mix<0, 24>("old: 24 MFMA + 120 vector", w, out);
    run_mix<1, 24>("24 MFMA + 64 vector (h2 stream)", w, out);
    run_mix<1, 18>("h2: 18 MFMA + 64 vector", w, out);
    run_mix<1, 14>("h2q4: 14 MFMA + 64 vector", w, out);
    run_mix<1, 12>("12 MFMA + 64 vector", w, out);
  }
  return 0;
}
