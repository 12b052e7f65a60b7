// Dev tool: how well do independent bf16 MFMAs (16x16x32) and VALU instructions overlap on one SIMD when interleaved
// in program order?  Loop body: 8 MFMAs on 8 independent accumulators, each followed by NV VALU ops of kind KIND
// (0 v_fma_f32, 1 v_exp_f32, 2 v_cvt_pk_bf16_f32, 3 v_dot2c_f32_bf16, 4 v_pk_add_f32).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NM, int NV, int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  float a = seed + threadIdx.x * 1e-3f, b = seed * 0.5f + threadIdx.x * 2e-3f;
  bf16x8 a8, b8;
  for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(a + i); b8[i] = (__bf16)(b - i); }
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float x[8];
  f32x2 y[8];
  unsigned u[8];
  for (int i = 0; i < 8; ++i) { x[i] = seed * (i + 1); y[i] = f32x2{seed, seed * i}; u[i] = i; }
  unsigned e0 = 0x0000BF80u;
  asm volatile("" : "+s"(e0));
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      if (m < NM) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(a8), "v"(b8));
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int r = (m * NV + i) & 7;
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[r]) : "v"(a), "v"(b));
        if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x[r]));
        if (KIND == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[r]) : "v"(x[r]), "v"(a));
        if (KIND == 3) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(x[r]) : "s"(e0), "v"(u[r]));
        if (KIND == 4) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(y[r]) : "v"(y[(r + 1) & 7]));
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + x[i] + y[i][0] + y[i][1] + (float)u[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NM, int NV, int KIND>
void run(int blocks_per_cu) {
  float* out;
  int blocks = 256 * blocks_per_cu;
  (void)hipMalloc(&out, blocks * 256 * 4);
  int iters = 20000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<NM, NV, KIND><<<blocks, 256>>>(out, 100, 0.3f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<NM, NV, KIND><<<blocks, 256>>>(out, iters, 0.3f);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const char* names[] = {"v_fma", "v_exp", "v_cvt_pk_bf16", "v_dot2c_bf16", "v_pk_add"};
  double cyc = ms * 1e-3 * 2.4e9 / iters;
  printf("waves/SIMD=%d  %d MFMA + %2d %-14s per iter: %7.1f cycles/iter/wave-slot  (x waves = %7.1f SIMD cycles)\n", blocks_per_cu, NM,
         8 * NV, names[KIND], cyc, cyc);
  (void)hipFree(out);
}

template <int KIND>
void sweep(int w) {
  run<0, 4, KIND>(w);
  run<8, 0, KIND>(w);
  run<8, 2, KIND>(w);
  run<8, 4, KIND>(w);
  run<8, 6, KIND>(w);
}

int main() {
  for (int w = 1; w <= 3; ++w) {
    sweep<0>(w); sweep<1>(w); sweep<2>(w); sweep<3>(w); sweep<4>(w);
  }
  return 0;
}
