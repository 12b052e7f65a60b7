// Dev tool: do bf16 MFMA and ordinary VALU instructions overlap on one SIMD, or do they share the FMA datapath?
// Loop body: 2 independent MFMA 16x16x32 bf16 + NV independent v_fma_f32 (or v_exp_f32).  Compare times for NV = 0, 4, 8, 16.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NM, int NV, int TRANS>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  float a = seed + threadIdx.x * 1e-3f, b = seed * 0.5f + threadIdx.x * 2e-3f;
  bf16x8 a8, b8;
  for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(a + i); b8[i] = (__bf16)(b - i); }
  f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  float x[16];
  for (int i = 0; i < 16; ++i) x[i] = seed * (i + 1);
  for (int it = 0; it < iters; ++it) {
    if (NM > 0) acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NV / 2; ++i) {
      if (TRANS) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
      else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
    }
    if (NM > 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc1, 0, 0, 0);
#pragma unroll
    for (int i = NV / 2; i < NV; ++i) {
      if (TRANS) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
      else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
    }
  }
  float s = acc0[0] + acc1[3];
  for (int i = 0; i < 16; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NM, int NV, int TRANS>
void run(int blocks_per_cu) {
  float* out;
  int blocks = 256 * blocks_per_cu;
  (void)hipMalloc(&out, blocks * 256 * 4);
  int iters = 40000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<NM, NV, TRANS><<<blocks, 256>>>(out, 100, 0.3f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<NM, NV, TRANS><<<blocks, 256>>>(out, iters, 0.3f);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  double cyc_per_iter = ms * 1e-3 * 2.4e9 / iters;
  printf("MFMA/iter=%d %s/iter=%2d waves/SIMD=%d: %.3f ms  -> %.1f cycles/iter (at 2.4 GHz)\n", NM, TRANS ? "v_exp" : "v_fma", NV,
         blocks_per_cu, ms, cyc_per_iter);
  (void)hipFree(out);
}

int main() {
  for (int w = 1; w <= 2; ++w) {
    if (w == 1) { run<2, 0, 0>(1); run<0, 8, 0>(1); run<2, 4, 0>(1); run<2, 8, 0>(1); run<2, 16, 0>(1); run<0, 8, 1>(1); run<2, 4, 1>(1); run<2, 8, 1>(1); }
    else { run<2, 0, 0>(2); run<0, 8, 0>(2); run<2, 4, 0>(2); run<2, 8, 0>(2); run<2, 16, 0>(2); run<0, 8, 1>(2); run<2, 4, 1>(2); run<2, 8, 1>(2); }
  }
  return 0;
}
