// Dev tool: bf16 MFMA shape vs VALU co-issue.  The same VALU work (NV plain instructions per 32 MFMA-pipe cycles) beside
// v_mfma_f32_32x32x16_bf16 (one per 32 cycles) and beside v_mfma_f32_16x16x32_bf16 (two per 32 cycles): does the larger
// shape, which holds the SIMD's vector issue for a smaller share of its time, hide more of it?
// Loop body: 8 x { MFMA(s) worth 32 pipe cycles ; NV VALU of kind KIND }.  KIND 0 v_fma_f32, 1 v_exp_f32,
// 2 the split mix (v_and, v_sub, v_perm in ratio 2:2:1), 3 split mix + exp (4 and/sub, 1.5 perm, 1 exp per 6.5).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int SHAPE, int NV, int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  float a = seed + threadIdx.x * 1e-3f, b = seed * 0.5f + threadIdx.x * 2e-3f;
  bf16x8 a8, b8;
  for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(a + i); b8[i] = (__bf16)(b - i); }
  f32x16 big[4];
  f32x4 small[8];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) big[i][j] = 0.f;
  for (int i = 0; i < 8; ++i) small[i] = f32x4{0, 0, 0, 0};
  float x[8];
  unsigned u[8];
  for (int i = 0; i < 8; ++i) { x[i] = seed * (i + 1); u[i] = i; }
  unsigned msk = 0xffff0000u, sel = 0x07060302u;
  asm volatile("" : "+s"(msk), "+s"(sel));
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      if (SHAPE == 32) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(big[m & 3]) : "v"(a8), "v"(b8));
      if (SHAPE == 16) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(small[m]) : "v"(a8), "v"(b8));
      }
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int r = (m * NV + i) & 7;
        if (SHAPE == 16 && i == NV / 2) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(small[(m + 4) & 7]) : "v"(a8), "v"(b8));
        int kind = KIND;
        if (KIND == 2) kind = (i % 5 == 4) ? 12 : ((i % 5) & 1 ? 11 : 10);
        if (KIND == 3) { const int j = i % 13; kind = (j == 12 || j == 5) ? 1 : (j == 4 || j == 9 || j == 11) ? 12 : ((j & 1) ? 11 : 10); }
        if (kind == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[r]) : "v"(a), "v"(b));
        if (kind == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x[r]));
        if (kind == 10) asm volatile("v_and_b32 %0, %1, %2" : "=v"(u[r]) : "s"(msk), "v"(x[r]));
        if (kind == 11) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x[r]) : "v"(u[(r + 7) & 7]));
        if (kind == 12) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(u[r]) : "v"(x[r]), "v"(x[(r + 1) & 7]), "s"(sel));
      }
      if (SHAPE == 16 && NV == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(small[(m + 4) & 7]) : "v"(a8), "v"(b8));
    }
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) s += big[i][0] + big[i][7];
  for (int i = 0; i < 8; ++i) s += small[i][0] + x[i] + (float)u[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int SHAPE, int NV, int KIND>
void run(int blocks_per_cu) {
  float* out;
  int blocks = 256 * blocks_per_cu;
  (void)hipMalloc(&out, blocks * 256 * 4);
  int iters = 20000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<SHAPE, NV, KIND><<<blocks, 256>>>(out, 100, 0.3f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<SHAPE, NV, KIND><<<blocks, 256>>>(out, iters, 0.3f);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const char* names[] = {"v_fma", "v_exp", "split-mix", "split+exp"};
  double cyc = ms * 1e-3 * 2.4e9 / iters / 8 / blocks_per_cu;       // SIMD cycles (at 2.4 GHz) per 32 MFMA-pipe cycles of ONE wave
  printf("waves/SIMD=%d  shape %2d  %2d %-10s per 32 MFMA cycles: %6.1f SIMD cycles per wave-unit (MFMA alone = 32)\n",
         blocks_per_cu, SHAPE, NV, names[KIND], cyc);
  (void)hipFree(out);
}

template <int KIND>
void sweep(int w) {
  run<32, 0, KIND>(w); run<16, 0, KIND>(w);
  run<32, 4, KIND>(w); run<16, 4, KIND>(w);
  run<32, 6, KIND>(w); run<16, 6, KIND>(w);
  run<32, 9, KIND>(w); run<16, 9, KIND>(w);
  run<32, 13, KIND>(w); run<16, 13, KIND>(w);
}

template <int NV>
__global__ __launch_bounds__(256) void valu_only(float* out, int iters, float seed) {      // the split+exp mix alone
  float x[8];
  unsigned u[8];
  for (int i = 0; i < 8; ++i) { x[i] = seed * (i + 1) + threadIdx.x; u[i] = i; }
  unsigned msk = 0xffff0000u, sel = 0x07060302u;
  asm volatile("" : "+s"(msk), "+s"(sel));
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int r = (m * NV + i) & 7;
        const int j = i % 13;
        const int kind = (j == 12 || j == 5) ? 1 : (j == 4 || j == 9 || j == 11) ? 12 : ((j & 1) ? 11 : 10);
        if (kind == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x[r]));
        if (kind == 10) asm volatile("v_and_b32 %0, %1, %2" : "=v"(u[r]) : "s"(msk), "v"(x[r]));
        if (kind == 11) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x[r]) : "v"(u[(r + 7) & 7]));
        if (kind == 12) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(u[r]) : "v"(x[r]), "v"(x[(r + 1) & 7]), "s"(sel));
      }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += x[i] + (float)u[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

void run_valu(int w) {
  float* out;
  int blocks = 256 * w;
  (void)hipMalloc(&out, blocks * 256 * 4);
  int iters = 20000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  valu_only<13><<<blocks, 256>>>(out, 100, 0.3f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  valu_only<13><<<blocks, 256>>>(out, iters, 0.3f);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("waves/SIMD=%d  VALU only, 13 split+exp per unit: %6.1f SIMD cycles per wave-unit = %.2f per instruction\n", w,
         ms * 1e-3 * 2.4e9 / iters / 8 / w, ms * 1e-3 * 2.4e9 / iters / 8 / w / 13);
  (void)hipFree(out);
}

int main(int argc, char** argv) {
  if (argc > 1) {        // more waves per SIMD: does the VALU side get cheaper?
    for (int w = 1; w <= 4; ++w) {
      run_valu(w);
      run<32, 0, 3>(w); run<32, 9, 3>(w); run<16, 9, 3>(w); run<32, 13, 3>(w); run<16, 13, 3>(w);
    }
    return 0;
  }
  for (int w = 1; w <= 2; ++w) { sweep<0>(w); sweep<2>(w); sweep<3>(w); }
  return 0;
}
