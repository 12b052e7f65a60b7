// Dev tool: sustained fp32-MFMA rate of this chip under load (registers only), to calibrate what fraction of the
// 157.3 TFLOP/s datasheet peak is reachable at the clock the chip holds.  hipcc --offload-arch=gfx950 -O3 mfma_peak.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE, int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  float a = seed + threadIdx.x * 1e-3f, b = seed * 0.5f + threadIdx.x * 2e-3f;
  if (SHAPE == 16) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  } else {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
      for (int r = 0; r < 16; ++r) acc[i][r] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][15];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  }
}

template <int SHAPE, int NACC>
void run(const char* name, int blocks_per_cu) {
  float* out;
  int blocks = 256 * blocks_per_cu;
  hipMalloc(&out, blocks * 256 * 4);
  int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<SHAPE, NACC><<<blocks, 256>>>(out, 100, 0.3f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<SHAPE, NACC><<<blocks, 256>>>(out, iters, 0.3f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double flop = (double)blocks * 4 * iters * NACC * (SHAPE == 16 ? 2048.0 : 4096.0);
  printf("%s blocks/CU=%d NACC=%d: %.2f ms  %.1f TFLOP/s\n", name, blocks_per_cu, NACC, ms, flop / ms / 1e9);
  hipFree(out);
}

int main() {
  run<16, 4>("mfma_f32_16x16x4", 1);
  run<16, 4>("mfma_f32_16x16x4", 2);
  run<16, 8>("mfma_f32_16x16x4", 1);
  run<16, 2>("mfma_f32_16x16x4", 2);
  run<16, 1>("mfma_f32_16x16x4", 2);
  run<32, 4>("mfma_f32_32x32x2", 1);
  run<32, 2>("mfma_f32_32x32x2", 2);
  run<32, 1>("mfma_f32_32x32x2", 1);
  return 0;
}
