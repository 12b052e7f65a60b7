// Dev tool (round 6): is the K = 16 fp16 MFMA (v_mfma_f32_16x16x16_f16, the CDNA1-3 form) cheaper than the gfx950 K = 32 form -- in cycles
// and in board power?  The three-term score product (k1 q1 dropped) leaves ONE term of 16 contraction slots per tile; it pays only if a
// 16-slot MFMA costs less than a 32-slot one.  hipcc --offload-arch=gfx950 -O3 tools/mfma_k16_probe.hip -o tools/bin/mfma_k16_probe
//   mfma_k16_probe <mode> <seconds>     mode 0: 16x16x32_f16, 1: 16x16x16_f16, 2: alternating 32 / 16, 3: 16x16x32 with every second one skipped,
//                                       4: 32x32x16_f16 (twice the FLOPs per instruction: energy per FLOP of the two gfx950 shapes)
// Prints ms per launch and MFMAs per second; run it under tools/kernel_power.py-style sampling (tools/scripts/r6_k16.sh).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters, const float* seed) {
  // random-looking operands (data toggling matters for power): per-lane values from memory
  f16x8 a8, b8;
  f16x4 a4, b4;
  for (int i = 0; i < 8; ++i) { a8[i] = (_Float16)seed[(threadIdx.x * 8 + i) & 4095]; b8[i] = (_Float16)seed[(threadIdx.x * 8 + i + 2048) & 4095]; }
  for (int i = 0; i < 4; ++i) { a4[i] = a8[i]; b4[i] = b8[i]; }
  if (MODE == 4) {
    f32x16 big[4];
    for (int i = 0; i < 4; ++i)
      for (int r = 0; r < 16; ++r) big[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) big[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, big[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) s += big[i][0] + big[i][15];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    return;
  }
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[i], 0, 0, 0);
      else if (MODE == 1) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[i], 0, 0, 0);
      else if (MODE == 2) acc[i] = (i & 1) ? __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[i], 0, 0, 0) : __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[i], 0, 0, 0);
      else if (!(i & 1)) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[i], 0, 0, 0);
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 0;
  const double secs = argc > 2 ? atof(argv[2]) : 2.0;
  float *out, *seed;
  hipMalloc(&out, 256 * 512 * 4);
  hipMalloc(&seed, 4096 * 4);
  float h[4096];
  srand(1);
  for (int i = 0; i < 4096; ++i) h[i] = (float)rand() / RAND_MAX * 4.f - 2.f;
  hipMemcpy(seed, h, sizeof(h), hipMemcpyHostToDevice);
  const int iters = 40000;
  auto launch = [&] {
    if (mode == 0) k<0><<<256, 512>>>(out, iters, seed);
    else if (mode == 1) k<1><<<256, 512>>>(out, iters, seed);
    else if (mode == 2) k<2><<<256, 512>>>(out, iters, seed);
    else if (mode == 3) k<3><<<256, 512>>>(out, iters, seed);
    else k<4><<<256, 512>>>(out, iters, seed);
  };
  launch();
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  int n = 0;
  auto t0 = std::chrono::steady_clock::now();
  hipEventRecord(e0);
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) { launch(); hipDeviceSynchronize(); ++n; }
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double per_wave = (double)iters * ((mode == 3 || mode == 4) ? 4 : 8);
  // 2 waves per SIMD: MFMAs per SIMD per launch = 2 * per_wave
  const double flop = (mode == 1 ? 8192.0 : mode == 4 ? 32768.0 : mode == 2 ? 12288.0 : 16384.0);
  printf("mode %d: %.3f ms per launch (%d launches), %.1f ns per MFMA per SIMD, %.0f TFLOP/s chip\n", mode, ms / n, n, ms / n * 1e6 / (2 * per_wave),
         flop * 2 * per_wave * 1024 / (ms / n * 1e-3) / 1e12);
  return 0;
}
