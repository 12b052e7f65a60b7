// Dev tool: (1) is split3 exact?  (2) operand / result layout of v_mfma_f32_16x16x32_bf16.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split3(float a, float b, unsigned& h0, unsigned& h1, unsigned& h2) {
  // (-1, 0) and (0, -1) as packed bf16.  Kept opaque in SGPRs: written as an immediate, (-1, 0) is encoded as the inline
  // constant -1.0, which this instruction reads as fp32 bits, i.e. as (0, -1) (seen on gfx950 with ROCm 7.2).
  unsigned e0u = 0x0000BF80u, e1u = 0xBF800000u;
  asm volatile("" : "+s"(e0u), "+s"(e1u));
  const bf16x2 e0 = __builtin_bit_cast(bf16x2, e0u), e1 = __builtin_bit_cast(bf16x2, e1u);
  const bf16x2 p0 = __builtin_convertvector(f32x2{a, b}, bf16x2);
  const float ra = __builtin_amdgcn_fdot2_f32_bf16(p0, e0, a, false);
  const float rb = __builtin_amdgcn_fdot2_f32_bf16(p0, e1, b, false);
  const bf16x2 p1 = __builtin_convertvector(f32x2{ra, rb}, bf16x2);
  const float sa = __builtin_amdgcn_fdot2_f32_bf16(p1, e0, ra, false);
  const float sb = __builtin_amdgcn_fdot2_f32_bf16(p1, e1, rb, false);
  const bf16x2 p2 = __builtin_convertvector(f32x2{sa, sb}, bf16x2);
  h0 = __builtin_bit_cast(unsigned, p0);
  h1 = __builtin_bit_cast(unsigned, p1);
  h2 = __builtin_bit_cast(unsigned, p2);
}
__global__ void k_split(const float* in, unsigned* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned h0, h1, h2;
  split3(in[2 * i], in[2 * i + 1], h0, h1, h2);
  out[3 * i] = h0; out[3 * i + 1] = h1; out[3 * i + 2] = h2;
}
// A[m][k] = bf16 values given in global memory as float A[16][32], B[k][n] as float B[32][16]
__global__ void k_mfma(const float* A, const float* B, float* D) {
  int lane = threadIdx.x, i16 = lane & 15, g = lane >> 4;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)A[i16 * 32 + 8 * g + j]; b[j] = (__bf16)B[(8 * g + j) * 16 + i16]; }
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + i16] = c[r];
}
static float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
int main() {
  const int n = 4096;
  float* h = (float*)malloc(2 * n * 4);
  srand(1);
  for (int i = 0; i < 2 * n; ++i) h[i] = ((rand() / (float)RAND_MAX) - 0.5f) * expf((rand() % 40) - 20.f);
  float* din; unsigned* dout;
  hipMalloc(&din, 2 * n * 4); hipMalloc(&dout, 3 * n * 4);
  hipMemcpy(din, h, 2 * n * 4, hipMemcpyHostToDevice);
  k_split<<<n / 256, 256>>>(din, dout, n);
  unsigned* ho = (unsigned*)malloc(3 * n * 4);
  hipMemcpy(ho, dout, 3 * n * 4, hipMemcpyDeviceToHost);
  double worst = 0; int bad = 0;
  for (int i = 0; i < n; ++i) {
    for (int half = 0; half < 2; ++half) {
      double x = h[2 * i + half];
      double s = 0;
      for (int p = 0; p < 3; ++p) s += bf2f((unsigned short)(half ? ho[3 * i + p] >> 16 : ho[3 * i + p] & 0xffff));
      double rel = fabs(s - x) / fabs(x);
      if (rel > worst) worst = rel;
      if (rel > 1e-6 && bad < 5) { printf("bad split: x=%g sum=%g pieces %08x %08x %08x half %d\n", x, s, ho[3*i], ho[3*i+1], ho[3*i+2], half); ++bad; }
    }
  }
  printf("split3: worst relative residual %.3e (2^-24 = 5.96e-08)\n", worst);
  float hA[16 * 32], hB[32 * 16], hD[256];
  for (int i = 0; i < 512; ++i) { hA[i] = (float)((rand() % 17) - 8); hB[i] = (float)((rand() % 9) - 4); }
  float *dA, *dB, *dD;
  hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dD, 1024);
  hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice);
  k_mfma<<<1, 64>>>(dA, dB, dD);
  hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
  int wrong = 0;
  for (int m = 0; m < 16; ++m) for (int nn = 0; nn < 16; ++nn) {
    float ref = 0; for (int k = 0; k < 32; ++k) ref += hA[m * 32 + k] * hB[k * 16 + nn];
    if (ref != hD[m * 16 + nn]) ++wrong;
  }
  printf("mfma 16x16x32 bf16 layout: %d of 256 entries wrong\n", wrong);
  return 0;
}
