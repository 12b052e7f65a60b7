// Dev tool: issue cost of the vector instructions the split-bf16 kernels are made of, one kind at a time and beside the
// bf16 MFMA -- in particular whether the PACKED fp32 forms (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32: two fp32 results
// per lane and instruction) cost one issue slot or two, and what v_exp_f32 costs against a plain instruction.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rates.hip -o valu_rates && ./valu_rates
// KIND 0 v_sub_f32   1 v_pk_add_f32 (neg on src1 = packed subtract)   2 v_and_b32   3 v_perm_b32   4 v_exp_f32
//      5 v_pk_mul_f32   6 v_pk_fma_f32   7 v_fma_f32   8 v_cvt_pk_bf16_f32   9 v_max3_f32
//      10 v_dot2_f32_bf16 (VOP3P)   11 v_dot2c_f32_bf16 (VOP2, accumulates into its destination)   12 v_pack_b32_f16 with
//      op_sel (the high halves of two registers into one)   13 v_and_b32 with the mask in a VGPR   14 v_add_f32
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>
__device__ __forceinline__ void one(f32x2& x, f32x2& y, unsigned& u, unsigned msk, unsigned sel) {
  if (KIND == 0) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x.x) : "v"(y.x));
  if (KIND == 1) asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(x) : "v"(y));
  if (KIND == 2) asm volatile("v_and_b32 %0, %1, %2" : "=v"(u) : "s"(msk), "v"(x.x));
  if (KIND == 3) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(u) : "v"(x.x), "v"(x.y), "s"(sel));
  if (KIND == 4) asm volatile("v_exp_f32 %0, %0" : "+v"(x.x));
  if (KIND == 5) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(y));
  if (KIND == 6) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y));
  if (KIND == 7) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x.x) : "v"(y.x));
  if (KIND == 8) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u) : "v"(x.x), "v"(x.y));
  if (KIND == 9) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x.x) : "v"(y.x), "v"(y.y));
  if (KIND == 10) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(x.x) : "v"(u), "s"(sel));
  if (KIND == 11) asm volatile("v_dot2c_f32_bf16 %0, %2, %1" : "+v"(x.x) : "v"(u), "s"(sel));
  if (KIND == 12) asm volatile("v_pack_b32_f16 %0, %1, %2 op_sel:[1,1,0]" : "=v"(u) : "v"(x.x), "v"(x.y));
  if (KIND == 13) asm volatile("v_and_b32 %0, %1, %2" : "=v"(u) : "v"(y.x), "v"(x.x));
  if (KIND == 14) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x.x) : "v"(y.x));
}

// NV instructions of kind KIND per MFMA (MF = 1) or alone (MF = 0); 8 independent register sets
template <int KIND, int NV, int MF>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  f32x2 x[8], y[8];
  unsigned u[8];
  for (int i = 0; i < 8; ++i) { x[i] = f32x2{seed * (i + 1), seed + i}; y[i] = f32x2{1.0f + seed * 1e-6f * i, 1.0f}; u[i] = i; }
  bf16x8 a8, b8;
  for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(seed + i); b8[i] = (__bf16)(seed - i); }
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
  unsigned msk = 0xffff0000u, sel = 0x07060302u;
  asm volatile("" : "+s"(msk), "+s"(sel));
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      if (MF) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(a8), "v"(b8));
#pragma unroll
      for (int i = 0; i < NV; ++i) one<KIND>(x[(m * NV + i) & 7], y[(m * NV + i) & 7], u[(m * NV + i) & 7], msk, sel);
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += x[i].x + x[i].y + (float)u[i] + acc[i][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND, int NV, int MF>
double run(int w) {
  float* out;
  const int blocks = 256 * w, iters = 20000;
  (void)hipMalloc(&out, blocks * 256 * 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<KIND, NV, MF><<<blocks, 256>>>(out, 100, 0.3f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<KIND, NV, MF><<<blocks, 256>>>(out, iters, 0.3f);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipFree(out);
  return ms * 1e-3 * 2.4e9 / iters / 8 / w;      // SIMD cycles (2.4 GHz) per unit of one wave
}

template <int KIND>
void kind(const char* name) {
  for (int w = 1; w <= 2; ++w) {
    const double alone = run<KIND, 8, 0>(w) / 8;
    const double m0 = run<KIND, 0, 1>(w), m2 = run<KIND, 2, 1>(w), m4 = run<KIND, 4, 1>(w), m8 = run<KIND, 8, 1>(w);
    printf("waves/SIMD=%d  %-18s alone %5.2f cycles/instr | beside one 16x16x32 bf16 MFMA (alone %5.1f): +2 -> %5.1f  +4 -> %5.1f  +8 -> %5.1f\n",
           w, name, alone, m0, m2, m4, m8);
  }
}

int main() {
  kind<0>("v_sub_f32");
  kind<1>("v_pk_add_f32");
  kind<2>("v_and_b32");
  kind<3>("v_perm_b32");
  kind<4>("v_exp_f32");
  kind<5>("v_pk_mul_f32");
  kind<6>("v_pk_fma_f32");
  kind<7>("v_fma_f32");
  kind<8>("v_cvt_pk_bf16_f32");
  kind<9>("v_max3_f32");
  kind<10>("v_dot2_f32_bf16");
  kind<11>("v_dot2c_f32_bf16");
  kind<12>("v_pack_b32_f16");
  kind<13>("v_and_b32 (vgpr)");
  kind<14>("v_add_f32");
  return 0;
}
