// Dev tool (round 4): the h2 attention stage (18 MFMA + 56 vector instructions, attention_h2.hip) as a register-only loop of
// asm statements in exactly the kernel's slot order, with and without the data dependencies between the two streams --
// what the stage costs when nothing but instruction issue and register dependencies are in play.
//   hipcc --offload-arch=gfx950 -O3 tools/h2_stage_probe.hip -o h2_stage_probe && ./h2_stage_probe
// DEP bit 1: the exp stream reads the S accumulators the MFMAs of the previous stage wrote (else: private registers)
// DEP bit 2: the P.V MFMAs read the pieces the vector stream wrote in the previous stage (else: constant operands)
// PER: vector instructions per slot pattern: 0 = even spread (3,3,3,...), 1 = all vector work after the MFMAs (clumped)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define MFMA_BF16(acc, a, b, c) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=v"(acc) : "v"(a), "v"(b), "v"(c))
#define MFMA_F16(acc, a, b) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))

// QKH = 1: the candidate with Q K^T on fp16 pairs of range-scaled operands (four products = two MFMAs per 16x16 tile, one
// K operand set) and one v_mul_f32 per score that takes the scale out again: 14 MFMA + 72 vector instructions per stage
template <int DEP, int PAT, int QKH>
__global__ __launch_bounds__(256, 2) void stage_kernel(float* out, int iters, float seed, float one) {
  constexpr int NMF = QKH ? 14 : 18, NVS = QKH ? 72 : 56, VPK = QKH ? 18 : 14;
  u32x4 kop[4][3], qcur[3], vop[2][2];
  f32x4 S[2][4], O[4], negm;
  u32x4 pop[2][2][2];
  float priv[16];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 3; ++j) kop[i][j] = u32x4{0x3c003c00u + i, 0x3c003c00u + j, 0x3c003c00u, 0x3c003c00u};
  for (int j = 0; j < 3; ++j) qcur[j] = u32x4{0x3c003c00u, 0x3c003c00u + j, 0x3c003c00u, 0x3c003c00u};
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j) {
      vop[i][j] = u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u + i, 0x3c003c00u + j};
      for (int k = 0; k < 2; ++k) pop[i][j][k] = u32x4{0x3c003c00u, 0x3c003c00u + k, 0x3c003c00u + i, 0x3c003c00u + j};
    }
  for (int i = 0; i < 4; ++i) { O[i] = f32x4{0, 0, 0, 0}; for (int p = 0; p < 2; ++p) S[p][i] = f32x4{seed, seed, seed, seed}; }
  for (int i = 0; i < 16; ++i) priv[i] = seed * 0.01f * i;
  negm = f32x4{-seed, -seed, -seed, -seed};
  float sum0 = 0, sum1 = 0;

  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const int par = st & 1;
      float pe[4][4], ad[4][2];
      unsigned u[4][2], r2[4][2];
      auto vstep = [&](int n) {
        const int kt = n / VPK;
        int r = n - kt * VPK;
        if (QKH) {
          if (r < 4) { asm volatile("v_mul_f32 %0, %1, %0" : "+v"(S[par][kt][r]) : "s"(one)); return; }
          r -= 4;
        }
        if (r < 4) {
          if (DEP & 1) asm volatile("v_exp_f32 %0, %1" : "=v"(pe[kt][r]) : "v"(S[par][kt][r]));
          else asm volatile("v_exp_f32 %0, %1" : "=v"(pe[kt][r]) : "v"(priv[kt * 4 + r]));
        } else if (r < 6) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u[kt][r - 4]) : "v"(pe[kt][2 * (r - 4)]), "v"(pe[kt][2 * (r - 4) + 1]));
        else if (r < 8) asm volatile("v_add_f32 %0, %1, %2" : "=v"(ad[kt][r - 6]) : "v"(pe[kt][r - 6]), "v"(pe[kt][r - 4]));
        else if (r < 10) asm volatile("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r2[kt][r - 8]) : "v"(pe[kt][2 * (r - 8)]), "s"(one), "v"(u[kt][r - 8]));
        else if (r < 12) asm volatile("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(r2[kt][r - 10]) : "v"(pe[kt][2 * (r - 10) + 1]), "s"(one), "v"(u[kt][r - 10]));
        else if (r == 12) asm volatile("v_add_f32 %0, %0, %1" : "+v"(sum0) : "v"(ad[kt][0]));
        else asm volatile("v_add_f32 %0, %0, %1" : "+v"(sum1) : "v"(ad[kt][1]));
      };
      auto mstep = [&](int i) {
        if (QKH) {
          const int np = (i + 1) * 6 / 14, pp = i * 6 / 14;
          if (np != pp) {
            const int n = pp, c = n / 3, term = n - c * 3;
            if (DEP & 2) MFMA_F16(O[(st + 3) & 3], vop[term == 0 ? 1 : 0][c], pop[par ^ 1][term == 1 ? 1 : 0][c]);
            else MFMA_F16(O[(st + 3) & 3], vop[term == 0 ? 1 : 0][c], vop[term == 1 ? 1 : 0][c]);
          } else {
            const int n = i - pp, j = n >> 2, kt = n & 3;
            if (j == 0) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %3" : "=v"(S[par ^ 1][kt]) : "v"(kop[kt][0]), "v"(qcur[j]), "v"(negm));
            else MFMA_F16(S[par ^ 1][kt], kop[kt][0], qcur[j]);
          }
          return;
        }
        if (i % 3 == 2) {
          const int n = i / 3, c = n / 3, term = n - c * 3;
          if (DEP & 2) MFMA_F16(O[(st + 3) & 3], vop[term == 0 ? 1 : 0][c], pop[par ^ 1][term == 1 ? 1 : 0][c]);
          else MFMA_F16(O[(st + 3) & 3], vop[term == 0 ? 1 : 0][c], vop[term == 1 ? 1 : 0][c]);
        } else {
          const int n = i - i / 3, j = n >> 2, kt = n & 3;
          if (j == 0) MFMA_BF16(S[par ^ 1][kt], kop[kt][j], qcur[j], negm);
          else MFMA_BF16(S[par ^ 1][kt], kop[kt][j], qcur[j], S[par ^ 1][kt]);
        }
      };
      if (PAT == 0) {
#pragma unroll
        for (int i = 0; i < NMF; ++i) {
          mstep(i);
#pragma unroll
          for (int n = NVS * i / NMF; n < NVS * (i + 1) / NMF; ++n) vstep(n);
        }
      } else {
#pragma unroll
        for (int i = 0; i < NMF; ++i) mstep(i);
#pragma unroll
        for (int n = 0; n < NVS; ++n) vstep(n);
      }
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        const int c = kt >> 1, o = (kt & 1) * 2;
        pop[par][0][c][o] = u[kt][0]; pop[par][1][c][o] = r2[kt][0];
        pop[par][0][c][o + 1] = u[kt][1]; pop[par][1][c][o + 1] = r2[kt][1];
      }
      if (!(DEP & 1))
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(priv[i]));
    }
  }
  float s = sum0 + sum1;
  for (int i = 0; i < 4; ++i) s += O[i][0] + S[0][i][0] + S[1][i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)pop[0][0][0][0] + (float)pop[1][1][1][3];
}

template <int DEP, int PAT, int QKH = 0>
void run(const char* name, int w, float* out) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 2000;
  stage_kernel<DEP, PAT, QKH><<<256 * w, 256>>>(out, 50, 0.3f, 1.0f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  stage_kernel<DEP, PAT, QKH><<<256 * w, 256>>>(out, iters, 0.3f, 1.0f);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("waves/SIMD=%d  %-50s %7.1f ns per stage and SIMD (= %6.1f cycles at 2.4 GHz)\n", w, name, ms * 1e6 / iters / 4 / w,
         ms * 1e-3 * 2.4e9 / iters / 4 / w);
}

int main() {
  float* out;
  (void)hipMalloc(&out, 512 * 256 * 4);
  for (int w = 1; w <= 2; ++w) {
    run<0, 0>("no dependencies, spread", w, out);
    run<1, 0>("exp reads MFMA results, spread", w, out);
    run<2, 0>("P.V reads vector results, spread", w, out);
    run<3, 0>("both (the kernel), spread", w, out);
    run<0, 1>("no dependencies, clumped", w, out);
    run<3, 1>("both, clumped", w, out);
    run<0, 0, 1>("fp16-pair QK candidate: no dependencies, spread", w, out);
    run<3, 0, 1>("fp16-pair QK candidate: both, spread", w, out);
  }
  printf("(kernel: 149.8 ms per launch at L = 65536, B = 16 = 285.7 ns per stage and SIMD)\n");
  return 0;
}
