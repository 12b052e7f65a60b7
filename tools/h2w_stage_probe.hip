// Dev tool (round 5): the d_head 16 attention stage on v_mfma_f32_32x32x16 tiles as a register-only loop, priced BEFORE the
// kernel is built.  One stage = 32 queries x 32 keys of one wave = 16 scores per lane, like the
// 16 x 64 stage of attention_h2.hip, so the vector work is the same 56 instructions; the matrix work becomes
//     6 x v_mfma_f32_32x32x16_bf16   S^T = K Q^T, one bf16-triple term per MFMA (d = 16 = the MFMA's contraction)
//     4 x v_mfma_f32_32x32x16_f16    O^T += [V0; V1] P: the two fp16 pieces of V stacked on the 32 M rows, so p0 yields
//                                    v0 p0 + v1 p0 in ONE MFMA and p1 yields v0 p1 (+ the free v1 p1); 2 k-steps of 16 keys
// = 10 MFMAs (320 matrix cycles, 80 cycles of held vector issue) against 18 (288 / 144).
//   hipcc --offload-arch=gfx950 -O3 tools/h2w_stage_probe.hip -o tools/bin/h2w_stage_probe && tools/bin/h2w_stage_probe
// DEP bit 1: the exp stream reads the S accumulator the previous stage's MFMAs wrote; bit 2: P.V reads the previous stage's pieces
// VAR: 0 = the full stage, 1 = without the 16 row-sum adds, 2 = without the exps (v_mov instead), 3 = MFMAs only, 4 = vector only
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));


// vector-instruction orders: 0 = the kernel's natural order, evenly counted; 1-3 = tools/h2w_sched.py with 24 / 28 / 32 cycles of
// vector issue per MFMA gap (v_exp 8, others 4; exps spread two per gap)
struct Sched { int order[56]; int gap_end[10]; };
__device__ constexpr Sched SCHED[4] = {
  {{0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,35,36,37,38,39,40,41,42,43,44,45,46,47,48,49,50,51,52,53,54,55},
   {5,11,16,22,28,33,39,44,50,56}},
  {{0, 1, 2, 3, 4, 5, 14, 15, 6, 7, 16, 17, 8, 9, 28, 29, 10, 11, 30, 31, 12, 13, 42, 18, 19, 20, 21, 43, 22, 23, 24, 25, 44, 26, 27, 32, 33, 45, 34, 35, 36, 37, 38, 39, 40, 41, 46, 47, 48, 49, 50, 51, 52, 53, 54, 55},
   {2, 6, 10, 14, 18, 22, 27, 32, 37, 56}},
  {{0, 1, 2, 3, 4, 5, 6, 14, 15, 7, 8, 9, 16, 17, 10, 11, 12, 28, 29, 13, 18, 19, 30, 31, 20, 21, 22, 42, 23, 24, 25, 26, 27, 43, 32, 33, 34, 35, 36, 44, 37, 38, 39, 40, 41, 45, 46, 47, 48, 49, 50, 51, 52, 53, 54, 55},
   {2, 7, 12, 17, 22, 27, 33, 39, 45, 56}},
  {{0, 1, 2, 3, 4, 5, 6, 7, 14, 15, 8, 9, 10, 11, 16, 17, 12, 13, 18, 19, 28, 29, 20, 21, 22, 23, 30, 31, 24, 25, 26, 27, 42, 32, 33, 34, 35, 36, 37, 43, 38, 39, 40, 41, 46, 50, 44, 52, 48, 54, 45, 47, 49, 51, 53, 55},
   {2, 8, 14, 20, 26, 32, 39, 46, 50, 56}}};

template <int DEP, int VAR, int PAT>
__global__ __launch_bounds__(256, 2) void stage_kernel(float* out, int iters, float seed, float one) {
  u32x4 kop[3], qop[2][3], vop[2];
  f32x16 S[2], O[2], negm[2];
  u32x4 pop[2][2][2];      // [parity][piece][k-step]
  float priv[16];
  for (int j = 0; j < 3; ++j) {
    kop[j] = u32x4{0x3c003c00u + j, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
    for (int g = 0; g < 2; ++g) qop[g][j] = u32x4{0x3c003c00u, 0x3c003c00u + j, 0x3c003c00u + g, 0x3c003c00u};
  }
  for (int i = 0; i < 2; ++i) {
    vop[i] = u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u + i, 0x3c003c00u};
    for (int j = 0; j < 2; ++j)
      for (int k = 0; k < 2; ++k) pop[i][j][k] = u32x4{0x3c003c00u, 0x3c003c00u + k, 0x3c003c00u + i, 0x3c003c00u + j};
    for (int r = 0; r < 16; ++r) { S[i][r] = seed; O[i][r] = 0.f; negm[i][r] = -seed; }
  }
  for (int i = 0; i < 16; ++i) priv[i] = seed * 0.01f * i;
  float sum0 = 0, sum1 = 0;

  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int st = 0; st < 2; ++st) {          // two stages = the two query groups of one 32-key block
      const int par = st & 1;
      float pe[16], ad[8];
      unsigned u[8], r2[8];
      // 56 vector steps in groups of 14 per four scores, as in the kernel: 4 exp, 2 cvt_pk, 2 adds, 2 mixlo, 2 mixhi, 2 adds
      auto vstep = [&](int n) {
        const int kt = n / 14, r = n - kt * 14;
        if (r < 4) {
          if (VAR == 2) asm volatile("v_mov_b32 %0, %1" : "=v"(pe[4 * kt + r]) : "v"(priv[4 * kt + r]));
          else if (DEP & 1) asm volatile("v_exp_f32 %0, %1" : "=v"(pe[4 * kt + r]) : "v"(S[par][4 * kt + r]));
          else asm volatile("v_exp_f32 %0, %1" : "=v"(pe[4 * kt + r]) : "v"(priv[4 * kt + r]));
        } else if (r < 6) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u[2 * kt + r - 4]) : "v"(pe[4 * kt + 2 * (r - 4)]), "v"(pe[4 * kt + 2 * (r - 4) + 1]));
        else if (r < 8) { if (VAR != 1) asm volatile("v_add_f32 %0, %1, %2" : "=v"(ad[2 * kt + r - 6]) : "v"(pe[4 * kt + r - 6]), "v"(pe[4 * kt + r - 4])); }
        else if (r < 10) asm volatile("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r2[2 * kt + r - 8]) : "v"(pe[4 * kt + 2 * (r - 8)]), "s"(one), "v"(u[2 * kt + r - 8]));
        else if (r < 12) asm volatile("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(r2[2 * kt + r - 10]) : "v"(pe[4 * kt + 2 * (r - 10) + 1]), "s"(one), "v"(u[2 * kt + r - 10]));
        else if (r == 12) { if (VAR != 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(sum0) : "v"(ad[2 * kt])); }
        else { if (VAR != 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(sum1) : "v"(ad[2 * kt + 1])); }
      };
      // 10 MFMA slots: q q p q p q p q p q  (QK^T of the next stage into S[par ^ 1], P.V of the previous one from pop[par ^ 1])
      auto mstep = [&](int i) {
        constexpr int kind[10] = {0, 0, 1, 0, 1, 0, 1, 0, 1, 0};
        int nq = 0, np = 0;
        for (int k = 0; k < i; ++k) (kind[k] ? np : nq)++;
        if (kind[i]) {
          const int ks = np >> 1, piece = 1 - (np & 1);       // small term first: p1, then p0
          if (DEP & 2) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(O[par ^ 1]) : "v"(vop[ks]), "v"(pop[par ^ 1][piece][ks]));
          else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(O[par ^ 1]) : "v"(vop[ks]), "v"(vop[piece]));
        } else {
          constexpr int TA[6] = {0, 1, 0, 2, 1, 0}, TB[6] = {0, 0, 1, 0, 1, 2};
          if (nq == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=v"(S[par ^ 1]) : "v"(kop[TA[nq]]), "v"(qop[par ^ 1][TB[nq]]), "v"(negm[par ^ 1]));
          else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(S[par ^ 1]) : "v"(kop[TA[nq]]), "v"(qop[par ^ 1][TB[nq]]));
        }
      };
      constexpr int NMF = 10, NVS = 56;
      if (PAT < 4) {
#pragma unroll
        for (int i = 0; i < NMF; ++i) {
          if (VAR != 4) mstep(i);
          if (VAR != 3)
#pragma unroll
            for (int n = (i ? SCHED[PAT].gap_end[i - 1] : 0); n < SCHED[PAT].gap_end[i]; ++n) vstep(SCHED[PAT].order[n]);
        }
      } else {
#pragma unroll
        for (int i = 0; i < NMF; ++i) mstep(i);
#pragma unroll
        for (int n = 0; n < NVS; ++n) vstep(n);
      }
      if (VAR != 3)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
          const int c = kt >> 1, o = (kt & 1) * 2;
          pop[par][0][c][o] = u[2 * kt]; pop[par][1][c][o] = r2[2 * kt];
          pop[par][0][c][o + 1] = u[2 * kt + 1]; pop[par][1][c][o + 1] = r2[2 * kt + 1];
        }
      if (!(DEP & 1))
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(priv[i]));
    }
  }
  float s = sum0 + sum1;
  for (int i = 0; i < 2; ++i) s += O[i][0] + S[i][i] + O[i][15];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)pop[0][0][0][0] + (float)pop[1][1][1][3];
}

template <int DEP, int VAR, int PAT>
void run(const char* name, int w, float* out) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 4000;
  stage_kernel<DEP, VAR, PAT><<<256 * w, 256>>>(out, 50, 0.3f, 1.0f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  stage_kernel<DEP, VAR, PAT><<<256 * w, 256>>>(out, iters, 0.3f, 1.0f);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("waves/SIMD=%d  %-58s %7.1f ns per stage and SIMD (= %6.1f cycles at 2.4 GHz)\n", w, name, ms * 1e6 / iters / 2 / w,
         ms * 1e-3 * 2.4e9 / iters / 2 / w);
}

int main() {
  float* out;
  (void)hipMalloc(&out, 512 * 256 * 4);
  for (int w = 1; w <= 2; ++w) {
    run<3, 0, 0>("32x32x16 stage: 10 MFMA + 56 vector, both deps, spread", w, out);
    run<0, 0, 0>("  no dependencies, spread", w, out);
    run<3, 0, 9>("  both deps, clumped (MFMAs first)", w, out);
    run<3, 0, 1>("  scheduled: exps spread, 24 cycles per gap, rest at the end", w, out);
    run<3, 0, 2>("  scheduled: exps spread, 28 cycles per gap", w, out);
    run<3, 0, 3>("  scheduled: exps spread, 32 cycles per gap", w, out);
    run<3, 4, 2>("  scheduled 28: vector only", w, out);
    run<3, 1, 0>("  without the 16 row-sum adds", w, out);
    run<3, 2, 0>("  v_mov in place of v_exp", w, out);
    run<3, 3, 0>("  MFMAs only", w, out);
    run<3, 4, 0>("  vector only", w, out);
  }
  printf("(compare tools/h2_stage_probe.hip 'both (the kernel), spread': the 16x16x32 stage of the same 1024 scores)\n");
  return 0;
}
