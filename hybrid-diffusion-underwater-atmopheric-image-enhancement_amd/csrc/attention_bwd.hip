// Backward of the flash self-attention core (attention.hip), fp32 MFMA 16x16x4, gfx950.
//
// Reference: autograd through nn.MultiheadAttention(C, 8)(h, h, h) (ModelCondition.py:189, 204-208) in
// TrainCondition.py:60 (loss.backward()).  P is recomputed from Q, K and the forward's log2-domain log-sum-exp
// (lse2 = m + log2 l), never stored:  p = exp2(s2 - lse2),  dP = dO V^T,  dS = P o (dP - delta),  delta = rowsum(dO o O).
//
// ONE kernel, the minimal five MFMA products per (query tile, key tile), no atomics, bitwise reproducible:
//   a workgroup owns a RANGE of keys and walks it in blocks of 64*NK keys (16*NK per wave, K and V held in registers as
//   MFMA B operands); for each block it sweeps all queries in 64-query tiles staged through LDS (Q, dO, -lse2, -delta):
//     S  = Q K^T - lse2      rows = queries (registers), columns = keys (lanes); -lse2 is the initial accumulator
//     dP = dO V^T - delta    same shape; -delta is the initial accumulator
//     P = exp2(S) ; dS = P o dP                                   (8 VALU instructions per 16x16 tile, nothing else)
//     dV^T += dO^T P ; dK^T += Q^T dS     the S / dP accumulators ARE the B operands (MFMA r consumes register r):
//                                         dK, dV stay in registers for the whole sweep -- no sum across workgroups
//     dQ^T += K^T dS^T                    sums over keys = over LANES of dS: the tile crosses LDS once, transposed
//                                         (one ds_write_b128 + four ds_read_b32 per lane, wave-private scratch, no barrier)
//   dQ of a 64-query tile is summed over the workgroup's four waves through LDS in a fixed order and added to a partial
//   slab in global memory: slab s belongs to key range s alone, and inside a workgroup the SAME thread re-reads the value it
//   stored during the previous key block (plain load / store, program order) -- so no atomics and no inter-workgroup
//   protocol.  The slabs of all ranges are summed in order by a small reduce kernel (or, with one range, the slab is the
//   output itself).  Extra traffic: 8 bytes per (key block, query, channel) = 1 byte per 320 FLOP, ~3 % of the HBM rate.
// All tensors keep the [B][3C][L] / [B][C][L] layouts of the forward.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

using namespace hdiff;

namespace {

constexpr int KT = 64;            // queries per staged tile
constexpr int KROW = KT + 4;      // LDS row stride of the staged tiles (floats): 16-byte aligned, conflict-free column reads
constexpr int ATT_THREADS = 256;
constexpr int TS = 20;            // row stride of the 16x16 transposition scratch: 4*TS = 16 (mod 32) -> conflict-free both ways

// delta[b][h][q] = sum_d dO[b][h*D+d][q] * O[b][h*D+d][q]
__global__ void mha_delta_kernel(const float* __restrict__ o, const float* __restrict__ d_o, float* __restrict__ delta,
                                 int C, int D, int L, int total) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int q = idx % L;
  const int bh = idx / L;
  const int heads = C / D;
  const int b = bh / heads, h = bh - b * heads;
  const size_t base = ((size_t)b * C + (size_t)h * D) * L + q;
  float s = 0.f;
  for (int d = 0; d < D; ++d) s = fmaf(o[base + (size_t)d * L], d_o[base + (size_t)d * L], s);
  delta[idx] = s;
}

// dqkv[b][c][q] (Q third) = sum over key ranges, in order, of the partial slabs [nsplit][B][C][L]
__global__ void mha_dq_reduce_kernel(const float* __restrict__ part, float* __restrict__ dqkv, int nsplit, int C, int L,
                                     size_t slab_floats, size_t per_sample) {
  const int b = blockIdx.y;
  const float* src = part + (size_t)b * per_sample;
  float* dst = dqkv + (size_t)b * 3 * per_sample;
  if ((per_sample & 3) == 0) {
    const size_t n4 = per_sample >> 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
      float4 a = reinterpret_cast<const float4*>(src)[i];
      for (int s = 1; s < nsplit; ++s) {
        const float4 v = reinterpret_cast<const float4*>(src + (size_t)s * slab_floats)[i];
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
      }
      reinterpret_cast<float4*>(dst)[i] = a;
    }
  } else {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_sample; i += (size_t)gridDim.x * blockDim.x) {
      float a = src[i];
      for (int s = 1; s < nsplit; ++s) a += src[(size_t)s * slab_floats + i];
      dst[i] = a;
    }
  }
}

struct BwdArgs {
  const float* qkv;
  const float* d_o;
  const float* lse2;
  const float* delta;
  float* dqkv;
  float* dq_part;             // partial dQ slabs: element (split s, sample b, channel c, query q) at s*split_stride + b*batch_stride + c*L + q
  size_t split_stride, batch_stride;
  int C, L, kb_per_split;     // key blocks (of 64*NK keys) per key range
  float qscale, inv_sqrt_d;
};

// FAST: L is a multiple of the key block (hence of the 64-query tile and of 4): no bounds checks, no masks, no branches in
// the staging code.  The generic instantiation handles ragged / unaligned sequences.
template <int D, int NK, bool FAST>
__global__ __launch_bounds__(ATT_THREADS, (D >= 48 ? 1 : 2)) void mha_bwd_fused_kernel(const BwdArgs a) {
  constexpr int KS = D / 4;                       // k-steps of the S / dP products
  constexpr int MT = (D + 15) / 16;               // 16-row M tiles of the accumulating products
  constexpr int DP = MT * 16;
  constexpr int KB = 64 * NK;                     // keys per block of the workgroup
  constexpr int DQS = KT + 4;                     // row stride of the per-wave dQ partial tiles
  constexpr int DQ_BUFS = (MT == 1) ? 2 : 1;      // double-buffered when it fits: one barrier per staged tile instead of two
  constexpr int NV4 = 2 * D * (KT / 4);           // float4 of one staged Q + dO tile
  constexpr int NLD = (NV4 + ATT_THREADS - 1) / ATT_THREADS;
  constexpr bool OLD_EARLY = (MT == 1);
  constexpr bool OPS_AHEAD = (MT == 1);           // operands of the next query subtile fetched during the current one
  constexpr bool PIPELINED = (MT == 1) && FAST && (NK >= 2);   // chains of the next pair issued ahead of this pair's VALU step (d_head 32: 47 spilled registers with the second S / dP pair)
  constexpr bool TWO_AHEAD = FAST && (MT == 1) && (NK % 2 == 0);   // query-tile loads two tiles ahead (ntiles = L / 64 is then even)
  constexpr bool KT_FENCE = (MT > 1);             // d_head 32: keep the key tiles' MFMA groups apart (register budget, see do_tile)

  __shared__ __attribute__((aligned(16))) float sQ[2][DP * KROW];
  __shared__ __attribute__((aligned(16))) float sO[2][DP * KROW];
  __shared__ __attribute__((aligned(16))) float sL[2][KT];          // -lse2 of the staged queries (-inf beyond L)
  __shared__ __attribute__((aligned(16))) float sD[2][KT];          // -delta
  __shared__ __attribute__((aligned(16))) float sT[4][16 * TS];     // per-wave transposition scratch for dS
  __shared__ __attribute__((aligned(16))) float sDQ[DQ_BUFS][4][DP * DQS];

  const int C = a.C, L = a.L;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  const TileId tile = xcd_tile();
  const int head = tile.head, b = tile.b, heads = gridDim.y, split = tile.x;
  const float* qbase = a.qkv + ((size_t)b * 3 * C + (size_t)head * D) * L;
  const float* kbase = qbase + (size_t)C * L;
  const float* vbase = kbase + (size_t)C * L;
  const float* dobase = a.d_o + ((size_t)b * C + (size_t)head * D) * L;
  const float* lbase = a.lse2 + ((size_t)b * heads + head) * L;
  const float* dbase = a.delta + ((size_t)b * heads + head) * L;
  float* part = a.dq_part + (size_t)split * a.split_stride + (size_t)b * a.batch_stride + (size_t)head * D * L;
  float* kout = a.dqkv + ((size_t)b * 3 * C + (size_t)C + (size_t)head * D) * L;
  float* vout = kout + (size_t)C * L;
  const bool vec_ok = FAST || (L & 3) == 0;
  const int ntiles = (L + KT - 1) / KT;
  const int nkb_total = (L + KB - 1) / KB;
  const int kb_begin = split * a.kb_per_split;
  const int kb_end = (kb_begin + a.kb_per_split < nkb_total) ? kb_begin + a.kb_per_split : nkb_total;

  // rows D..DP-1 of the staged tiles are the zero padding of the accumulating products' A operands (D < 16)
  if (DP > D) {
    for (int idx = tid; idx < 2 * (DP - D) * KROW; idx += ATT_THREADS) {
      const int bufi = idx / ((DP - D) * KROW), rem = idx - bufi * (DP - D) * KROW;
      sQ[bufi][D * KROW + rem] = 0.f;
      sO[bufi][D * KROW + rem] = 0.f;
    }
  }

  // ---- staging of one 64-query tile: global -> registers (before the MFMA work) -> LDS (after it)
  struct Stage { float4 v[NLD]; float4 ld; int col; };
  Stage st0, st1;            // st1: second register set of the two-tiles-ahead loop (TWO_AHEAD below)
  st0.ld = st1.ld = make_float4(0.f, 0.f, 0.f, 0.f);
  st0.col = st1.col = 0;
  auto load4 = [&](const float* src, int col) {
    if (FAST) return *reinterpret_cast<const float4*>(src);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (vec_ok) {
      if (col < L) v = *reinterpret_cast<const float4*>(src);
    } else {
      if (col + 0 < L) v.x = src[0];
      if (col + 1 < L) v.y = src[1];
      if (col + 2 < L) v.z = src[2];
      if (col + 3 < L) v.w = src[3];
    }
    return v;
  };
  auto stage_load = [&](Stage& st, int t) {
    const int q0 = t * KT;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = tid + i * ATT_THREADS;
      if (idx < NV4) {
        const int row = idx >> 4, col = q0 + (idx & 15) * 4;
        const float* src = (row < D ? qbase + (size_t)row * L : dobase + (size_t)(row - D) * L) + col;
        st.v[i] = load4(src, col);
      }
    }
    if (tid < 32) {
      const int col = q0 + (tid & 15) * 4;
      st.ld = load4((tid < 16 ? lbase : dbase) + col, col);     // negated when it is stored (no wait on the load here)
      st.col = col;
    }
  };
  auto stage_store = [&](const Stage& st, int buf) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = tid + i * ATT_THREADS;
      if (idx < NV4) {
        const int row = idx >> 4, seg = idx & 15;
        float* dst = (row < D) ? &sQ[buf][row * KROW + seg * 4] : &sO[buf][(row - D) * KROW + seg * 4];
        *reinterpret_cast<float4*>(dst) = st.v[i];
      }
    }
    if (tid < 32) {
      float4 v = make_float4(-st.ld.x, -st.ld.y, -st.ld.z, -st.ld.w);
      if (!FAST && tid < 16) {
        // -lse2 is -inf for queries beyond L: p = exp2(s - inf) = 0 there
        if (st.col + 0 >= L) v.x = -__builtin_inff();
        if (st.col + 1 >= L) v.y = -__builtin_inff();
        if (st.col + 2 >= L) v.z = -__builtin_inff();
        if (st.col + 3 >= L) v.w = -__builtin_inff();
      }
      *reinterpret_cast<float4*>(tid < 16 ? &sL[buf][tid * 4] : &sD[buf][(tid - 16) * 4]) = v;
    }
  };

  for (int kb = kb_begin; kb < kb_end; ++kb) {
    const int kblk0 = kb * KB + wave * (16 * NK);
    const bool first_block = (kb == kb_begin);        // the slab is written, not added to
    const bool ragged_keys = !FAST && (kb * KB + KB > L);

    // this wave's keys: B operands of S and dP (key on the lane), and K^T as the A operand of the dQ product
    // (channel on the lane, keys 4g..4g+3 of each 16-key tile, pre-scaled by 1/sqrt(d))
    float kreg[NK][KS], vreg[NK][KS];
    f32x4 ktr[NK][MT];
    f32x4 dK[MT][NK], dV[MT][NK];
    bool kvalid[NK];
#pragma unroll
    for (int kt = 0; kt < NK; ++kt) {
      const int key = kblk0 + kt * 16 + i16;
      const bool ok = FAST || key < L;
      kvalid[kt] = ok;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        kreg[kt][s] = ok ? kbase[(size_t)(4 * s + g) * L + key] * a.qscale : 0.f;
        vreg[kt][s] = ok ? vbase[(size_t)(4 * s + g) * L + key] : 0.f;
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int d = mt * 16 + i16, k0 = kblk0 + kt * 16 + 4 * g;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (d < D) v = load4(kbase + (size_t)d * L + k0, k0);
        ktr[kt][mt] = f32x4{v.x * a.inv_sqrt_d, v.y * a.inv_sqrt_d, v.z * a.inv_sqrt_d, v.w * a.inv_sqrt_d};
        dK[mt][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
        dV[mt][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }

    // One staged query tile.  MASK: the key block holds keys >= L (their P must not reach dQ even as 0 * inf).
    // operands of one 16-query subtile: A fragments of S / dP (query on the lane), of the accumulating products (channel
    // on the lane, queries 4g..4g+3), and the row constants that enter as initial accumulators
    struct QOps {
      float qa[KS], doa[KS];
      f32x4 qv[MT], dov[MT], nl, nd;
    };
    auto load_ops = [&](QOps& o, int buf, int qs) {
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        o.qa[s] = sQ[buf][(4 * s + g) * KROW + qs * 16 + i16];
        o.doa[s] = sO[buf][(4 * s + g) * KROW + qs * 16 + i16];
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const float4 a4 = *reinterpret_cast<const float4*>(&sQ[buf][(mt * 16 + i16) * KROW + qs * 16 + 4 * g]);
        const float4 b4 = *reinterpret_cast<const float4*>(&sO[buf][(mt * 16 + i16) * KROW + qs * 16 + 4 * g]);
        o.qv[mt] = f32x4{a4.x, a4.y, a4.z, a4.w};
        o.dov[mt] = f32x4{b4.x, b4.y, b4.z, b4.w};
      }
      // rows of the accumulators are queries 4g + r of this subtile: -lse2 / -delta enter as the initial accumulators
      const float4 l4 = *reinterpret_cast<const float4*>(&sL[buf][qs * 16 + 4 * g]);
      const float4 d4 = *reinterpret_cast<const float4*>(&sD[buf][qs * 16 + 4 * g]);
      o.nl = f32x4{l4.x, l4.y, l4.z, l4.w};
      o.nd = f32x4{d4.x, d4.y, d4.z, d4.w};
    };

    // The MFMA work of one (query subtile, key tile) pair: the two 16x16 chains S / dP, the VALU step P = exp2(S),
    // dS = P o dP, the transposition of dS through the wave's scratch, and the three accumulating products.
    auto chains = [&](const QOps& o, int kt, f32x4& S, f32x4& dP) {
      S = o.nl;
      dP = o.nd;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        S = __builtin_amdgcn_mfma_f32_16x16x4f32(o.qa[s], kreg[kt][s], S, 0, 0, 0);
        dP = __builtin_amdgcn_mfma_f32_16x16x4f32(o.doa[s], vreg[kt][s], dP, 0, 0, 0);
      }
    };
    // (tried: dS as two v_pk_mul_f32 through inline asm, with the four v_exp pinned above them for the transcendental-use
    // wait state -- 1 % SLOWER: the pinning costs more interleaving freedom than the two saved instructions are worth)
    auto softmax_grad = [&](auto mask_tag, int kt, const f32x4& S, const f32x4& dP, f32x4& P, f32x4& dS, f32x4& tr) {
      constexpr bool MASK = decltype(mask_tag)::value;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float p = __builtin_amdgcn_exp2f(S[r]);
        if (MASK && !kvalid[kt]) p = 0.f;
        P[r] = p;
        dS[r] = p * dP[r];
      }
      // transpose dS through the wave's scratch: [key = lane column][query 4g..4g+3] -> rows of keys for the reader
      asm volatile("" ::: "memory");
      *reinterpret_cast<f32x4*>(&sT[wave][i16 * TS + 4 * g]) = dS;
      asm volatile("" ::: "memory");      // DS operations of one wave execute in order: no wait between write and read
#pragma unroll
      for (int r = 0; r < 4; ++r) tr[r] = sT[wave][(4 * g + r) * TS + i16];
      asm volatile("" ::: "memory");
    };
    auto accumulate = [&](const QOps& o, int kt, const f32x4& P, const f32x4& dS, const f32x4& tr, f32x4 (&dQa)[MT]) {
      // dV / dK first (they only need P / dS), dQ behind them: its operand is the LDS read-back, covered by the MFMAs in
      // front, and its accumulation chain is spread out instead of four dependent MFMAs back to back
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          dV[mt][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.dov[mt][r], P[r], dV[mt][kt], 0, 0, 0);
          dK[mt][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.qv[mt][r], dS[r], dK[mt][kt], 0, 0, 0);
          if (r >= 2) dQa[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ktr[kt][mt][r - 2], tr[r - 2], dQa[mt], 0, 0, 0);
        }
#pragma unroll
      for (int r = 2; r < 4; ++r)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          dQa[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ktr[kt][mt][r], tr[r], dQa[mt], 0, 0, 0);
    };
    auto store_dq = [&](int dqbuf, int qs, const f32x4 (&dQa)[MT]) {
      // dQ^T tile of this wave: lane = query, registers = channels 4g..4g+3
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) sDQ[dqbuf][wave][(mt * 16 + 4 * g + r) * DQS + qs * 16 + i16] = dQa[mt][r];
    };

    // One staged query tile.  MASK: the key block holds keys >= L (their P must not reach dQ even as 0 * inf).
    auto do_tile = [&](auto mask_tag, int buf, int dqbuf) {
      QOps ops[2];
      if constexpr (PIPELINED) {
        // Software pipeline over the 4 * NK (subtile, key tile) pairs: the S / dP chains of pair i+1 are ISSUED before the
        // VALU step of pair i, so that step never waits for an MFMA result (its chains were issued a whole pair earlier) and
        // the matrix pipe has the next chains queued while the wave does its exp / multiply / LDS transposition.  Operands of
        // subtile qs+1 are fetched from LDS at the first pair of subtile qs.  Scheduling barriers pin the stage order.
        constexpr int NPAIR = 4 * NK;
        f32x4 S[2], dP[2], dQa[MT];
        load_ops(ops[0], buf, 0);
        __builtin_amdgcn_sched_barrier(0);
        chains(ops[0], 0, S[0], dP[0]);
#pragma unroll
        for (int i = 0; i < NPAIR; ++i) {
          const int qs = i / NK, kt = i - qs * NK;
          if (kt == 0) {
            if (qs + 1 < 4) load_ops(ops[(qs + 1) & 1], buf, qs + 1);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) dQa[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
          __builtin_amdgcn_sched_barrier(0);
          if (i + 1 < NPAIR) chains(ops[((i + 1) / NK) & 1], (i + 1) % NK, S[(i + 1) & 1], dP[(i + 1) & 1]);
          __builtin_amdgcn_sched_barrier(0);
          f32x4 P, dS, tr;
          softmax_grad(mask_tag, kt, S[i & 1], dP[i & 1], P, dS, tr);
          __builtin_amdgcn_sched_barrier(0);
          accumulate(ops[qs & 1], kt, P, dS, tr, dQa);
          if (kt == NK - 1) store_dq(dqbuf, qs, dQa);
        }
      } else {
        if (OPS_AHEAD) load_ops(ops[0], buf, 0);
#pragma unroll
        for (int qs = 0; qs < 4; ++qs) {
          // The scheduler must not hoist operand loads across subtiles on its own (it then holds the operands of all four
          // live at once: 96 extra registers, one wave per SIMD).  d_head <= 16 has the registers to fetch ONE subtile ahead
          // by hand, so that no subtile starts with an LDS round trip; d_head 32 loads them in place.
          __builtin_amdgcn_sched_barrier(0);
          if (OPS_AHEAD) {
            if (qs + 1 < 4) load_ops(ops[(qs + 1) & 1], buf, qs + 1);
          } else {
            load_ops(ops[qs & 1], buf, qs);
          }
          __builtin_amdgcn_sched_barrier(0);
          const QOps& o = ops[qs & 1];
          f32x4 dQa[MT];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) dQa[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kt = 0; kt < NK; ++kt) {
            if (KT_FENCE) __builtin_amdgcn_sched_barrier(0);
            f32x4 S, dP, P, dS, tr;
            chains(o, kt, S, dP);
            softmax_grad(mask_tag, kt, S, dP, P, dS, tr);
            // pin the read-back HERE, in front of the dV / dK MFMAs that cover its latency (the scheduler otherwise sinks it
            // next to its first use and the wave waits a full LDS round trip per pair)
            __builtin_amdgcn_sched_barrier(0);
            accumulate(o, kt, P, dS, tr, dQa);
          }
          store_dq(dqbuf, qs, dQa);
        }
      }
    };

    // One tile of the sweep: start the global loads of tile t_ld into `ld`, fetch the running dQ partial, do the MFMA work of
    // tile t (LDS buffer buf), publish the tile held in `stf` to the other LDS buffer, then reduce this tile's dQ.
    auto tile_step = [&](int t, int buf, int dqbuf, Stage& ld, int t_ld, const Stage& stf) {
      stage_load(ld, t_ld);
      // this thread's share of the running dQ partial of tile t: rows d = idx / 16, queries 4 * (idx % 16) ..+3.
      // d_head <= 16: fetched before the MFMA work; d_head 32 has no registers to hold it that long and fetches it behind
      // (the co-resident workgroup's waves cover the wait).  The slab is streamed (each line is touched once per key block,
      // 4 MB of other slab lines later): non-temporal loads / stores keep it from evicting the Q / dO tiles that the 32
      // workgroups of a (head, sample) pair share through L2 / Infinity Cache.
      float4 old[MT];
      auto fetch_old = [&]() {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const int idx = tid + i * ATT_THREADS;
          const int d = idx >> 4, q = t * KT + (idx & 15) * 4;
          old[i] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (!first_block && d < D) {
            if (FAST) {
              const f32x4 nv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(part + (size_t)d * L + q));
              old[i] = make_float4(nv[0], nv[1], nv[2], nv[3]);
            } else {
              old[i] = load4(part + (size_t)d * L + q, q);
            }
          }
        }
      };
      if (OLD_EARLY) fetch_old();
      if (!FAST && ragged_keys) do_tile(std::true_type{}, buf, dqbuf);
      else do_tile(std::false_type{}, buf, dqbuf);
      if (!OLD_EARLY) fetch_old();
      stage_store(stf, buf ^ 1);
      __syncthreads();
      // sum the four waves' partial tiles in wave order, add the running value, store
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int idx = tid + i * ATT_THREADS;
        const int d = idx >> 4, q = t * KT + (idx & 15) * 4;
        float4 acc = old[i];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const float4 v = *reinterpret_cast<const float4*>(&sDQ[dqbuf][w][d * DQS + (idx & 15) * 4]);
          acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        if (d < D) {
          float* dst = part + (size_t)d * L + q;
          if (FAST) {
            __builtin_nontemporal_store(f32x4{acc.x, acc.y, acc.z, acc.w}, reinterpret_cast<f32x4*>(dst));
          } else if (vec_ok) {
            if (q < L) *reinterpret_cast<float4*>(dst) = acc;
          } else {
            if (q + 0 < L) dst[0] = acc.x;
            if (q + 1 < L) dst[1] = acc.y;
            if (q + 2 < L) dst[2] = acc.z;
            if (q + 3 < L) dst[3] = acc.w;
          }
        }
      }
      if (DQ_BUFS == 1) __syncthreads();
    };

    __syncthreads();              // the previous key block's last tile is fully consumed (LDS buffers are reused)
    stage_load(st0, 0);
    stage_store(st0, 0);
    __syncthreads();
    if constexpr (TWO_AHEAD) {
      // Global loads run TWO tiles ahead on two register sets, two tiles per iteration (ntiles is even here): with the dQ
      // slab traffic in the memory system a Q / dO tile load takes longer than one tile of MFMA work (timing ablations:
      // without the slab traffic, or with the tile loads hitting in cache, the kernel is 5 % faster; one tile ahead the
      // wave stalls at the s_waitcnt in front of the LDS store).  The LDS / dQ buffer indices become compile-time constants.
      const int last = ntiles - 1;
      stage_load(st0, 1 < last ? 1 : last);
      for (int t = 0; t < ntiles; t += 2) {
        tile_step(t, 0, 0, st1, t + 2 < last ? t + 2 : last, st0);          // st0 holds tile t + 1
        tile_step(t + 1, 1, 1, st0, t + 3 < last ? t + 3 : last, st1);      // st1 holds tile t + 2
      }
    } else {
      for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1, dqbuf = (DQ_BUFS == 2) ? (t & 1) : 0;
        const int tn = (t + 1 < ntiles) ? t + 1 : t;     // the last iteration re-stages its own tile (branch-free; unused)
        tile_step(t, buf, dqbuf, st0, tn, st0);
      }
    }

    // ---- dK, dV of this key block (complete: the sweep covered every query)
#pragma unroll
    for (int kt = 0; kt < NK; ++kt) {
      const int key = kblk0 + kt * 16 + i16;
      if (FAST || key < L) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int d = mt * 16 + 4 * g + r;
            if (d < D) {
              kout[(size_t)d * L + key] = dK[mt][kt][r] * a.inv_sqrt_d;
              vout[(size_t)d * L + key] = dV[mt][kt][r];
            }
          }
      }
    }
  }
}

// Geometry shared by the workspace query and the launch.
struct BwdGeom { int nk, nkb_total, per, nsplit; };
static BwdGeom bwd_geometry(int B, int heads, int L, int D) {
  BwdGeom g;
  g.nk = (L <= 4096 || D >= 48) ? 1 : (D > 16 ? 2 : 4);   // short sequences: 64-key blocks for enough workgroups
  if (L % (64 * g.nk) != 0) g.nk = 1;                      // ragged sequences: the generic (bounds-checked) kernel, 64-key blocks
  const int KB = 64 * g.nk;
  g.nkb_total = cdiv(L, KB);
  const int pairs = B * heads;
  int want = cdiv(1024, pairs);                          // ~2 rounds of 2 workgroups per CU on 256 CUs
  if (want > g.nkb_total) want = g.nkb_total;
  if (want < 1) want = 1;
  g.per = cdiv(g.nkb_total, want);
  g.nsplit = cdiv(g.nkb_total, g.per);                   // no empty key range
  return g;
}

template <int D, int NK>
void launch_fused(const BwdArgs& a, const BwdGeom& g, int B, int heads, hipStream_t stream) {
  if (a.L % (64 * NK) == 0) {
    hipLaunchKernelGGL((mha_bwd_fused_kernel<D, NK, true>), dim3(g.nsplit, heads, B), dim3(ATT_THREADS), 0, stream, a);
  } else if constexpr (NK == 1) {
    // ragged / unaligned sequences: the generic instantiation exists for 64-key blocks only (bwd_geometry picks NK = 1 for
    // them: with more key tiles per wave its bounds handling spills registers in the MFMA loop)
    hipLaunchKernelGGL((mha_bwd_fused_kernel<D, 1, false>), dim3(g.nsplit, heads, B), dim3(ATT_THREADS), 0, stream, a);
  }
}

template <int D>
int launch_bwd(const float* qkv, const float* o, const float* d_o, const float* lse2, float* delta, float* dqkv, float* ws,
               int B, int C, int heads, int L, hipStream_t stream) {
  if constexpr (D == 16 || D == 32) {
    if (mha_bwd_x3_applicable(B, C, heads, L)) {       // bf16x3 mode: all five products on the bf16 matrix core
      HDIFF_CHECK_ARG(ws != nullptr, "mha_flash_bwd: this shape needs a workspace (hdiff_mha_flash_bwd_workspace)");
      const int total = B * heads * L;
      (void)hipGetLastError();
      hipLaunchKernelGGL(mha_delta_kernel, dim3(cdiv(total, 256)), dim3(256), 0, stream, o, d_o, delta, C, D, L, total);
      if (launch_mha_bwd_h2(qkv, d_o, lse2, delta, dqkv, ws, B, C, heads, L, stream)) {
        HDIFF_CHECK_LAUNCH("mha_bwd (split-bf16) kernels");
        return HDIFF_OK;
      }
      // the device refused the split kernel's LDS size (nothing of it was launched): the fp32-input backward below takes the call -- its
      // workspace need is never larger (hdiff_mha_flash_bwd_workspace reports the maximum of the two)
    }
  }
  const BwdGeom g = bwd_geometry(B, heads, L, D);
  HDIFF_CHECK_ARG(g.nsplit == 1 || ws != nullptr, "mha_flash_bwd: this shape needs a workspace (hdiff_mha_flash_bwd_workspace)");
  BwdArgs a;
  a.qkv = qkv; a.d_o = d_o; a.lse2 = lse2; a.delta = delta; a.dqkv = dqkv;
  a.C = C; a.L = L; a.kb_per_split = g.per;
  a.inv_sqrt_d = 1.0f / sqrtf((float)D);
  a.qscale = 1.4426950408889634f * a.inv_sqrt_d;
  const size_t per_sample = (size_t)C * L;
  if (g.nsplit == 1) {           // one key range: its slab is the Q third of the output itself
    a.dq_part = dqkv; a.split_stride = 0; a.batch_stride = 3 * per_sample;
  } else {
    a.dq_part = ws; a.split_stride = (size_t)B * per_sample; a.batch_stride = per_sample;
  }
  const int total = B * heads * L;
  (void)hipGetLastError();  // drop any stale error left by another HIP user in this thread
  hipLaunchKernelGGL(mha_delta_kernel, dim3(cdiv(total, 256)), dim3(256), 0, stream, o, d_o, delta, C, D, L, total);
  if constexpr (D >= 48) {
    launch_fused<D, 1>(a, g, B, heads, stream);             // d_head 48 / 64: 16 keys per wave is what the registers hold
  } else {
    if (g.nk == 1) launch_fused<D, 1>(a, g, B, heads, stream);
    else if (g.nk == 2) launch_fused<D, 2>(a, g, B, heads, stream);
    else launch_fused<D, (D > 16 ? 2 : 4)>(a, g, B, heads, stream);
  }
  if (g.nsplit > 1) {
    const size_t n4 = per_sample / 4 + 1;
    const int bx = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
    hipLaunchKernelGGL(mha_dq_reduce_kernel, dim3(bx, B), dim3(256), 0, stream, ws, dqkv, g.nsplit, C, L, a.split_stride,
                       per_sample);
  }
  HDIFF_CHECK_LAUNCH("mha_bwd kernels");
  return HDIFF_OK;
}

}  // namespace

extern "C" int hdiff_mha_flash_bwd_workspace(int B, int C, int heads, int L, int64_t* n_floats) {
  HDIFF_CHECK_ARG(n_floats, "mha_flash_bwd_workspace: null pointer");
  HDIFF_CHECK_ARG(B > 0 && L > 0 && heads > 0 && C % heads == 0, "mha_flash_bwd_workspace: bad sizes B=%d C=%d heads=%d L=%d",
                  B, C, heads, L);
  const BwdGeom g = bwd_geometry(B, heads, L, C / heads);
  *n_floats = (g.nsplit > 1) ? (int64_t)g.nsplit * B * C * L : 0;
  // The answer does not depend on the contraction mode: a caller that sizes the buffer, switches the mode and then calls
  // must not overrun it (the call takes no size).  Shapes the split-bf16 kernel covers get the larger of the two needs.
  if (mha_bwd_x3_shape_ok(B, C, heads, L)) {
    const int64_t x3 = mha_bwd_x3_workspace_floats(B, C, heads, L);
    if (x3 > *n_floats) *n_floats = x3;
  }
  return HDIFF_OK;
}

extern "C" int hdiff_mha_flash_bwd(const float* qkv, const float* o, const float* d_o, const float* lse2, float* delta,
                                   float* dqkv, float* ws, int B, int C, int heads, int L, hdiff_stream_t stream) {
  HDIFF_CHECK_ARG(qkv && o && d_o && lse2 && delta && dqkv, "mha_flash_bwd: null pointer");
  HDIFF_CHECK_ARG(B > 0 && L > 0 && heads > 0 && C % heads == 0, "mha_flash_bwd: bad sizes B=%d C=%d heads=%d L=%d", B, C,
                  heads, L);
  HDIFF_CHECK_ARG(B <= 65535 && heads <= 65535, "mha_flash_bwd: B=%d / heads=%d exceed the grid limits", B, heads);
  const int D = C / heads;
  hipStream_t s = (hipStream_t)stream;
  switch (D) {
    case 4: return launch_bwd<4>(qkv, o, d_o, lse2, delta, dqkv, ws, B, C, heads, L, s);
    case 8: return launch_bwd<8>(qkv, o, d_o, lse2, delta, dqkv, ws, B, C, heads, L, s);
    case 12: return launch_bwd<12>(qkv, o, d_o, lse2, delta, dqkv, ws, B, C, heads, L, s);
    case 16: return launch_bwd<16>(qkv, o, d_o, lse2, delta, dqkv, ws, B, C, heads, L, s);
    case 24: return launch_bwd<24>(qkv, o, d_o, lse2, delta, dqkv, ws, B, C, heads, L, s);
    case 32: return launch_bwd<32>(qkv, o, d_o, lse2, delta, dqkv, ws, B, C, heads, L, s);
    case 48: return launch_bwd<48>(qkv, o, d_o, lse2, delta, dqkv, ws, B, C, heads, L, s);
    case 64: return launch_bwd<64>(qkv, o, d_o, lse2, delta, dqkv, ws, B, C, heads, L, s);
    default: break;
  }
  hdiff::set_error("mha_flash_bwd: head dim %d not in {4, 8, 12, 16, 24, 32, 48, 64}", D);
  return HDIFF_ERR_INVALID;
}
