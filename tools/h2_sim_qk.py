"""Dev tool (round 4): error of S = Q K^T (d_head 16) in candidate operand formats against float64 -- can Q and K be carried as
two fp16 pieces each (four products) without leaving the error class of the fp32-MFMA chain?   python tools/h2_sim_qk.py [scale]"""
import sys
import numpy as np
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
rng = np.random.default_rng(1)
NQ, NK, D = 512, 4096, 16
q = (rng.standard_normal((NQ, D)) * scale * 1.4426950408889634 / 4).astype(np.float32)      # pre-scaled, as the split pass does
k = (rng.standard_normal((NK, D)) * scale).astype(np.float32)
truth = q.astype(np.float64) @ k.astype(np.float64).T


def f16(x):
    return x.astype(np.float16).astype(np.float32)


def bf16_trunc(x):
    return (x.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)


def pieces(x, n, cut):
    out, r = [], x.astype(np.float32).copy()
    for _ in range(n):
        p = cut(r.copy())
        out.append(p)
        r = (r - p).astype(np.float32)
    return out


def report(name, s):
    e = s.astype(np.float64) - truth
    print(f"{name:58s} rms abs err {np.sqrt((e**2).mean()):.3e}   max {np.abs(e).max():.3e}   (rms |S| {np.sqrt((truth**2).mean()):.2f})")


# fp32 chain, k-ordered fma: one rounding per d
acc = np.zeros((NQ, NK), np.float32)
for d in range(D):
    acc = (acc.astype(np.float64) + q[:, d:d + 1].astype(np.float64) * k[None, :, d].astype(np.float64)).astype(np.float32)
report("fp32 fma chain (16 roundings)", acc)
# 4-deep MFMA steps (v_mfma_f32_16x16x4_f32: exact inside a step? unknown -- one rounding per 4)
acc = np.zeros((NQ, NK), np.float32)
for d in range(0, D, 4):
    acc = (acc.astype(np.float64) + q[:, d:d + 4].astype(np.float64) @ k[:, d:d + 4].astype(np.float64).T).astype(np.float32)
report("fp32, one rounding per 4 d", acc)

qb, kb = pieces(q, 3, bf16_trunc), pieces(k, 3, bf16_trunc)
terms = [(0, 0), (1, 0), (0, 1), (2, 0), (1, 1), (0, 2)]
# at d 16 two terms share one MFMA: three roundings
acc = np.zeros((NQ, NK), np.float64)
out = np.zeros((NQ, NK), np.float32)
for t in range(0, 6, 2):
    part = sum(kb[a].astype(np.float64) @ qb[b].astype(np.float64).T for a, b in terms[t:t + 2]).T
    out = (out.astype(np.float64) + part).astype(np.float32)
report("bf16 x3, six products (three MFMAs)", out)

# fp16 pairs with balanced power-of-two scales: q * 2^a, k * 2^-a, a chosen so that both maxima sit near 2^14 / 2^? ...
def scaled_pairs(x, target_exp):
    amax = np.abs(x).max()
    s = 2.0 ** (target_exp - np.floor(np.log2(amax)))
    return pieces((x * np.float32(s)).astype(np.float32), 2, f16), s

for tq, tk in [(14, 14), (10, 10), (4, 4), (14, -6)]:
    (q0, q1), sq = scaled_pairs(q, tq)
    (k0, k1), sk = scaled_pairs(k, tk)
    # MFMA 1: k0 q0 + k1 q0, MFMA 2: k0 q1 + k1 q1 (two roundings), then unscale (exact)
    p1 = (q0.astype(np.float64) @ (k0.astype(np.float64) + k1.astype(np.float64)).T)
    p2 = (q1.astype(np.float64) @ (k0.astype(np.float64) + k1.astype(np.float64)).T)
    out = (p2.astype(np.float32).astype(np.float64) + p1).astype(np.float32)      # small term first
    report(f"fp16 pairs, 4 products, max |q'| 2^{tq} max |k'| 2^{tk}", out.astype(np.float64) / (sq * sk))
    out3 = ((q1.astype(np.float64) @ k0.astype(np.float64).T).astype(np.float32).astype(np.float64) + p1).astype(np.float32)
    report(f"fp16 pairs, 3 products (no q1 k1)", out3.astype(np.float64) / (sq * sk))
