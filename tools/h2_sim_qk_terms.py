"""Dev tool (round 5): S = Q K^T on fp16 pairs with a balance PER TERM (numpy emulation against float64).
Round 4 rejected fp16 pairs for the scores (tools/h2_sim_qk_balanced.py): S is an exponent, it must come out unscaled, so ONE
scale pair q' = q 2^a, k' = k 2^-a cannot keep both second pieces above fp16's 2^-3 full-precision floor.  But every product
term is its own set of MFMA contraction slots and can carry its own balance:
    k = k0 + k1, q = q0 + q1 (fp16 roundings);   S ~ k0 q0  +  (k0 2^-c)(q1 2^c)  +  (k1 2^c)(q0 2^-c)
the second pieces are stored scaled UP by 2^c (always normal numbers), their partners scaled DOWN, where fp16's absolute floor
(2^-25) is multiplied by a factor 2^(c-11) |x|: harmless.  Three products instead of the six of the bf16 triples, nothing to take
out of the exponent afterwards.   python tools/h2_sim_qk_terms.py [scale] [gauss|spike|spike1000|ramp|smallq|tinyk] [c]"""
import sys, numpy as np
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
tail = sys.argv[2] if len(sys.argv) > 2 else "gauss"
c = int(sys.argv[3]) if len(sys.argv) > 3 else 8
rng = np.random.default_rng(1)
NQ, NK, D = 512, 4096, 16
q = (rng.standard_normal((NQ, D)) * scale * 1.4426950408889634 / 4).astype(np.float32)
k = (rng.standard_normal((NK, D)) * scale).astype(np.float32)
if tail == "spike": k[100] *= 25; k[NK - 3] *= 40
if tail == "spike1000": k[100] *= 1000
if tail == "ramp": k *= np.linspace(0.3, 5, NK, dtype=np.float32)[:, None]
if tail == "smallq": q *= np.float32(2.0 ** -10); k *= np.float32(2.0 ** 10)
if tail == "tinyk": k[:, ::2] *= np.float32(2.0 ** -9)          # half of the channels nine binades below the rest
truth = q.astype(np.float64) @ k.astype(np.float64).T
f16 = lambda x: x.astype(np.float16).astype(np.float32)
def bf16_trunc(x): return (x.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)
def pieces(x, n, cut):
    out, r = [], x.astype(np.float32).copy()
    for _ in range(n):
        p = cut(r.copy()); out.append(p); r = (r - p).astype(np.float32)
    return out
def report(name, s):
    e = s.astype(np.float64) - truth
    print(f"{name:64s} rms {np.sqrt((e**2).mean()):.3e} max {np.abs(e).max():.3e} median|e| {np.median(np.abs(e)):.3e}")
print(f"scale {scale} {tail}: rms |S| {np.sqrt((truth**2).mean()):.2f}, max |q| {np.abs(q).max():.3g}, max |k| {np.abs(k).max():.3g}")
acc = np.zeros((NQ, NK), np.float32)
for d in range(D):
    acc = (acc.astype(np.float64) + q[:, d:d + 1].astype(np.float64) * k[None, :, d].astype(np.float64)).astype(np.float32)
report("fp32 fma chain (the fp32-MFMA kernel)", acc)
qb, kb = pieces(q, 3, bf16_trunc), pieces(k, 3, bf16_trunc)
terms = [(0, 0), (1, 0), (0, 1), (2, 0), (1, 1), (0, 2)]
out = np.zeros((NQ, NK), np.float32)
for t in range(0, 6, 2):
    part = sum(kb[a].astype(np.float64) @ qb[b].astype(np.float64).T for a, b in terms[t:t + 2]).T
    out = (out.astype(np.float64) + part).astype(np.float32)
report("bf16 triples, six products (the shipped kernel)", out)
# per-head balance: bring max |k| 2^a and max |q| 2^-a to the same binade (both far inside fp16's range)
ek, eq = np.floor(np.log2(np.abs(k).max())), np.floor(np.log2(np.abs(q).max()))
a = np.floor((eq - ek) / 2)
ks, qs = (k * np.float32(2.0 ** a)).astype(np.float32), (q * np.float32(2.0 ** -a)).astype(np.float32)
for cc in sorted({c, 11}):
    up, dn = np.float32(2.0 ** cc), np.float32(2.0 ** -cc)
    k0 = f16(ks); k1s = f16(((ks - k0) * up).astype(np.float32)); k0s = f16((k0 * dn).astype(np.float32))
    q0 = f16(qs); q1s = f16(((qs - q0) * up).astype(np.float32)); q0s = f16((q0 * dn).astype(np.float32))
    d64 = lambda x: x.astype(np.float64)
    m1 = d64(q0) @ d64(k0).T + d64(q1s) @ d64(k0s).T           # one 16x16x32 MFMA: terms (k0, q0) and (k0 2^-c, q1 2^c)
    out = m1.astype(np.float32)
    out = (out.astype(np.float64) + d64(q0s) @ d64(k1s).T).astype(np.float32)      # second MFMA: (k1 2^c, q0 2^-c)
    report(f"fp16 pairs, three products, balance per term c = {cc} (a = {int(a)})", out)
    out4 = (out.astype(np.float64) + (d64(f16(((qs - q0) * np.float32(1.0)).astype(np.float32))) @ d64(f16((ks - k0).astype(np.float32))).T)).astype(np.float32)
    report(f"   + the fourth product k1 q1 (unscaled pieces)", out4)
