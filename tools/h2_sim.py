"""Dev tool (round 4): numpy emulation of the attention forward's P.V accumulation in the candidate operand formats, against
float64 -- which piece products are needed for the error to stay in the class of the fp32-MFMA kernel's k-ordered chain.
   python tools/h2_sim.py [L] [scale]
Emulated per 32-key MFMA chunk: exact piece products, one fp32 rounding when the chunk joins the accumulator (the fp32
kernel: one rounding per 4 keys, its MFMA's contraction depth).  S and exp2 are computed ONCE in fp32 and shared by every
variant (their error is common to all of them): what differs is only how P and V enter the contraction."""
import sys
import numpy as np

L = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
qk_scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
NQ, D = 128, 16
rng = np.random.default_rng(0)
q = (rng.standard_normal((NQ, D)) * qk_scale).astype(np.float32)
k = (rng.standard_normal((L, D)) * qk_scale).astype(np.float32)
v = rng.standard_normal((L, D)).astype(np.float32)
v[:, 3] *= np.exp(rng.standard_normal(L) * 4).astype(np.float32)        # one channel with a wide dynamic range
v[:, 5] *= 1e-3

s = (q.astype(np.float64) @ k.astype(np.float64).T) * (1.4426950408889634 / 4.0)
s32 = s.astype(np.float32)                                 # the fp32 scores every variant starts from
m = s32[:, :64].max(axis=1, keepdims=True)                 # fixed reference: the first tile's maximum
truth_p = np.exp2(s32.astype(np.float64) - m)              # float64 softmax of the SAME fp32 scores
truth = (truth_p @ v.astype(np.float64)) / truth_p.sum(axis=1, keepdims=True)


def f16(x):
    with np.errstate(over="ignore"):
        return x.astype(np.float16).astype(np.float32)


def bf16_trunc(x):
    return (x.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)


def pieces_f16(x, n):
    out, r = [], x.astype(np.float32)
    for _ in range(n):
        p = f16(r)
        out.append(p)
        r = (r - p).astype(np.float32)
    return out


def pieces_bf16(x, n):
    out, r = [], x.astype(np.float32)
    for _ in range(n):
        p = bf16_trunc(r.copy())
        out.append(p)
        r = (r - p).astype(np.float32)
    return out


def accumulate(terms, chunk):
    """terms: list of (P piece [NQ, L], V piece [L, D]); fp32 accumulator, one rounding per (chunk, term), small terms first."""
    acc = np.zeros((NQ, D), np.float32)
    for c0 in range(0, L, chunk):
        for pp, vv in reversed(terms):
            part = pp[:, c0:c0 + chunk].astype(np.float64) @ vv[c0:c0 + chunk].astype(np.float64)
            acc = (acc.astype(np.float64) + part).astype(np.float32)
    return acc


def report(name, acc, l):
    o = acc.astype(np.float64) / l
    err = o - truth
    scale = np.abs(truth).max(axis=0, keepdims=True)
    rel = err / scale
    per_ch = np.sqrt((rel ** 2).mean(axis=0))
    print(f"{name:48s} rms {np.sqrt((rel**2).mean()):.3e}  max {np.abs(rel).max():.3e}   ch3(wide) {per_ch[3]:.2e}  ch5(small) {per_ch[5]:.2e}")


for shift in (0, 8):
    p32 = np.exp2((s32 - m) + np.float32(shift)).astype(np.float32)      # what v_exp_f32 hands over (taken as exact fp32 here)
    l = p32.astype(np.float64).sum(axis=1, keepdims=True)
    tp = np.exp2(s32.astype(np.float64) - m + shift)
    print(f"--- L={L}  q/k scale {qk_scale}  P shift 2^{shift}   (P range {p32.min():.2e} .. {p32.max():.2e})")
    if shift == 0:
        # fp32-MFMA kernel: 4 keys per MFMA
        report("fp32 chain (4 keys per rounding)", accumulate([(p32, v)], 4), l)
        pb, vb = pieces_bf16(p32, 3), pieces_bf16(v, 3)
        terms = [(pb[0], vb[0]), (pb[0], vb[1]), (pb[1], vb[0]), (pb[0], vb[2]), (pb[1], vb[1]), (pb[2], vb[0])]
        report("bf16 x3, six products", accumulate(terms, 32), l)
    vmax = np.abs(v).max(axis=0, keepdims=True)
    sv = np.exp2(14 - np.floor(np.log2(vmax)))             # per channel: max |v'| in [2^14, 2^15)
    ph = pieces_f16(p32, 2)
    for nv in (2, 3):
        vh = pieces_f16(v * sv, nv)
        unscale = lambda a: (a.astype(np.float64) / sv).astype(np.float64)
        t3 = [(ph[0], vh[0]), (ph[0], vh[1]), (ph[1], vh[0])]
        t4 = t3 + [(ph[1], vh[1])]
        report(f"fp16: P 2 pieces, V {nv} pieces, 3 products", unscale(accumulate(t3, 32)), l)
        report(f"fp16: P 2 pieces, V {nv} pieces, 4 products", unscale(accumulate(t4, 32)), l)
        if nv == 3:
            report(f"fp16: P 2 pieces, V 3 pieces, 5 products", unscale(accumulate(t4 + [(ph[0], vh[2])], 32)), l)
    p1 = pieces_f16(p32, 1)
    vh = pieces_f16(v * sv, 2)
    report("fp16: P ONE piece, V 2 pieces, 2 products (too few)", unscale(accumulate([(p1[0], vh[0]), (p1[0], vh[1])], 32)), l)
