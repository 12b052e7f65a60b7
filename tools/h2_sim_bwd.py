"""Dev tool (round 4, groundwork for the next round): can dS = P o (dP - delta) of the attention backward be carried as fp16 PAIRS?
dS feeds dK = Q^T dS (sum over queries) and dQ = dS K (sum over keys).  Emulates, against float64, for one head (d = 16):
  triples   dS as three bf16 pieces, Q / K as three bf16 pieces, six products each (the kernel of rounds 3-4)
  pairs-t   dS as an fp16 pair under ONE power of two per tensor (max |dS| -> 2^14), Q / K as range-scaled pairs, three products
  pairs-t/2^n  the same with a power of two from a bound that overestimates max |dS| by 2^n (what a kernel has before it has seen dS)
  pairs-q   dS as an fp16 pair under one power of two PER QUERY (that row's max |dS| -> 2^14), the row scale folded into Q for
            dK and taken out of dQ's row at the end
for near-uniform attention (random initialisation: P ~ 1 / L everywhere) and for peaked attention (logit scale 4).
   python tools/h2_sim_bwd.py"""
import numpy as np
rng = np.random.default_rng(0)
NQ, NK, D = 1024, 2048, 16


def f16(x): return x.astype(np.float16).astype(np.float32)
def bf16_trunc(x): return (x.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)


def pieces(x, n, cut):
    out, r = [], x.astype(np.float32).copy()
    for _ in range(n):
        p = cut(r.copy()); out.append(p); r = (r - p).astype(np.float32)
    return out


def p2(x, e=14):   # power of two that puts max |x| into [2^e, 2^(e+1))
    m = np.abs(x).max()
    return np.float32(2.0 ** (e - np.floor(np.log2(m)))) if m > 0 else np.float32(1)


def contract(a_pieces, b_pieces, terms):   # sum over terms of a[i] @ b[j], each product rounded into an fp32 accumulator
    acc = None
    for i, j in terms:
        t = a_pieces[i].astype(np.float64) @ b_pieces[j].astype(np.float64)
        acc = t.astype(np.float32) if acc is None else (acc.astype(np.float64) + t).astype(np.float32)
    return acc


T6 = [(2, 0), (1, 1), (0, 2), (1, 0), (0, 1), (0, 0)]
T3 = [(1, 0), (0, 1), (0, 0)]
for name, logit in (("near-uniform attention (logit scale 0.05)", 0.05), ("Gaussian logits (scale 1)", 1.0), ("peaked attention (logit scale 4)", 4.0)):
    q = rng.standard_normal((NQ, D)).astype(np.float32)
    k = rng.standard_normal((NK, D)).astype(np.float32)
    v = rng.standard_normal((NK, D)).astype(np.float32)
    do = rng.standard_normal((NQ, D)).astype(np.float32)
    s = (q.astype(np.float64) @ k.astype(np.float64).T) * logit / 4
    p = np.exp(s - s.max(1, keepdims=True)); p /= p.sum(1, keepdims=True)
    dp = do.astype(np.float64) @ v.astype(np.float64).T
    delta = (p * dp).sum(1, keepdims=True)
    ds64 = p * (dp - delta)
    dk64, dq64 = ds64.T @ q.astype(np.float64), ds64 @ k.astype(np.float64)
    ds = ds64.astype(np.float32)            # what the kernel holds in registers (its own fp32 rounding)
    print(f"== {name}: max P {p.max():.2e}, median row max {np.median(p.max(1)):.2e}")

    def rep(tag, dk, dq):
        ek, eq = dk.astype(np.float64) - dk64, dq.astype(np.float64) - dq64
        rk = np.abs(ek).max(1) / np.abs(dk64).max(1)          # per key: worst error relative to that key's own gradient
        print(f"   {tag:13s} dK rms {np.sqrt((ek**2).mean())/np.sqrt((dk64**2).mean()):.2e} worst/key-own-max {rk.max():.2e} (median {np.median(rk):.2e})"
              f" | dQ rms {np.sqrt((eq**2).mean())/np.sqrt((dq64**2).mean()):.2e} worst {np.abs(eq).max()/np.abs(dq64).max():.2e}")
    # fp32 chain reference point: plain fp32 matmul emulation (round per 4, as the fp32 MFMA does)
    def chain(a, b):
        acc = np.zeros((a.shape[0], b.shape[1]), np.float32)
        for i in range(0, a.shape[1], 4):
            acc = (acc.astype(np.float64) + a[:, i:i+4].astype(np.float64) @ b[i:i+4].astype(np.float64)).astype(np.float32)
        return acc
    rep("fp32", chain(ds.T.copy(), q), chain(ds, k))
    d3, q3, k3 = pieces(ds, 3, bf16_trunc), pieces(q, 3, bf16_trunc), pieces(k, 3, bf16_trunc)
    rep("triples", contract([x.T for x in d3], q3, T6), contract(d3, k3, T6))
    sq, sk = p2(q), p2(k)
    qh, kh = pieces(q * sq, 2, f16), pieces(k * sk, 2, f16)
    st = p2(ds)
    dt = pieces(ds * st, 2, f16)
    rep("pairs-t", contract([x.T for x in dt], qh, T3).astype(np.float64) / (float(st) * float(sq)),
        contract(dt, kh, T3).astype(np.float64) / (float(st) * float(sk)))
    for loose in (9, 14):          # the power of two from an a-priori BOUND that overestimates max |dS| by 2^loose
        dl = pieces(ds * (st / np.float32(2.0 ** loose)), 2, f16)
        rep(f"pairs-t/2^{loose}", contract([x.T for x in dl], qh, T3).astype(np.float64) / (float(st) / 2.0 ** loose * float(sq)),
            contract(dl, kh, T3).astype(np.float64) / (float(st) / 2.0 ** loose * float(sk)))
    rowmax = np.abs(ds).max(1, keepdims=True)
    sr = (2.0 ** (14 - np.floor(np.log2(np.maximum(rowmax, 1e-300))))).astype(np.float32)      # per query
    dr = pieces(ds * sr, 2, f16)
    # dK = sum_q (q_row / s_row) * (ds_row * s_row): the row scale goes into a second copy of Q (exact: powers of two)
    qrs = q / sr
    sq2 = p2(qrs)
    qh2 = pieces(qrs * sq2, 2, f16)
    rep("pairs-q", contract([x.T for x in dr], qh2, T3).astype(np.float64) / float(sq2),
        contract(dr, kh, T3).astype(np.float64) / (sr.astype(np.float64) * float(sk)))
