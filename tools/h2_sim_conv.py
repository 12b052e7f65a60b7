"""Dev tool (round 4): error of a 3x3 convolution's contraction (n = 9 * Cin products per output) in candidate operand formats
against float64 -- fp32 chain as the fp32 MFMA forms it, bf16 triples (six products per 16-channel chunk and tap, each rounded
into the fp32 accumulator), fp16 pairs of range-scaled operands (three or four products).   python tools/h2_sim_conv.py"""
import numpy as np
rng = np.random.default_rng(0)
def f16(x): return x.astype(np.float16).astype(np.float32)
def bf16_trunc(x): return (x.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)
def pieces(x, n, cut):
    out, r = [], x.astype(np.float32).copy()
    for _ in range(n):
        p = cut(r.copy()); out.append(p); r = (r - p).astype(np.float32)
    return out
def scale_to(x, e):  # power of two putting max|x| in [2^e, 2^(e+1))
    return np.float32(2.0 ** (e - np.floor(np.log2(np.abs(x).max()))))
for n, kind in [(1152, "swish"), (4608, "swish"), (1152, "gauss")]:
    M, N = 256, 256
    y = rng.standard_normal((n, N)).astype(np.float32) * 1.5
    x = (y / (1 + np.exp(-y))).astype(np.float32) if kind == "swish" else y
    w = (rng.standard_normal((M, n)) / np.sqrt(n)).astype(np.float32)
    truth = w.astype(np.float64) @ x.astype(np.float64)
    def rep(name, out):
        e = out.astype(np.float64) - truth
        print(f"n={n} {kind:5s} {name:44s} rms {np.sqrt((e**2).mean()):.3e} max {np.abs(e).max():.3e} (rms y {np.sqrt((truth**2).mean()):.2f})")
    # fp32 chain, one rounding per 2 (v_mfma_f32_32x32x2)
    acc = np.zeros((M, N), np.float32)
    for k in range(0, n, 2):
        acc = (acc.astype(np.float64) + w[:, k:k+2].astype(np.float64) @ x[k:k+2].astype(np.float64)).astype(np.float32)
    rep("fp32 chain (round per 2)", acc)
    # bf16x3: per 16-chunk, 6 products each rounded into fp32 acc
    wb, xb = pieces(w, 3, bf16_trunc), pieces(x, 3, bf16_trunc)
    terms = [(0,0),(1,0),(0,1),(2,0),(1,1),(0,2)]
    acc = np.zeros((M, N), np.float32)
    for k in range(0, n, 16):
        for a, b in terms:
            acc = (acc.astype(np.float64) + wb[a][:, k:k+16].astype(np.float64) @ xb[b][k:k+16].astype(np.float64)).astype(np.float32)
    rep("bf16x3 six products, round per 16-chunk product", acc)
    sw, sx = scale_to(w, 14), scale_to(x, 14)
    w0, w1 = pieces(w * sw, 2, f16); x0, x1 = pieces(x * sx, 2, f16)
    for prods, nm in [([(1,0),(0,1),(0,0)], "fp16 pairs 3 products"), ([(1,1),(1,0),(0,1),(0,0)], "fp16 pairs 4 products")]:
        acc = np.zeros((M, N), np.float32)
        ws, xs = [w0, w1], [x0, x1]
        for k in range(0, n, 16):
            for a, b in prods:
                acc = (acc.astype(np.float64) + ws[a][:, k:k+16].astype(np.float64) @ xs[b][k:k+16].astype(np.float64)).astype(np.float32)
        rep(nm, acc.astype(np.float64) / (float(sw) * float(sx)))
