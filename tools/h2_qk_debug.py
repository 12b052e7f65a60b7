"""Dev tool (round 5): error anatomy of the d_head 16 forward (fp16-pair scores) against float64 on a small case.
Runs the kernel on (a) random inputs, (b) inputs whose Q, K are exactly fp16-representable after the kernel's own balance
(second pieces vanish: only k0 q0 is exercised), and prints the error per head and per sample."""
import ctypes as C, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, hdiff_amd
lib = hdiff_amd.lib(); lib.hdiff_set_contraction_mode(1)
s = torch.cuda.current_stream().cuda_stream
def ref64(qkv, heads):
    B, C3, L = qkv.shape; Cc = C3 // 3; d = Cc // heads
    q, k, v = [z.reshape(B, heads, d, L).transpose(2, 3).double() for z in qkv.split(Cc, dim=1)]
    w = torch.softmax(q @ k.transpose(2, 3) / math.sqrt(d), dim=-1)
    return (w @ v).transpose(2, 3).reshape(B, Cc, L)
def run(qkv, heads=8, ws_on=True):
    B, C3, L = qkv.shape; Cc = C3 // 3
    need = C.c_int64(0); lib.hdiff_mha_flash_fwd_workspace(B, Cc, heads, L, C.byref(need))
    ws = torch.empty(need.value // 4 + 1, device="cuda")
    o = torch.empty(B, Cc, L, device="cuda")
    if ws_on: assert lib.hdiff_mha_flash_fwd_ws(qkv.data_ptr(), o.data_ptr(), None, B, Cc, heads, L, ws.data_ptr(), need.value, s) == 0
    else: assert lib.hdiff_mha_flash_fwd(qkv.data_ptr(), o.data_ptr(), None, B, Cc, heads, L, s) == 0
    torch.cuda.synchronize()
    return o
d, L, B = 16, int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 2
g = torch.Generator().manual_seed(100 + d + L)
qkv = torch.randn(B, 3 * 8 * d, L, generator=g).cuda()
for name, x in [("random", qkv), ("q, k exactly fp16 (x 2^-1 / 2^1 balance-safe)", torch.cat([qkv[:, :256].half().float(), qkv[:, 256:]], 1))]:
    r = ref64(x, 8)
    o = run(x); o_in = run(x, ws_on=False)
    e = (o.double() - r).abs(); e2 = (o_in.double() - r).abs()
    print(f"{name}: max err workspace kernel {e.max().item():.3e}, in-loop split kernel {e2.max().item():.3e}; per sample {[f'{v:.1e}' for v in e.amax(dim=(1,2)).tolist()]}")
    print("   per head:", [f"{e[:, h*d:(h+1)*d].max().item():.1e}" for h in range(8)])
    print("   per 256-query block (sample 0, all heads):", [f"{e[0, :, i:i+256].max().item():.1e}" for i in range(0, L, 256)][:8])
    print("   rms", (e**2).mean().sqrt().item(), "rms in-loop", (e2**2).mean().sqrt().item())

# ---- the operand pieces themselves: read them back from the workspace and compare with a numpy restatement of the split pass
import numpy as np
B, heads, Cc = 2, 8, 128
x = qkv
need = C.c_int64(0); lib.hdiff_mha_flash_fwd_workspace(B, Cc, heads, L, C.byref(need))
ws = torch.zeros(need.value // 2 + 1, device="cuda", dtype=torch.float16)
o = torch.empty(B, Cc, L, device="cuda")
assert lib.hdiff_mha_flash_fwd_ws(x.data_ptr(), o.data_ptr(), None, B, Cc, heads, L, ws.data_ptr(), need.value, s) == 0
torch.cuda.synchronize()
piece = L * d
w = ws[: B * heads * 9 * piece].view(B, heads, 9, L, d).float().cpu().numpy().astype(np.float64)
xq = x[:, :Cc].reshape(B, heads, d, L).transpose(2, 3).cpu().numpy()
xk = x[:, Cc:2 * Cc].reshape(B, heads, d, L).transpose(2, 3).cpu().numpy()
qscale = np.float32(1.4426950408889634 / 4.0)
for b in range(1):
    for h in range(2):
        mq = np.float32(np.abs(xq[b, h]).max()) * qscale; mk = np.float32(np.abs(xk[b, h]).max())
        eq, ek = int(np.frexp(mq)[1]) - 1 + 127, int(np.frexp(mk)[1]) - 1 + 127
        a = int((eq - ek) / 2)
        qs = (xq[b, h] * np.float32(qscale * np.float32(2.0 ** -a))).astype(np.float32); ks = (xk[b, h] * np.float32(2.0 ** a)).astype(np.float32)
        q0, q1s, k0, k0s, k1s, k1 = (w[b, h, i] for i in range(6))
        f16 = lambda t: t.astype(np.float16).astype(np.float64)
        print(f"b {b} head {h}: a = {a};  |q0 - f16(q')| {np.abs(q0 - f16(qs)).max():.2e}  |q1s - f16((q'-q0) 256)| {np.abs(q1s - f16((qs - q0.astype(np.float32)) * 256)).max():.2e}")
        print(f"     |k0 - f16(k')| {np.abs(k0 - f16(ks)).max():.2e}  |k0s - k0/256| {np.abs(k0s - k0 / 256).max():.2e}  |k1s - f16((k'-k0) 256)| {np.abs(k1s - f16((ks - k0.astype(np.float32)) * 256)).max():.2e}  |k1 - f16(k'-k0)| {np.abs(k1 - f16(ks - k0.astype(np.float32))).max():.2e}")
        S_true = (xq[b, h].astype(np.float64) * float(qscale)) @ xk[b, h].astype(np.float64).T
        S_pieces = q0 @ k0.T + q1s @ k0s.T + (q0 / 256) @ k1s.T + (q1s / 256) @ k1.T
        S_3 = q0 @ k0.T + q1s @ k0s.T
        print(f"     S from the four products of the pieces read back: max err {np.abs(S_pieces - S_true).max():.2e}; without the k1 products {np.abs(S_3 - S_true).max():.2e}")

# ---- -DH2_DIAG=2 builds: the kernel's own first score tile (wave 0, query tile 0, key tile 0) against the exact scores
dg = ws[: B * heads * 9 * piece].view(torch.float32).view(B, heads, 9 * piece // 2)
off = (8 * piece * 2 + 128) // 4
for b in range(1):
    for h in range(2):
        t = dg[b, h, off: off + 64 * 16].cpu().numpy().reshape(64, 4, 4)       # [lane][kt][r]
        if not np.any(t): print("(no score dump: not a -DH2_DIAG=2 build)"); break
        S_k = np.zeros((16, 64))
        for lane in range(64):
            i16, g = lane & 15, lane >> 4
            for kt in range(4):
                for r in range(4): S_k[i16, 16 * kt + 4 * g + r] = t[lane, kt, r]
        S_true = ((xq[b, h].astype(np.float64) * float(qscale)) @ xk[b, h].astype(np.float64).T)[:16, :64]
        q0, q1s, k0, k0s, k1s, k1 = (w[b, h, i] for i in range(6))
        S_3 = (q0 @ k0.T + q1s @ k0s.T)[:16, :64]; S_m1 = ((q0 / 256) @ k1s.T + (q1s / 256) @ k1.T)[:16, :64]
        print(f"b {b} head {h}: kernel S vs exact {np.abs(S_k - S_true).max():.2e}; vs MFMA 0 alone {np.abs(S_k - S_3).max():.2e}; vs MFMA 0 + MFMA 1 {np.abs(S_k - S_3 - S_m1).max():.2e}; vs MFMA 0 - MFMA 1 {np.abs(S_k - S_3 + S_m1).max():.2e}; |MFMA 1| max {np.abs(S_m1).max():.2e}")
        print("   kernel - MFMA0 (first row):", np.array2string((S_k - S_3)[0, :8], precision=2), " MFMA1 expected:", np.array2string(S_m1[0, :8], precision=2))
        # which (K piece, Q piece) products would explain what the kernel's second MFMA contributed?  (lo = d of lanes g 0,1; hi = g 2,3 -- both cover d 0..15)
        resid = S_k - S_3
        Ks = {"k0": k0, "k0s": k0s, "k1s": k1s, "k1": k1}; Qs = {"q0": q0, "q1s": q1s, "q0s": q0 / 256, "q1": q1s / 256}
        prods = {(kn, qn): (Qs[qn] @ Ks[kn].T)[:16, :64] for kn in Ks for qn in Qs}
        best = sorted(((np.abs(resid - prods[a] - prods[c]).max(), a, c) for a in prods for c in prods if a <= c), key=lambda z: z[0])[:4]
        for e, a, c in best: print(f"      residual ~ {a[0]} x {a[1]} + {c[0]} x {c[1]}: max dev {e:.2e}")
        # the operands lane (i16, g) holds: Q set 0 / 1 of query i16, K set 0 / 1 of key i16 (key tile 0), 8 fp16 each
        ops = dg[b, h, off + 64 * 16: off + 2 * 64 * 16].view(torch.int32).cpu().numpy().reshape(64, 16).astype(np.uint32)
        halves = np.stack([(ops & 0xffff).astype(np.uint16), (ops >> 16).astype(np.uint16)], -1).reshape(64, 32).view(np.float16).astype(np.float64).reshape(64, 4, 8)
        worst = [0.0] * 4
        for lane in range(64):
            i16, g = lane & 15, lane >> 4
            dsl = slice(8 * (g & 1), 8 * (g & 1) + 8); hi = g >> 1
            exp = [(q1s if hi else q0)[i16, dsl], ((q1s if hi else q0) / 256)[i16, dsl], (k0s if hi else k0)[i16, dsl], (k1 if hi else k1s)[i16, dsl]]
            for n in range(4): worst[n] = max(worst[n], np.abs(halves[lane, n] - exp[n]).max())
        print(f"      operands held by the lanes vs expected: Q set 0 {worst[0]:.2e}, Q set 1 {worst[1]:.2e}, K set 0 {worst[2]:.2e}, K set 1 {worst[3]:.2e}")
