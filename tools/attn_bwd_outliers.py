"""Dev tool (round 5): where the WORST elements of the fp16-pair attention backward sit -- the ten largest errors of dQ / dK / dV against
float64 beside the fp32-input kernel's at the same elements, and the tail of the error distribution (multiples of the rms).
   python3 tools/attn_bwd_outliers.py [d = 16] [L = 2048] [B = 2] [case = plain]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, hdiff_amd
import _attn_bwd_cases as K
lib = hdiff_amd.lib()
d = int(sys.argv[1]) if len(sys.argv) > 1 else 16
L = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
B = int(sys.argv[3]) if len(sys.argv) > 3 else 2
case = sys.argv[4] if len(sys.argv) > 4 else "plain"
heads = 8; Cc = heads * d
g = torch.Generator().manual_seed(7 + d)
qkv, d_o = K.make_case(case, d, L, B, heads, g)
qkv, d_o = qkv.to(K.DEV), d_o.to(K.DEV)
g32, gh2 = K.run_bwd(lib, qkv, d_o, heads, 0), K.run_bwd(lib, qkv, d_o, heads, 1)
for i, n in enumerate(("dQ", "dK", "dV")):
    e2s, e0s, refs = [], [], []
    for b in range(B):
        for h in range(heads):
            ref = K.ref64(qkv, d_o, b, h, d, Cc)[n]
            rows = slice(i * Cc + h * d, i * Cc + (h + 1) * d)
            e2s.append(gh2[b, rows].double() - ref); e0s.append(g32[b, rows].double() - ref); refs.append(ref)
    e2, e0, ref = torch.stack(e2s), torch.stack(e0s), torch.stack(refs)      # [pair][d][L]
    r2, r0 = e2.pow(2).mean().sqrt().item(), e0.pow(2).mean().sqrt().item()
    print(f"{n}: rms pairs {r2:.3e} fp32 {r0:.3e};  |err| > 4 rms: pairs {(e2.abs() > 4 * r2).sum().item()} fp32 {(e0.abs() > 4 * r0).sum().item()};"
          f"  > 6 rms: {(e2.abs() > 6 * r2).sum().item()} / {(e0.abs() > 6 * r0).sum().item()}  of {e2.numel()}")
    top = e2.abs().flatten().topk(10).indices
    for t in top.tolist():
        p, rem = divmod(t, d * L); dd, l = divmod(rem, L)
        print(f"   pair {p:2d} d {dd:2d} pos {l:5d}: ref {ref[p, dd, l].item(): .4e}  err pairs {e2[p, dd, l].item(): .3e}  fp32 {e0[p, dd, l].item(): .3e}"
              f"   row rms err pairs {e2[p, :, l].pow(2).mean().sqrt().item():.2e} fp32 {e0[p, :, l].pow(2).mean().sqrt().item():.2e}")
