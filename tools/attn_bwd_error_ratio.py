"""Dev tool (round 5): error class of the split-operand attention BACKWARD -- rms and worst error of dQ, dK, dV against float64 as
ratios to the fp32-input kernel's, over all (sample, head) pairs and for the worst pair, for plain and adversarial inputs
(tests/_attn_bwd_cases.py; the gates of tests/test_gpu_backward.py come from this table).
   python3 tools/attn_bwd_error_ratio.py [d = 16] [L = 2048] [B = 2] [seed offset = 0] [case = all]     (HDIFF_LIB selects a variant library, e.g. a mutant)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, hdiff_amd
import _attn_bwd_cases as K
lib = hdiff_amd.lib()
d = int(sys.argv[1]) if len(sys.argv) > 1 else 16
L = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
B = int(sys.argv[3]) if len(sys.argv) > 3 else 2
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 0
only = sys.argv[5] if len(sys.argv) > 5 else None
heads = 8
for name in K.CASES:
    if only and name != only: continue
    g = torch.Generator().manual_seed(7 + d + 1000 * seed)
    qkv, d_o = K.make_case(name, d, L, B, heads, g)
    st = K.error_stats(lib, qkv.to(K.DEV), d_o.to(K.DEV), heads)
    row = []
    for n, s in st.items():
        pr = max(p[2] / p[3] for p in s["pair"]); pw = max(p[4] / p[5] for p in s["pair"])
        row.append(f"{n}: all pairs rms x{s['rms'][0] / s['rms'][1]:.2f} worst x{s['worst'][0] / s['worst'][1]:.2f} (worst / mag {s['worst'][0] / s['mag']:.1e}); worst pair rms x{pr:.2f} worst x{pw:.2f}")
    print(f"d {d} L {L} B {B} seed {seed} {name}: " + " | ".join(row), flush=True)
