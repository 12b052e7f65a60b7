"""Dev tool: launch one 3x3 convolution of the hot path a few times (for rocprofv3 --pmc passes / timing)."""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, hdiff_amd
from hdiff_amd import engine as E
B, Cin, Cout, S, k = [int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (4, 128, 128, 256, 3))]
gn = len(sys.argv) > 6 and sys.argv[6] in ("gn", "pairs")
pairs = len(sys.argv) > 6 and sys.argv[6] == "pairs"       # real GroupNorm statistics: the fp16-pair kernel may run (bf16x3 mode)
dev = "cuda:0"
x = torch.randn(B, Cin, S, S, device=dev)
w = torch.randn(Cout, Cin, k, k, device=dev) / math.sqrt(Cin * k * k)
b = torch.randn(Cout, device=dev)
plan = E.Plan(dev)
pk = E._std_pack(plan, w, k, k // 2)
out = plan.buf(B, Cout, S, S)
g = (torch.rand(B, Cin, device=dev) + 0.5, torch.randn(B, Cin, device=dev)) if gn else None
if pairs:
    gamma, beta = torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev) * 0.3
    pre = E.Plan(dev)
    g = pre.gn_scale_shift(x, None, gamma, beta, B, S * S)
    pre.run(); torch.cuda.synchronize()
    plan._gn_src[id(g[0])] = (gamma, beta, (Cin // 32) * S * S, 1.0)
plan.conv(x, None, pk, b, out, B=B, H=S, W=S, VH=S, VW=S, gn=g)
plan.pack_weights()
for _ in range(2): plan.run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): plan.run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"conv {Cin}->{Cout} k{k} @{S} B={B} gn={gn}: {ms:.3f} ms  {2.0*k*k*Cin*Cout*S*S*B/ms/1e9:.1f} TFLOP/s")
