"""Dev tool: launch ONE 3x3 weight-gradient shape a few times (for rocprofv3 --pmc passes).  argv: B Cin Cout S [gn]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, hdiff_amd
from hdiff_amd import autograd as A, engine as E
B, cin, cout, S = [int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (4, 128, 128, 256))]
gn = len(sys.argv) > 5 and sys.argv[5] == "gn"
dev = "cuda:0"
x = torch.randn(B, cin, S, S, device=dev); dy = torch.randn(B, cout, S, S, device=dev)
g = (torch.rand(B, cin, device=dev) + 0.5, torch.randn(B, cin, device=dev)) if gn else None
dw = torch.empty(cout, cin, 3, 3, device=dev); taps = E.conv_taps(3, 1)
for _ in range(3):
    A._run_wgrad(x, None, g, dy, taps, cout, cin, B=B, H=S, W=S, VH=S, VW=S, targets=[(dw, 0, taps.ky, taps.kx, 0)])
torch.cuda.synchronize(); print("done")
