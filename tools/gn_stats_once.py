"""Dev tool: launch hdiff_gn_stats (+ finalize) on ONE tensor of the bench's shape a few times, each time on a different
tensor of a rotation larger than the Infinity Cache (cold HBM reads), for rocprofv3 --kernel-trace / --pmc passes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, hdiff_amd
from hdiff_amd import engine as E
B, Cc, S = [int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (16, 128, 256))]
dev = "cuda:0"
xs = [torch.randn(B, Cc, S, S, device=dev) for _ in range(4)]          # 4 x 537 MB at the default shape
gamma, beta = torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev)
plan = E.Plan(dev)
for x in xs:
    plan.gn_scale_shift(x, None, gamma, beta, B, S * S)
for _ in range(3):
    plan.run()
torch.cuda.synchronize(); print("done")
