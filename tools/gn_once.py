"""Dev tool: achieved bytes/s of the HBM-bound kernels (GroupNorm statistics, DDPM update) at the bench's shapes."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, hdiff_amd
from hdiff_amd import engine as E
dev = "cuda:0"


def timeit(plan, reps=20):
    for _ in range(3): plan.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): plan.run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for (B, Cc, S) in [(16, 128, 256), (16, 256, 128), (16, 384, 256), (16, 512, 128)]:
    # a tensor larger than the 256 MB Infinity Cache is rotated so that every pass streams from HBM
    xs = [torch.randn(B, Cc, S, S, device=dev) for _ in range(max(2, int(1.2e9 // (B * Cc * S * S * 4)) + 1))]
    gamma, beta = torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev)
    plan = E.Plan(dev)
    for x in xs:
        plan.gn_scale_shift(x, None, gamma, beta, B, S * S)
    ms = timeit(plan) / len(xs)
    nbytes = B * Cc * S * S * 4
    print(f"gn_stats+finalize [{B},{Cc},{S},{S}]: {ms*1e3:.1f} us per tensor, {nbytes/ms/1e6:.0f} GB/s of activation read ({nbytes/1e6:.0f} MB, cold)")

n = 16 * 3 * 256 * 256
bufs = [torch.randn(n, device=dev) for _ in range(5)]
c = torch.ones(1000, device=dev)
step = torch.full((1,), 500, dtype=torch.int32, device=dev)
flag = torch.zeros(1, dtype=torch.int32, device=dev)
plan = E.Plan(dev)
plan.call("hdiff_ddpm_step", bufs[0].data_ptr(), bufs[1].data_ptr(), bufs[2].data_ptr(), bufs[3].data_ptr(), bufs[4].data_ptr(),
          c.data_ptr(), c.data_ptr(), c.data_ptr(), step.data_ptr(), 1000, C.c_double(1.8), C.c_uint64(0), flag.data_ptr(), n)
ms = timeit(plan)
print(f"ddpm_step {n} elements: {ms*1e3:.1f} us, {5*n*4/ms/1e6:.0f} GB/s over its 5 tensors")
