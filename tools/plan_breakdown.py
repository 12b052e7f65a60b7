"""Dev tool: time every launch of one UNet forward plan on its own (HIP events around repeated launches) and print the conv
launches by shape with their TFLOP/s.   python3 tools/plan_breakdown.py [size=256] [batch=16] [reps=5]"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hdiff_amd  # noqa: F401
from hdiff_amd.DiffusionFreeGuidence.ModelCondition import UNet

size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device('cuda', 0)
torch.manual_seed(0)
net = UNet(T=1000, num_labels=10, ch=128, ch_mult=[1, 2, 2, 2], num_res_blocks=2, dropout=0.15).to(dev).eval()
up = net.plan_for(B, size, size, dev)
up.x.copy_(torch.randn(B, 3, size, size, device=dev))
up.t.fill_(500)
up.labels.fill_(1)
plan = up.plan
plan.run()
torch.cuda.synchronize()
s = torch.cuda.current_stream().cuda_stream
rows = collections.OrderedDict()
total = 0.0
for name, fn, args in plan.ops:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(*args, s)
    e0.record()
    for _ in range(reps):
        fn(*args, s)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    total += ms
    key, flops = name, 0.0
    if name == 'hdiff_conv2d_fwd':
        d = args[0]._obj
        cin = d.C0 + d.C1
        key = (f"conv {cin:4d}->{d.Cout:4d} taps {d.ntaps:2d} {d.VH}x{d.VW} s{d.in_stride}"
               f"{' gn' if d.gn_scale else ''}{' res' if d.residual else ''}{' cat' if d.C1 else ''}{' splitk' if d.splitk_floats else ''}")
        flops = 2.0 * d.ntaps * cin * d.Cout * d.VH * d.VW * d.B
    r = rows.setdefault(key, [0, 0.0, 0.0])
    r[0] += 1
    r[1] += ms
    r[2] += flops
print(f"plan {size}x{size} batch {B}: {len(plan.ops)} launches, sum of stand-alone times {total:.1f} ms")
for key, (n, ms, fl) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    tf = f"{fl / ms / 1e9:7.1f} TFLOP/s" if fl else ""
    print(f"{key:60s} x{n:3d} {ms:9.3f} ms {100 * ms / total:5.1f}%  {tf}")
