"""Dev tool: time hdiff_conv2d_wgrad (+ unpack) for the 3x3 convolutions of the 256x256 training step.
  python tools/bench_wgrad.py [B]"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import hdiff_amd  # noqa: E402
from hdiff_amd import autograd as A, engine as E  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = "cuda:0"
for (cin, cout, S, gn, k) in [(128, 128, 256, True, 3), (384, 128, 256, True, 3), (256, 256, 128, True, 3), (512, 256, 64, True, 3),
                              (128, 128, 256, False, 3), (128, 384, 256, False, 1), (128, 128, 256, False, 1), (256, 768, 128, False, 1),
                              (512, 256, 64, False, 1)]:
    x = torch.randn(B, cin, S, S, device=dev)
    dy = torch.randn(B, cout, S, S, device=dev)
    g = (torch.rand(B, cin, device=dev) + 0.5, torch.randn(B, cin, device=dev)) if gn else None
    dw = torch.empty(cout, cin, k, k, device=dev)
    taps = E.conv_taps(k, k // 2)

    def run():
        A._run_wgrad(x, None, g, dy, taps, cout, cin, B=B, H=S, W=S, VH=S, VW=S, targets=[(dw, 0, taps.ky, taps.kx, 0)])
    for _ in range(5):          # warm-up (the first shape otherwise pays the clock ramp of an idle GPU)
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"wgrad {cin}->{cout} k{k} @{S} B={B} gn={gn}: {ms:.3f} ms  {2.0 * k * k * cin * cout * S * S * B / ms / 1e9:.1f} TFLOP/s", flush=True)
