"""Dev tool (round 6): what do the plan's side lanes (engine.FORK_MAX_PIXELS: the 1x1 shortcut beside a ResBlock's main path, the four transposed-conv
phases beside each other -- parallel branches of the captured hipGraph) buy at a small size?   python tools/fork_ab.py [size=64] [batch=1] [T=300]
Times GaussianDiffusionSampler.forward (T replays of the captured CFG denoising step) with the lanes off and on, alternating, in one process, and checks
that the two graphs leave the same bits."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, hdiff_amd
from hdiff_amd import engine as E
from hdiff_amd.DiffusionFreeGuidence.ModelCondition import UNet
from hdiff_amd.DiffusionFreeGuidence.DiffusionCondition import GaussianDiffusionSampler
size = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
T = int(sys.argv[3]) if len(sys.argv) > 3 else 300
dev = torch.device("cuda", 0)
ON = E.FORK_MAX_PIXELS
out = {}
for tag, cap in (("lanes off", 0), ("lanes on", ON), ("lanes off", 0), ("lanes on", ON)):
    E.FORK_MAX_PIXELS = cap
    torch.manual_seed(0)
    net = UNet(T=T, num_labels=10, ch=128, ch_mult=[1, 2, 2, 2], num_res_blocks=2, dropout=0.15).eval()
    with torch.no_grad():
        net.tail[2].weight.mul_(0.1)
    net = net.to(dev)
    samp = GaussianDiffusionSampler(net, 1e-4, 0.028, T, w=1.8).to(dev)
    g = torch.Generator().manual_seed(1)
    x_T = torch.randn(B, 3, size, size, generator=g).to(dev)
    labels = (torch.arange(B) % 2 + 1).to(dev)
    noise = torch.randn(T, B, 3, size, size, generator=g).to(dev)
    with torch.no_grad():
        samp(x_T, labels, noise_by_step=noise)            # builds the plan, captures, runs once
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        y = samp(x_T, labels, noise_by_step=noise)
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out.setdefault(tag, []).append(y.clone())
    print(f"{tag}: {dt / T * 1e3:.3f} ms per step ({T / dt:.1f} steps/s) at {size}x{size}, batch {B}", flush=True)
E.FORK_MAX_PIXELS = ON
print("bitwise equal (lanes off vs on):", torch.equal(out["lanes off"][0], out["lanes on"][0]), "| repeatable:", torch.equal(out["lanes on"][0], out["lanes on"][1]))
