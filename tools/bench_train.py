"""Dev/bench tool: time optimizer steps of the CFG-DDPM trainer on the HIP path (fwd + bwd + clip + AdamW).
  python tools/bench_train.py --size 256 --batch 8 --steps 3"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import hdiff_amd
from hdiff_amd.DiffusionFreeGuidence.ModelCondition import UNet
from hdiff_amd.DiffusionFreeGuidence.DiffusionCondition import GaussianDiffusionTrainer

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=256); ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--steps", type=int, default=3); ap.add_argument("--warmup", type=int, default=1)
ap.add_argument("--dropout", type=float, default=0.15)
a = ap.parse_args()
dev = "cuda:0"
torch.manual_seed(0)
m = UNet(T=1000, num_labels=10, ch=128, ch_mult=[1, 2, 2, 2], num_res_blocks=2, dropout=a.dropout).to(dev).train()
tr = GaussianDiffusionTrainer(m, 1e-4, 0.02, 1000).to(dev)
opt = torch.optim.AdamW(m.parameters(), lr=1e-4, weight_decay=1e-4)
g = torch.Generator().manual_seed(1)
x0 = (torch.rand(a.batch, 3, a.size, a.size, generator=g) * 2 - 1).to(dev)
labels = (torch.arange(a.batch) % 2 + 1).to(dev)
def step():
    opt.zero_grad()
    loss = tr(x0, labels).sum() / a.batch ** 2.
    loss.backward()
    torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
    opt.step()
    return loss
for _ in range(a.warmup): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps): l = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
fwd = {64: 74.0, 128: 529.6, 256: 5857.4}[a.size]
print(json.dumps({"size": a.size, "batch": a.batch, "s_per_step": dt, "samples_per_s": a.batch / dt,
                  "fwd_equiv_tflops": 3 * fwd * a.batch / 1e3 / dt, "loss": float(l),
                  "max_mem_GB": torch.cuda.max_memory_allocated() / 1e9}))
