"""Throughput of the second tree's DDIM sampler (DynamicUNet, image-conditioned) on one MI355X.

    python tools/bench_ddim.py [--size 256] [--batch 8] [--ddim-step 100] [--reps 2]

One "step" = one DDIM iteration of diffusion/Diffusion.py:248-263 for the whole batch = one DynamicUNet forward + the fused
update.  Prints one JSON line (same field names as bench.py where they apply)."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import hdiff_amd
from hdiff_amd.diffusion.Model import DynamicUNet
from hdiff_amd.diffusion.Diffusion import GaussianDiffusionSampler

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=256)
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--ddim-step", type=int, default=100)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--contract", choices=["f32", "bf16x3"], default="bf16x3", help="the library's default mode is bf16x3")
a = ap.parse_args()
hdiff_amd.set_contraction_mode(a.contract)
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = DynamicUNet(T=1000, ch=128, ch_mult=[1, 2, 2, 2], num_res_blocks=2, dropout=0.0).eval().to(dev)
samp = GaussianDiffusionSampler(model, 1e-4, 0.02, 1000).to(dev)
img = torch.randint(0, 256, (a.batch, 3, a.size, a.size), generator=torch.Generator().manual_seed(1)).float().to(dev)
with torch.no_grad():
    samp(img, ddim=True, ddim_step=a.ddim_step)            # warm-up: builds the plan, packs weights, captures the graph
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        out = samp(img, ddim=True, ddim_step=a.ddim_step)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.reps
n_steps = len(range(0, 1000, int(1000 / a.ddim_step)))
assert torch.isfinite(out).all()
step_tflop = model.plan_for(a.batch, a.size, a.size, dev, True).plan.flops / 1e12
print(json.dumps({"metric": f"DDIM denoising-steps/sec ({a.size}x{a.size}, {n_steps} steps)", "value": n_steps / dt,
                  "unit": "denoising-steps/s", "ms_per_step": dt / n_steps * 1e3, "images_per_s": a.batch / dt,
                  "algorithmic_tflop_per_step": step_tflop, "whole_step_tflops": step_tflop / (dt / n_steps),
                  "config": {"workload": f"image-conditioned DDIM sampling, {a.size}x{a.size}, batch {a.batch}, DynamicUNet ch=128 "
                                         "ch_mult=[1,2,2,2] num_res_blocks=2 (43.2 M params), random-init weights, hipGraph replay",
                             "attention_contract": a.contract}}))
