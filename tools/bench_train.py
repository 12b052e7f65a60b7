"""Bench tool: time optimizer steps of the CFG-DDPM trainer on the HIP path (fwd + bwd [+ gradient exchange] + clip + AdamW).

  python tools/bench_train.py --size 256 --batch 8 --steps 3                        one GPU (config C3 with --batch 64)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \\
      tools/bench_train.py --size 256 --batch 64 --steps 3                          config C4: data-parallel, batch per GPU

Under torch.distributed.run every rank trains on its own synthetic batch (seed = base + rank); the step adds the ONE
collective of data-parallel training (hdiff_amd.parallel.FlatGradients: reduce-scatter + all-gather of the 190.8 MB flat
gradient buffer).  Timing: barrier + synchronize on both sides, max over ranks; rank 0 prints one JSON line
(samples/s = world * batch / time).  No 8-GPU run has been made from the build box (one GPU); the driver can run this."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import hdiff_amd
from hdiff_amd import parallel
from hdiff_amd.DiffusionFreeGuidence.ModelCondition import UNet
from hdiff_amd.DiffusionFreeGuidence.DiffusionCondition import GaussianDiffusionTrainer

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=256); ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--steps", type=int, default=3); ap.add_argument("--warmup", type=int, default=1)
ap.add_argument("--dropout", type=float, default=0.15)
ap.add_argument("--rehearse-one-gpu", action="store_true",
                help="dev: the data-parallel code path with every rank on cuda:0 over gloo (RCCL refuses two ranks on one device)")
a = ap.parse_args()
rank, local, world = parallel.init_from_env(backend="gloo" if a.rehearse_one_gpu else None)
if a.rehearse_one_gpu:
    local = 0
dev = torch.device("cuda", local if world > 1 else 0)
torch.cuda.set_device(dev)
torch.manual_seed(0)                                     # identical replicas
m = UNet(T=1000, num_labels=10, ch=128, ch_mult=[1, 2, 2, 2], num_res_blocks=2, dropout=a.dropout).to(dev).train()
parallel.broadcast_parameters_(m.parameters())
tr = GaussianDiffusionTrainer(m, 1e-4, 0.02, 1000).to(dev)
weights = list(m.parameters())
opt = torch.optim.AdamW(weights, lr=1e-4, weight_decay=1e-4)
flat = parallel.FlatGradients(weights, world, overlap=True) if world > 1 else None
g = torch.Generator().manual_seed(1 + rank)              # per-rank data
x0 = (torch.rand(a.batch, 3, a.size, a.size, generator=g) * 2 - 1).to(dev)
labels = (torch.arange(a.batch) % 2 + 1).to(dev)
torch.manual_seed(100 + rank)                            # per-rank t / noise / dropout streams
exchanged = 0
def step():
    global exchanged
    if flat is not None: flat.zero_()
    else: opt.zero_grad()
    loss = tr(x0, labels).sum() / a.batch ** 2.
    loss.backward()
    if flat is not None: exchanged = flat.exchange_mean_()
    torch.nn.utils.clip_grad_norm_(weights, 1.0)
    opt.step()
    return loss
for _ in range(a.warmup): step()
if world > 1: torch.distributed.barrier()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps): l = step()
torch.cuda.synchronize()
if world > 1: torch.distributed.barrier()
dt = parallel.max_over_ranks((time.perf_counter() - t0) / a.steps, None if a.rehearse_one_gpu else dev)
fwd = {64: 74.0, 128: 529.6, 256: 5857.4}[a.size]
if rank == 0:
    print(json.dumps({"size": a.size, "batch_per_gpu": a.batch, "n_gpus": world, "s_per_step": dt,
                      "samples_per_s": world * a.batch / dt, "fwd_equiv_tflops_per_gpu": 3 * fwd * a.batch / 1e3 / dt,
                      "loss": float(l), "max_mem_GB": torch.cuda.max_memory_allocated() / 1e9,
                      "gradient_exchange_bytes_per_rank": exchanged}))
if world > 1: torch.distributed.destroy_process_group()
