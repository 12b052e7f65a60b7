"""Bench tool: time optimizer steps of the CFG-DDPM trainer on the HIP path (fwd + bwd [+ gradient exchange] + clip + AdamW).

  python tools/bench_train.py --size 256 --batch 8 --steps 3                 one GPU (config C3 with --batch 64)
  python tools/bench_train.py --gpus 8 --size 256 --batch 64 --steps 3       config C4: data-parallel, batch per GPU --
                                                                             this process starts the 8 ranks itself
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \\
      tools/bench_train.py --gpus 8 --size 256 --batch 64 --steps 3          the same under a launcher

Every rank trains on its own synthetic batch (seed = base + rank); the step adds the ONE collective of data-parallel
training (hdiff_amd.parallel.FlatGradients: reduce-scatter + all-gather of the 190.8 MB flat gradient buffer).  Timing:
barrier + synchronize on both sides, max over ranks; rank 0 prints one JSON line (samples/s = world * batch / time).
The step itself is bench.train_steps -- the same code bench.py runs for its `configs.C3` entry.
No 8-GPU run has been made from the build box (one GPU); the driver can run this."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--gpus", type=int, default=None, help="ranks (one per GPU); > 1 without a launcher: started from here")
ap.add_argument("--size", type=int, default=256); ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--steps", type=int, default=3); ap.add_argument("--warmup", type=int, default=1)
ap.add_argument("--dropout", type=float, default=0.15)
ap.add_argument("--rehearse-one-gpu", action="store_true",
                help="dev: the data-parallel code path with every rank on cuda:0 over gloo (RCCL refuses two ranks on one device)")
a = ap.parse_args()
if a.gpus and a.gpus > 1 and "WORLD_SIZE" not in os.environ:      # plain start: become the launcher (the GPU is untouched so far)
    from hdiff_amd.parallel import launch_ranks
    sys.exit(launch_ranks(os.path.abspath(__file__), sys.argv[1:], a.gpus))
import bench
from hdiff_amd import parallel
rank, local, world = parallel.init_from_env(backend="gloo" if a.rehearse_one_gpu else None)
if a.gpus is not None and world != a.gpus:
    sys.exit(f"bench_train.py: --gpus {a.gpus} but WORLD_SIZE={world}")
if a.rehearse_one_gpu:
    local = 0
dev = torch.device("cuda", local if world > 1 else 0)
torch.cuda.set_device(dev)
res = bench.train_steps(a.size, a.batch, a.steps, a.warmup, a.dropout, dev, rank=rank, world=world, rehearse=a.rehearse_one_gpu)
if rank == 0:
    print(json.dumps(res), flush=True)
if world > 1: torch.distributed.destroy_process_group()
