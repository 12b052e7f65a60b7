"""Data-parallel training (BASELINE config C4's code path) with two ranks on the one GPU of the test box: replicated weights,
per-rank batches, gradients accumulated by the HIP backward straight into FlatGradients' views, ONE reduce-scatter + all-gather
per step, clip, AdamW.  The process group is gloo (RCCL refuses two ranks on one device); the collectives are the same calls
the RCCL run makes (parallel._exchange_mean_).  Asserts: replicas stay bitwise identical over two steps, and the exchanged
gradient is the mean of the two ranks' own gradients."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    import hdiff_amd  # noqa: F401
    from hdiff_amd import parallel as P
    from hdiff_amd.DiffusionFreeGuidence.DiffusionCondition import GaussianDiffusionTrainer
    from hdiff_amd.DiffusionFreeGuidence.ModelCondition import UNet
    P.init_from_env(backend="gloo")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    torch.manual_seed(100 + rank)                                   # ranks start from DIFFERENT weights ...
    net = UNet(T=8, num_labels=3, ch=32, ch_mult=[1, 2], num_res_blocks=1, dropout=0.0).to(dev).train()
    P.broadcast_parameters_(net.parameters())                       # ... and are made replicas of rank 0
    tr = GaussianDiffusionTrainer(net, 1e-4, 0.02, 8).to(dev)
    weights = list(net.parameters())
    opt = torch.optim.AdamW(weights, lr=1e-3, weight_decay=1e-4)
    flat = P.FlatGradients(weights, world, overlap=True, bucket_bytes=1 << 20)      # ~4 buckets, sent from the gradient hooks
    assert len(flat.buckets) >= 3
    g = torch.Generator().manual_seed(7 + rank)                     # per-rank data
    x0 = (torch.rand(2, 3, 16, 16, generator=g) * 2 - 1).to(dev)
    labels = torch.tensor([1, 2 + rank], device=dev)
    t = torch.tensor([3, 5 - rank], device=dev)
    noise = torch.randn(2, 3, 16, 16, generator=g).to(dev)
    checks = []
    for step in range(2):
        flat.zero_()
        loss = tr(x0, labels, t=t, noise=noise).sum() / 2 ** 2.
        loss.backward()
        assert flat._next >= 1, "no bucket left during backward"
        mine = flat.flat.clone()                                     # (the reduce-scatter writes the shards, not the buffer)
        sent = flat.exchange_mean_()
        both = [torch.empty_like(mine.cpu()) for _ in range(world)]
        dist.all_gather(both, mine.cpu())
        want = (both[0] + both[1]) / 2
        err = (flat.flat.cpu() - want).abs().max().item()
        torch.nn.utils.clip_grad_norm_(weights, 1.0)
        opt.step()
        checks.append((err, want.abs().max().item(), sent))
    digest = torch.cat([p.detach().reshape(-1) for p in weights]).cpu()
    q.put((rank, checks, digest.numpy().copy()))
    dist.destroy_process_group()


def test_two_ranks_train_identical_replicas_on_one_gpu():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    (_, c0, w0), (_, c1, w1) = res
    for (err, mag, sent) in c0 + c1:
        assert err <= 1e-6 * mag + 1e-12, (err, mag)       # exchanged gradient = mean of the ranks' own gradients
        assert sent >= 4 * 800_000                            # the whole flat buffer (0.88 M parameters) went through the collective
    assert (w0 == w1).all(), "replicas diverged"


def _rccl_worker(port, q):
    """One rank, backend "nccl" (= RCCL): the training step's exchange with the REAL collective library -- asynchronous
    reduce-scatters started from the gradient hooks inside backward, all-gathers at the end.  The mean over one rank must
    give back exactly the gradients a plain backward produces; a collective that ran ahead of the kernels writing its
    bucket (wrong stream ordering) would return stale values."""
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    import hdiff_amd  # noqa: F401
    from hdiff_amd import parallel as P
    from hdiff_amd.DiffusionFreeGuidence.DiffusionCondition import GaussianDiffusionTrainer
    from hdiff_amd.DiffusionFreeGuidence.ModelCondition import UNet
    try:
        torch.cuda.set_device(0)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
        dev = torch.device("cuda", 0)
        torch.manual_seed(5)
        net = UNet(T=8, num_labels=3, ch=32, ch_mult=[1, 2], num_res_blocks=1, dropout=0.0).to(dev).train()
        tr = GaussianDiffusionTrainer(net, 1e-4, 0.02, 8).to(dev)
        weights = list(net.parameters())
        g = torch.Generator().manual_seed(11)
        x0 = (torch.rand(2, 3, 32, 32, generator=g) * 2 - 1).to(dev)
        labels, t = torch.tensor([1, 2], device=dev), torch.tensor([3, 5], device=dev)
        noise = torch.randn(2, 3, 32, 32, generator=g).to(dev)

        def loss():
            return tr(x0, labels, t=t, noise=noise).sum() / 2 ** 2.
        for p in weights:
            p.grad = None
        loss().backward()
        plain = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in weights]).clone()
        flat = P.FlatGradients(weights, 1, overlap=True, bucket_bytes=1 << 20, single_rank_collectives=True)
        out = []
        for step in range(2):
            flat.zero_()
            loss().backward()
            started = flat._next
            sent = flat.exchange_mean_()
            got = torch.cat([p.grad.reshape(-1) for p in weights])
            out.append((started, len(flat.buckets), sent, bool(torch.equal(got, plain)), float((got - plain).abs().max())))
        # what bench.py --gpus N does on every rank: capture the sampler step into a hipGraph and replay it while the
        # RCCL group (and its watchdog thread) is alive, a barrier and a MAX all-reduce around it
        from hdiff_amd.DiffusionFreeGuidence.DiffusionCondition import GaussianDiffusionSampler
        net.eval()
        sm = GaussianDiffusionSampler(net, 1e-4, 0.02, 8, w=1.8).to(dev)
        xT = torch.randn(2, 3, 32, 32, generator=g).to(dev)
        z = torch.randn(8, 2, 3, 32, 32, generator=g).to(dev)
        with torch.no_grad():
            sm.use_graph = False
            eager = sm(xT, labels, noise_by_step=z)
            sm.use_graph = True
            dist.barrier()
            graphed = sm(xT, labels, noise_by_step=z)
        tmax = torch.tensor([1.5], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        out.append(("sampler", bool(torch.equal(eager, graphed)), float(tmax.item())))
        q.put(("ok", dist.get_backend(), out))
        dist.destroy_process_group()
    except Exception as e:           # report instead of hanging the parent on the queue
        q.put(("error", repr(e), []))


def test_rccl_single_rank_runs_the_bucketed_exchange_on_the_gpu():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
    p.start()
    status, backend, out = q.get(timeout=300)
    p.join(120)
    assert status == "ok", backend
    assert backend == "nccl"
    tag, same_sample, tmax = out.pop()
    assert tag == "sampler" and same_sample and tmax == 1.5     # graph capture / replay beside the live RCCL group
    for started, nb, sent, same, err in out:
        assert nb >= 3 and started >= 1                      # buckets left from the hooks while backward was running
        assert sent >= 4 * 800_000
        assert same, f"RCCL exchange over one rank changed the gradients (max diff {err})"
