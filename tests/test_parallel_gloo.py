"""N > 1 path on CPU: two gloo ranks rehearse the sharding and the one gradient exchange of data-parallel training."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import hdiff_amd
from hdiff_amd import parallel as P


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, l, w = P.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)                      # ranks start from different weights
    lin = torch.nn.Linear(5, 3)
    extra = torch.nn.Parameter(torch.ones(7))          # never receives a gradient on rank 1
    params = list(lin.parameters()) + [extra]
    P.broadcast_parameters_(params, src=0)
    w0 = lin.weight.detach().clone()
    x = torch.full((2, 5), float(rank + 1))
    loss = lin(x).sum() + (extra.sum() if rank == 0 else 0.0)
    loss.backward()
    local = [None if p.grad is None else p.grad.clone() for p in params]
    nbytes = P.allreduce_mean_grads_(params)
    lo, hi = P.shard_range(11, world, rank)
    t = P.max_over_ranks(1.0 + rank)
    q.put((rank, w0, local, [p.grad.clone() for p in params], nbytes, (lo, hi), t, P.rank_seed(5, rank)))
    dist.destroy_process_group()


def test_two_rank_gradient_exchange_and_sharding():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, w0a, loc0, g0, nb0, s0, t0, seed0), (r1, w0b, loc1, g1, nb1, s1, t1, seed1) = res
    assert torch.equal(w0a, w0b)                                       # broadcast made the replicas identical
    for a, b in zip(g0, g1):
        assert torch.equal(a, b)                                       # every rank ends with the same averaged gradient
    assert torch.allclose(g0[0], (loc0[0] + loc1[0]) / 2)
    assert torch.allclose(g0[2], torch.full((7,), 0.5))                # missing gradient on rank 1 counted as zero
    assert nb0 == nb1 and nb0 >= (15 + 3 + 7) * 4
    assert s0 == (0, 6) and s1 == (6, 11)                              # contiguous, balanced, exhaustive
    assert t0 == t1 == 2.0                                             # max over ranks
    assert seed0 != seed1
