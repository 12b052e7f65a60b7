"""N > 1 path on CPU: two gloo ranks rehearse the sharding and the one gradient exchange of data-parallel training."""
import os
import socket

import pytest

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


def _plain(o):
    """Tensors as numpy arrays: a tensor sent through the queue shares memory through a file descriptor that dies with the
    worker, which may exit before the parent has read it."""
    if torch.is_tensor(o):
        return o.detach().numpy().copy()
    if isinstance(o, (list, tuple)):
        return type(o)(_plain(v) for v in o)
    return o


def _tensors(o):
    import numpy as np
    if isinstance(o, np.ndarray):
        return torch.from_numpy(o)
    if isinstance(o, (list, tuple)):
        return type(o)(_tensors(v) for v in o)
    return o


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
    # the training loop's form of the same exchange: gradients accumulated by autograd straight into views of one flat
    # buffer, reduce-scatter + all-gather in place (the code path of the 8-GPU run: identical collectives on RCCL and gloo)
    torch.manual_seed(7)                               # identical replicas
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 2))
    unused = torch.nn.Parameter(torch.ones(3))
    ps = list(net.parameters()) + [unused]
    fg = P.FlatGradients(ps)
    flat_results = []
    for step in range(2):                              # two steps: the views must survive zero_() / backward / exchange
        fg.zero_()
        xb = torch.randn(4, 6, generator=torch.Generator().manual_seed(10 * step + rank))
        net(xb).pow(2).sum().backward()
        mine = [p.grad.clone() for p in ps]
        assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(fg.params, fg.views)), "autograd replaced a view"
        sent = fg.exchange_mean_()
        flat_results.append((mine, [p.grad.clone() for p in ps], sent))
    # the overlapped form the training loop uses: tiny buckets (several per model), reduce-scatters started from the
    # gradient hooks during backward, in bucket order; step 1 replaces one view with None (zero_grad(set_to_none) style)
    # (buckets are sent in order, so a parameter that never gets a gradient holds back its bucket and the later ones until
    # exchange_mean_: it is placed first here = in the last bucket)
    fo = P.FlatGradients([unused] + list(net.parameters()), overlap=True, bucket_bytes=64)
    assert fo.overlap and len(fo.buckets) >= 3 and all((hi - lo) % world == 0 for lo, hi in fo.buckets)
    started_in_backward = []
    for step in range(2):
        fo.zero_()
        xb = torch.randn(4, 6, generator=torch.Generator().manual_seed(10 * step + rank))
        net(xb).pow(2).sum().backward()
        started_in_backward.append(fo._next)
        sent = fo.exchange_mean_()
        flat_results.append((flat_results[step][0], [p.grad.clone() for p in ps], sent))
    flat_results.append(started_in_backward)
    # the unused parameter LAST = in the FIRST bucket to leave: in step 0 it holds back every bucket until exchange_mean_;
    # from step 1 on it is known to be unused (skip_unused) and the buckets leave from the hooks again
    fu = P.FlatGradients(list(net.parameters()) + [unused], overlap=True, bucket_bytes=64)
    unused_trace = []
    for step in range(3):
        fu.zero_()
        xb = torch.randn(4, 6, generator=torch.Generator().manual_seed(10 * (step % 2) + rank))
        net(xb).pow(2).sum().backward()
        started = fu._next
        fu.exchange_mean_()
        unused_trace.append((started, [p.grad.clone() for p in ps]))
    flat_results.append(unused_trace)
    lo, hi = P.shard_range(11, world, rank)
    t = P.max_over_ranks(1.0 + rank)
    q.put(_plain((rank, w0, local, [p.grad.clone() for p in params], nbytes, (lo, hi), t, P.rank_seed(5, rank), flat_results)))
    dist.destroy_process_group()


def test_two_rank_gradient_exchange_and_sharding():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([_tensors(q.get(timeout=120)) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, w0a, loc0, g0, nb0, s0, t0, seed0, fr0), (r1, w0b, loc1, g1, nb1, s1, t1, seed1, fr1) = res
    assert torch.equal(w0a, w0b)                                       # broadcast made the replicas identical
    for a, b in zip(g0, g1):
        assert torch.equal(a, b)                                       # every rank ends with the same averaged gradient
    assert torch.allclose(g0[0], (loc0[0] + loc1[0]) / 2)
    assert torch.allclose(g0[2], torch.full((7,), 0.5))                # missing gradient on rank 1 counted as zero
    assert nb0 == nb1 and nb0 >= (15 + 3 + 7) * 4
    assert s0 == (0, 6) and s1 == (6, 11)                              # contiguous, balanced, exhaustive
    assert t0 == t1 == 2.0                                             # max over ranks
    assert seed0 != seed1
    ut0, ut1 = fr0.pop(), fr1.pop()
    assert [st for st, _ in ut0] == [st for st, _ in ut1]
    assert ut0[0][0] == 0 and ut0[1][0] >= 2 and ut0[2][0] >= 2        # step 0 waits for the unused parameter, later steps do not
    for step, ((_, ga), (_, gb)) in enumerate(zip(ut0, ut1)):
        for a, b, want in zip(ga, gb, fr0[step % 2][1]):
            assert torch.equal(a, b) and torch.equal(a, want)              # same means as the plain exchange of the same batch
    started0, started1 = fr0.pop(), fr1.pop()
    assert started0 == started1 and min(started0) >= 1                 # buckets left the hooks during backward, same on both ranks
    assert len(fr0) == 4
    for (mine0, avg0, sent0), (mine1, avg1, sent1) in zip(fr0, fr1):
        assert sent0 == sent1 and sent0 >= (6 * 5 + 5 + 5 * 2 + 2 + 3) * 4
        for a, b, m0, m1 in zip(avg0, avg1, mine0, mine1):
            assert torch.equal(a, b) and torch.allclose(a, (m0 + m1) / 2, atol=1e-7)
        assert torch.equal(avg0[-1], torch.zeros(3))                   # a parameter outside the graph keeps a zero gradient


def test_epoch_shards_are_disjoint_exhaustive_and_equal():
    """TrainCondition._epoch_indices = DistributedSampler semantics (the reference's other tree: utils/rotinas.py:589-600):
    per epoch one permutation shared by all ranks, padded to a multiple of the world size, dealt out strided."""
    from hdiff_amd.DiffusionFreeGuidence.TrainCondition import _epoch_indices
    for n, world in ((512, 8), (101, 8), (7, 2), (64, 1), (5, 8)):
        for epoch in (0, 3):
            shards = [_epoch_indices(n, epoch, r, world) for r in range(world)]
            sizes = {len(s) for s in shards}
            assert sizes == {(n + world - 1) // world}, (n, world, sizes)          # equal shards: no rank waits in the exchange
            flat = [i for s in shards for i in s]
            assert set(flat) == set(range(n))                                       # exhaustive
            assert len(flat) - len(set(flat)) == (-n) % world if n >= world else True   # only the padding repeats
            if n % world == 0:
                assert len(set(flat)) == len(flat)                                  # disjoint
        assert _epoch_indices(n, 0, 0, world) != _epoch_indices(n, 1, 0, world) or n < 3   # reshuffled every epoch
        assert _epoch_indices(n, 2, 0, world) == _epoch_indices(n, 2, 0, world)            # deterministic


def _worker_world(rank, world, port, q):
    """The exchange at the world size it ships at (BASELINE config C4: 8 ranks): a parameter count that is not a multiple of
    the world size, one parameter outside the graph, bucketed reduce-scatters started from the gradient hooks."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, l, w = P.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(3)                               # identical replicas
    net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))      # 35 + 5 + 15 + 3 = 58 parameters
    unused = torch.nn.Parameter(torch.ones(3))                                                      # 61 in all: 61 % 8 = 5
    ps = list(net.parameters()) + [unused]
    P.broadcast_parameters_(ps, src=0)
    out = []
    for overlap in (False, True):
        fg = P.FlatGradients(ps, overlap=overlap, bucket_bytes=64) if overlap else P.FlatGradients(ps)
        assert all((hi - lo) % world == 0 for lo, hi in fg.buckets)          # every bucket splits evenly over the ranks
        for step in range(3):
            fg.zero_()
            xb = torch.randn(4, 7, generator=torch.Generator().manual_seed(100 * step + rank))
            net(xb).pow(2).sum().backward()
            mine = [p.grad.clone() for p in ps]
            started = fg._next if overlap else -1
            sent = fg.exchange_mean_()
            out.append((overlap, step, mine, [p.grad.clone() for p in ps], sent, started))
    lo, hi = P.shard_range(13, world, rank)            # 13 images over 8 ranks: five ranks take 2, three take 1
    q.put(_plain((rank, out, (lo, hi), P.max_over_ranks(float(rank)), P.rank_seed(11, rank))))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [8])
def test_exchange_at_the_shipping_world_size(world):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_world, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([_tensors(q.get(timeout=300)) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    outs = [r[1] for r in res]
    for k in range(len(outs[0])):                      # 2 exchange forms x 3 steps
        overlap, step = outs[0][k][0], outs[0][k][1]
        mean = [sum(o[k][2][i] for o in outs) / world for i in range(len(outs[0][k][2]))]
        for o in outs:
            assert o[k][4] == outs[0][k][4]                                   # same bytes on the wire on every rank
            assert o[k][5] == outs[0][k][5]                                   # same buckets started from the hooks
            for a, b, m in zip(o[k][3], outs[0][k][3], mean):
                assert torch.equal(a, b)                                      # replicas stay bitwise identical
                assert torch.allclose(a, m, rtol=1e-6, atol=1e-7), (overlap, step)
            assert torch.equal(o[k][3][-1], torch.zeros(3))                   # the parameter outside the graph: zero gradient
        if overlap and step >= 1:
            assert outs[0][k][5] >= 1, "from the second step on the buckets must leave during backward (skip_unused)"
    ranges = [r[2] for r in res]
    assert ranges[0][0] == 0 and ranges[-1][1] == 13 and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
    assert sorted(hi - lo for lo, hi in ranges) == [1, 1, 1, 2, 2, 2, 2, 2]
    assert all(r[3] == float(world - 1) for r in res)
    assert len({r[4] for r in res}) == world           # distinct per-rank seeds
