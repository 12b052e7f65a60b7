"""One process per GPU over torch.distributed (backend "nccl" is RCCL on ROCm; "gloo" for the CPU rehearsal in tests).

The reference's hot path has no distributed code (TrainCondition.py:21 uses a single device); its other tree launches
DDP with one process per GPU (utils/rotinas.py:572-577, 619).  The hot path shards like this:

* sampling: images are independent -- each rank takes a contiguous slice of the batch with its own seed and there is NO
  data-path collective (only the bench's barrier / max-over-ranks timing);
* training: replicated weights, per-rank mini-batch, ONE exchange per optimizer step: the mean of the 47.8 M fp32
  gradients (190.8 MB).  On the MI355X node xGMI is a full mesh of point-to-point links, so the flat gradient buffer is
  reduced with reduce-scatter + all-gather (every link busy at once) rather than a ring all-reduce of small buckets.
  The gradients live as views in that flat buffer (FlatGradients): no gather / scatter copies around the exchange.
  NO scaling curve has been measured yet (the build box has one GPU; the driver's 8-GPU run was skipped in round 1).
"""
from __future__ import annotations

import os
from typing import Iterable, List, Sequence, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None) -> Tuple[int, int, int]:
    """Initialise the default process group from RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*; returns (rank, local, world)."""
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend=backend, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend)
    return rank, local, world


def shard_range(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced slice [lo, hi) of n independent units for this rank (first n % world ranks get one extra)."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def rank_seed(base_seed: int, rank: int) -> int:
    """Per-rank RNG seed for independent sampling trajectories."""
    return (int(base_seed) * 1000003 + 7919 * int(rank)) & 0x7FFFFFFFFFFFFFFF


def broadcast_parameters_(params: Iterable[torch.Tensor], src: int = 0) -> None:
    """Make every rank start from rank `src`'s weights (what DDP does at construction, rotinas.py:619)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    with torch.no_grad():
        for p in params:
            buf = p.detach().clone()
            dist.broadcast(buf, src=src)
            p.copy_(buf)      # an in-place write autograd sees: bumps p._version, so cached weight packs are refreshed


def _exchange_mean_(flat: torch.Tensor, world: int) -> None:
    """Mean over ranks of one flat fp32 buffer (numel divisible by world), in place: reduce-scatter + all-gather.
    The same two collectives on every backend (RCCL on the GPUs, gloo in the CPU rehearsal): on the MI355X node xGMI is a
    full mesh of point-to-point links, and a reduce-scatter / all-gather pair keeps all seven links of every GPU busy where
    a ring all-reduce of small buckets is bound by one link."""
    shard = torch.empty(flat.numel() // world, dtype=flat.dtype, device=flat.device)
    dist.reduce_scatter_tensor(shard, flat, op=dist.ReduceOp.SUM)
    shard.div_(world)
    dist.all_gather_into_tensor(flat, shard)


class FlatGradients:
    """The gradients of a parameter list as views into ONE flat fp32 buffer, so that the per-step exchange of data-parallel
    training (TrainCondition.train; the reference's other tree wraps its model in DDP, utils/rotinas.py:619) needs no gather /
    scatter copies: autograd accumulates straight into the views, the exchange runs in place, the optimizer reads the views.

        flat = FlatGradients(params)         # once
        flat.zero_()                         # instead of optimizer.zero_grad(): zero the buffer, (re)attach p.grad
        loss.backward(); flat.exchange_mean_(); clip_grad_norm_(params, ...); optimizer.step()
    """

    def __init__(self, params: Sequence[torch.nn.Parameter], world: int | None = None):
        self.params = [p for p in params if p.requires_grad]
        self.world = world if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        n = sum(p.numel() for p in self.params)
        self.numel = n
        padded = (n + self.world - 1) // self.world * self.world
        dev = self.params[0].device if self.params else "cpu"
        self.flat = torch.zeros(padded, dtype=torch.float32, device=dev)
        self.views: List[torch.Tensor] = []
        off = 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()

    def zero_(self) -> None:
        self.flat.zero_()
        for p, v in zip(self.params, self.views):
            p.grad = v                      # parameters that receive no gradient this step contribute zeros

    def exchange_mean_(self) -> int:
        """Returns the bytes each rank contributes to the exchange (0 when there is nothing to exchange)."""
        if self.world == 1 or not (dist.is_available() and dist.is_initialized()):
            return 0
        for p, v in zip(self.params, self.views):
            if p.grad is not v:             # someone replaced the view (e.g. optimizer.zero_grad(set_to_none=True))
                if p.grad is None:
                    v.zero_()
                else:
                    v.copy_(p.grad)
                p.grad = v
        _exchange_mean_(self.flat, self.world)
        return self.flat.numel() * 4


def allreduce_mean_grads_(params: Sequence[torch.nn.Parameter]) -> int:
    """Average the gradients of `params` over all ranks in ONE flat fp32 buffer; returns the bytes exchanged per rank.
    One-shot form of :class:`FlatGradients` (gathers the gradients into a fresh flat buffer and scatters the means back);
    parameters without a gradient contribute zeros (the reference's optimizer skips them identically on every rank)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 0
    world = dist.get_world_size()
    ps = [p for p in params if p.requires_grad]
    if not ps:
        return 0
    dev = ps[0].device
    n = sum(p.numel() for p in ps)
    padded = (n + world - 1) // world * world
    flat = torch.zeros(padded, dtype=torch.float32, device=dev)
    off = 0
    for p in ps:
        if p.grad is not None:
            flat[off:off + p.numel()].copy_(p.grad.reshape(-1))
        off += p.numel()
    _exchange_mean_(flat, world)
    off = 0
    for p in ps:
        g = flat[off:off + p.numel()].view_as(p)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        off += p.numel()
    return padded * 4


def max_over_ranks(value: float, device=None) -> float:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
