"""One process per GPU over torch.distributed (backend "nccl" is RCCL on ROCm; "gloo" for the CPU rehearsal in tests).

The reference's hot path has no distributed code (TrainCondition.py:21 uses a single device); its other tree launches
DDP with one process per GPU (utils/rotinas.py:572-577, 619).  The hot path shards like this:

* sampling: images are independent -- each rank takes a contiguous slice of the batch with its own seed and there is NO
  data-path collective (only the bench's barrier / max-over-ranks timing);
* training: replicated weights, per-rank mini-batch, ONE exchange per optimizer step: the mean of the 47.8 M fp32
  gradients (190.8 MB).  On the MI355X node xGMI is a full mesh of point-to-point links, so the flat gradient buffer is
  reduced with reduce-scatter + all-gather (every link busy at once) rather than a ring all-reduce of small buckets.
  The gradients live as views in that flat buffer (FlatGradients): no gather / scatter copies around the exchange, and
  the reduce-scatter of each 64 MB bucket starts from a gradient hook while backward is still running.
  NO scaling curve has been measured yet (the build box has one GPU; the driver's 8-GPU run was skipped in round 1);
  RCCL has executed this exchange in a one-rank group on the GPU, gloo with two ranks (tests/).
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import time
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


def launch_ranks(script: str, argv: Sequence[str], nproc: int, poll_s: float = 0.2) -> int:
    """Start `nproc` fresh rank processes of `script` on this node, one per GPU, and wait for them; returns the job's exit
    code (0 only if every rank returned 0).  The counterpart of the reference's `mp.spawn(train, nprocs=world_size)`
    (utils/rotinas.py:572-577) for programs that are started as plain `python prog.py --gpus N`.

    The caller must NOT have touched the GPU (no HIP call, no `torch.cuda.is_available()`): children are new processes
    (`subprocess.Popen`, never `os.exec*`) that initialise the device themselves from RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT, exactly the environment `torch.distributed.run` would hand them.  Rank 0 inherits stdout
    (the one JSON line of a bench goes through untouched); the other ranks' stdout is sent to stderr.  When a rank
    fails, the remaining ranks -- exactly the PIDs started here -- are terminated so that nobody waits in a collective."""
    nproc = int(nproc)
    if nproc < 1:
        raise ValueError("launch_ranks: nproc must be >= 1")
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs: List[subprocess.Popen] = []
    for r in range(nproc):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nproc), LOCAL_WORLD_SIZE=str(nproc),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, script, *argv], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    live = set(range(nproc))
    try:
        while live:
            for r in sorted(live):
                code = procs[r].poll()
                if code is None:
                    continue
                live.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    print(f"launch_ranks: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr)
                    for q in live:
                        procs[q].terminate()
            if live:
                time.sleep(poll_s)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
            p.wait()
    return rc


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

    The buffer holds the parameters in REVERSE order (backward produces the last layer's gradients first) and is cut into
    buckets of about `bucket_bytes` (each padded to a multiple of the world size).  With `overlap=True` every parameter
    carries a post-accumulate hook; when the last gradient of a bucket has landed -- and every earlier bucket has been
    started, so all ranks issue the collectives in one order -- the bucket's reduce-scatter is started asynchronously while
    backward goes on.  `exchange_mean_()` starts whatever has not been started (parameters without a gradient), waits,
    scales the shards and all-gathers them back into the views.  64 MB buckets: large enough that the full-mesh xGMI
    reduce-scatter runs at link rate, small enough that only the first layers' bucket is left when backward ends.
    """

    def __init__(self, params: Sequence[torch.nn.Parameter], world: int | None = None, overlap: bool = False,
                 bucket_bytes: int = 64 << 20, single_rank_collectives: bool = False, skip_unused: bool = True):
        self.params = [p for p in params if p.requires_grad]
        self.world = world if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        w = self.world
        dev = self.params[0].device if self.params else "cpu"
        order = list(range(len(self.params)))[::-1]
        # bucket boundaries over the reversed parameter list
        self.buckets: List[Tuple[int, int]] = []           # [lo, hi) in the flat buffer, hi - lo divisible by world
        owner = [0] * len(self.params)
        offs = [0] * len(self.params)
        off, lo, cap = 0, 0, max(int(bucket_bytes) // 4, 1)
        for i in order:
            p = self.params[i]
            if off > lo and off - lo + p.numel() > cap:     # close the bucket before this parameter
                off = lo + (off - lo + w - 1) // w * w
                self.buckets.append((lo, off))
                lo = off
            owner[i], offs[i] = len(self.buckets), off
            off += p.numel()
        off = lo + (off - lo + w - 1) // w * w
        if off > lo or not self.buckets:
            self.buckets.append((lo, off))
        self.numel = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        self.views: List[torch.Tensor] = [self.flat[offs[i]:offs[i] + p.numel()].view_as(p) for i, p in enumerate(self.params)]
        self.shards = [torch.empty((hi - lo) // w, dtype=torch.float32, device=dev) for lo, hi in self.buckets]
        self._owner = owner
        self._count = [0] * len(self.buckets)
        for b in owner:
            self._count[b] += 1
        # single_rank_collectives: issue the collectives even in a one-rank group (tests: the RCCL calls, their stream
        # ordering against the backward kernels and the hook threading run on the one GPU of the build box; the mean over
        # one rank must give the gradients back bit for bit)
        self.single = bool(single_rank_collectives)
        self.overlap = bool(overlap) and (w > 1 or self.single)
        # skip_unused (overlap only): a parameter that received no gradient in the FIRST step (an unused module, a frozen
        # branch) is not waited for from then on -- otherwise it would hold back its bucket and every later one until
        # exchange_mean_() and the overlap would silently be lost.  The assumption is DDP's static graph: the set of used
        # parameters does not change; a gradient for such a parameter arriving after its bucket has left raises.
        self.skip_unused = bool(skip_unused)
        self._absent: set = set()
        self._learned = False
        self._fired: List[bool] = [False] * len(self.params)
        self._pending: List[int] = []
        self._ready: List[bool] = []
        self._work: list = []
        self._next = 0
        self._hooks = []
        if self.overlap:
            for i, p in enumerate(self.params):
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))
        self._arm()

    def _arm(self) -> None:
        self._pending = list(self._count)
        for i in self._absent:
            self._pending[self._owner[i]] -= 1
        self._fired = [False] * len(self.params)
        self._ready = [n == 0 for n in self._pending]      # a bucket of unused parameters only: goes out with its neighbours
        self._work = [None] * len(self.buckets)
        self._next = 0

    def _make_hook(self, i: int):
        def hook(p: torch.Tensor) -> None:
            if p.grad is None or p.grad.data_ptr() != self.views[i].data_ptr():     # the view was replaced behind our back: exchange_mean_ repairs and starts it
                self._fired[i] = True     # it DID receive a gradient: must not be learned as unused (skip_unused)
                return
            b = self._owner[i]
            if b < self._next:
                raise RuntimeError("FlatGradients(overlap=True): a gradient arrived after its bucket had been sent -- one "
                                   "backward per zero_() / exchange_mean_(); accumulate over several backwards with overlap=False"
                                   + ("; this parameter got no gradient in the first step and was no longer waited for: "
                                      "pass skip_unused=False if the set of used parameters changes" if i in self._absent else ""))
            self._fired[i] = True
            if i in self._absent:                            # used again, and in time: its bucket simply was not waiting for it
                return
            self._pending[b] -= 1
            if self._pending[b] == 0:
                self._ready[b] = True
                self._start_ready()
        return hook

    def _start(self, b: int) -> None:
        lo, hi = self.buckets[b]
        self._work[b] = dist.reduce_scatter_tensor(self.shards[b], self.flat[lo:hi], op=dist.ReduceOp.SUM, async_op=True)

    def _start_ready(self) -> None:
        while self._next < len(self.buckets) and self._ready[self._next]:
            self._start(self._next)
            self._next += 1

    def zero_(self) -> None:
        self.flat.zero_()
        for p, v in zip(self.params, self.views):
            p.grad = v                      # parameters that receive no gradient this step contribute zeros
        self._arm()

    def exchange_mean_(self) -> int:
        """Returns the bytes each rank contributes to the exchange (0 when there is nothing to exchange)."""
        if (self.world == 1 and not self.single) or not (dist.is_available() and dist.is_initialized()):
            return 0
        started = self._next
        for i, (p, v) in enumerate(zip(self.params, self.views)):
            if p.grad is None or p.grad.data_ptr() != v.data_ptr():     # someone replaced the view (e.g. zero_grad(set_to_none=True))
                assert self._owner[i] >= started, "a gradient view was replaced after its bucket had been sent"
                if p.grad is None:
                    v.zero_()
                else:
                    v.copy_(p.grad)
                p.grad = v
        for b in range(self._next, len(self.buckets)):      # buckets the hooks did not start (or all of them, overlap off)
            self._start(b)
        self._next = len(self.buckets)
        gathers = []
        for b, (lo, hi) in enumerate(self.buckets):
            self._work[b].wait()
            self.shards[b].div_(self.world)
            gathers.append(dist.all_gather_into_tensor(self.flat[lo:hi], self.shards[b], async_op=True))
        for g in gathers:
            g.wait()
        if self.overlap and self.skip_unused and not self._learned:
            self._absent = {i for i, f in enumerate(self._fired) if not f}
            self._learned = True
        self._arm()
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
