"""The optimizer tail of a training step on the HIP path: ``AdamW`` with the gradient clipping fused in.

Reference: ``TrainCondition.py:39-40`` builds ``torch.optim.AdamW(net_model.parameters(), lr, weight_decay=1e-4)`` and ``:61-63`` runs
``torch.nn.utils.clip_grad_norm_(net_model.parameters(), grad_clip); optimizer.step()`` after every backward.  Through round 5 this
harness did the same with torch's own kernels ("may stay torch-native at first"); this class is the native form:

    opt = hdiff_amd.optim.AdamW(net.parameters(), lr=1e-4, weight_decay=1e-4)
    ...
    loss.backward()
    total_norm = opt.step(max_grad_norm=1.0)      # == clip_grad_norm_(params, 1.0); optimizer.step(): three launches for all tensors

It is a ``torch.optim.Optimizer`` (param groups, ``zero_grad``, LR schedulers -- ``Scheduler.GradualWarmupScheduler`` and
``CosineAnnealingLR`` set ``group["lr"]`` -- and ``state_dict`` work as for torch's AdamW: per-parameter ``step`` / ``exp_avg`` /
``exp_avg_sq``, the moments being views into one flat buffer per group).  The arithmetic is torch's single-tensor AdamW in its order of
operations (``csrc/optimizer.hip``); the norm is ONE sum over all elements (fixed order, float64 final sum) instead of torch's norm of
per-tensor norms: the same number to fp32 rounding.  No CPU path: CPU parameters raise.
"""
from __future__ import annotations

import ctypes as C
from typing import Iterable, List, Optional, Tuple

import numpy as np
import torch

from . import _capi


class AdamW(torch.optim.Optimizer):
    def __init__(self, params: Iterable, lr: float = 1e-3, betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-2):
        if lr < 0.0 or eps < 0.0 or weight_decay < 0.0 or not (0.0 <= betas[0] < 1.0) or not (0.0 <= betas[1] < 1.0):
            raise ValueError(f"Invalid hyper-parameters: lr={lr} betas={betas} eps={eps} weight_decay={weight_decay}")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self._lib = _capi.lib()
        self._chunk = int(self._lib.hdiff_opt_chunk())
        self._tables: dict = {}          # group index -> (signature, table tensor, chunk tensor, nchunks)
        self._scratch: Optional[torch.Tensor] = None
        self._norm_coef: Optional[torch.Tensor] = None

    # -- state: one flat buffer per group for each moment, the per-parameter entries are views (torch's state layout) ------------
    def _ensure_state(self, gi: int, group: dict) -> None:
        need = [p for p in group["params"] if p.requires_grad and "exp_avg" not in self.state[p]]
        if not need:
            return
        dev = need[0].device
        n = sum(p.numel() for p in need)
        m_flat, v_flat = torch.zeros(n, dtype=torch.float32, device=dev), torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        shared_step = torch.tensor(0.0)      # ONE step counter object for the parameters that start together (incremented once per step)
        for p in need:
            st = self.state[p]
            st["step"] = shared_step
            st["exp_avg"] = m_flat[off:off + p.numel()].view_as(p)
            st["exp_avg_sq"] = v_flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def _table(self, gi: int, group: dict):
        ps = [p for p in group["params"] if p.grad is not None]
        for p in ps:
            if not p.is_cuda or p.dtype != torch.float32 or p.grad.dtype != torch.float32:
                raise RuntimeError("hdiff_amd.optim.AdamW runs on the GPU in fp32 only (there is no CPU path)")
            if not (p.is_contiguous() and p.grad.is_contiguous()):
                raise RuntimeError("hdiff_amd.optim.AdamW: parameters and gradients must be contiguous")
        sig = tuple((p.data_ptr(), p.grad.data_ptr(), self.state[p]["exp_avg"].data_ptr()) for p in ps)
        hit = self._tables.get(gi)
        if hit is not None and hit[0] == sig:
            return hit
        if not ps:
            self._tables[gi] = (sig, None, None, 0, ps)
            return self._tables[gi]
        tab = np.zeros((len(ps), 5), dtype=np.int64)
        chunks: List[Tuple[int, int]] = []
        for i, p in enumerate(ps):
            st = self.state[p]
            tab[i] = (p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel())
            chunks.extend((i, c) for c in range((p.numel() + self._chunk - 1) // self._chunk))
        dev = ps[0].device
        t_dev = torch.from_numpy(tab).to(dev)
        c_dev = torch.tensor(chunks, dtype=torch.int32).to(dev)
        self._tables[gi] = (sig, t_dev, c_dev, len(chunks), ps)
        return self._tables[gi]

    @torch.no_grad()
    def step(self, closure=None, max_grad_norm: Optional[float] = None):
        """One AdamW step; with ``max_grad_norm`` the gradients of ALL groups are first clipped to that global norm (in place, like
        ``clip_grad_norm_``) and the total norm (0-dim device tensor) is returned, otherwise the closure's loss / None."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        tables = []
        for gi, group in enumerate(self.param_groups):
            self._ensure_state(gi, group)
            tables.append(self._table(gi, group))
        total = sum(t[3] for t in tables)
        if total == 0:
            return loss
        dev = next(t[1] for t in tables if t[1] is not None).device
        stream = torch.cuda.current_stream(dev).cuda_stream
        coef_ptr = None
        if max_grad_norm is not None:
            if self._scratch is None or self._scratch.numel() < total or self._scratch.device != dev:
                self._scratch = torch.empty(total, dtype=torch.float32, device=dev)
                self._norm_coef = torch.empty(2, dtype=torch.float32, device=dev)
            if len([t for t in tables if t[3]]) == 1:
                _, t_dev, c_dev, n, _ = next(t for t in tables if t[3])
                _capi.check(self._lib.hdiff_grad_norm_clip_coef(t_dev.data_ptr(), c_dev.data_ptr(), n, self._scratch.data_ptr(),
                                                                C.c_float(float(max_grad_norm)), self._norm_coef.data_ptr(), stream),
                            "grad_norm_clip_coef")
            else:      # several groups: one table over all of them for the norm (built per call: rare, small)
                ps = [p for t in tables for p in t[4]]
                tab = np.array([(p.data_ptr(), p.grad.data_ptr(), 0, 0, p.numel()) for p in ps], dtype=np.int64)
                ch = [(i, c) for i, p in enumerate(ps) for c in range((p.numel() + self._chunk - 1) // self._chunk)]
                t_dev, c_dev = torch.from_numpy(tab).to(dev), torch.tensor(ch, dtype=torch.int32).to(dev)
                _capi.check(self._lib.hdiff_grad_norm_clip_coef(t_dev.data_ptr(), c_dev.data_ptr(), len(ch), self._scratch.data_ptr(),
                                                                C.c_float(float(max_grad_norm)), self._norm_coef.data_ptr(), stream),
                            "grad_norm_clip_coef")
                self._keep = (t_dev, c_dev)
            coef_ptr = self._norm_coef.data_ptr()
        for group, (_, t_dev, c_dev, n, ps) in zip(self.param_groups, tables):
            if n == 0:
                continue
            counters = {id(self.state[p]["step"]): self.state[p]["step"] for p in ps}      # distinct counter objects (one, unless states were loaded)
            steps = {int(c.item()) if torch.is_tensor(c) else int(c) for c in counters.values()}
            if len(steps) != 1:
                raise RuntimeError("hdiff_amd.optim.AdamW: the parameters of a group must have taken the same number of steps")
            step = steps.pop() + 1
            b1, b2 = group["betas"]
            _capi.check(self._lib.hdiff_adamw_step(t_dev.data_ptr(), c_dev.data_ptr(), n, coef_ptr, float(group["lr"]), float(b1), float(b2),
                                                   float(group["eps"]), float(group["weight_decay"]), step, stream), "adamw_step")
            if all(torch.is_tensor(c) for c in counters.values()):
                for c in counters.values():
                    c.fill_(float(step))
            else:
                for p in ps:
                    self.state[p]["step"] = torch.tensor(float(step))
        if max_grad_norm is not None:
            return self._norm_coef[0].clone()
        return loss
