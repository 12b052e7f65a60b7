"""Drop-in for the reference's ``DiffusionFreeGuidence/ModelCondition.py`` (lines 1-277): same class names, constructor
and ``forward`` signatures, parameter names/shapes (the 366-entry ``state_dict``) and default initialisation.

The ``torch.nn`` leaf modules below (``nn.Conv2d``, ``nn.GroupNorm``, ``nn.Linear``, ``nn.Embedding``,
``nn.MultiheadAttention`` ...) are used ONLY as parameter containers, created in the reference's order so that the same
``torch.manual_seed`` gives bit-identical initial weights and the same ``state_dict`` keys.  Their own ``forward`` methods
are never called: every ``forward`` in this file issues hand-written gfx950 kernels through ``libhdiff.so``
(``..engine``).  On a tensor that is not on an MI355X device the forwards raise -- there is no CPU fallback.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
from torch import nn

from .. import engine as E

__all__ = ["Swish", "TimeEmbedding", "ConditionalEmbedding", "DownSample", "UpSample", "AttnBlock", "ResBlock", "UNet"]


def _params_of(mod: nn.Module, prefix: str = "") -> Dict[str, torch.Tensor]:
    return {prefix + k: v for k, v in mod.state_dict(keep_vars=True).items()}


def _inputs(**tensors: torch.Tensor):
    """The caller's tensors as the kernels need them: on the GPU (there is no CPU fallback), dense NCHW (a strided view is
    copied, as ATen would do internally), indices as int64 (nn.Embedding also takes int32)."""
    out = []
    for name, t in tensors.items():
        if t.is_cuda and not t.is_contiguous():
            t = t.contiguous()
        E.require_gpu_tensor(t, name)
        if t.dtype == torch.int32:
            t = t.to(torch.int64)
        out.append(t)
    return out[0] if len(out) == 1 else tuple(out)


class _EagerMixin:
    """Run a sub-module on its own: build a one-off plan, pack its weights, launch, return a fresh tensor."""

    @staticmethod
    def _finish(plan: E.Plan, out: torch.Tensor) -> torch.Tensor:
        plan.pack_weights()
        plan.run()
        return out.clone()


class Swish(nn.Module):
    """x * sigmoid(x) (reference ModelCondition.py:22-24).  Inside the UNet it is fused into the consuming kernels."""

    def forward(self, x):
        x = _inputs(x=x)
        plan = E.Plan(x.device)
        n = x.numel()
        one = torch.ones(1, device=x.device)
        zero = torch.zeros(1, device=x.device)
        y = torch.empty_like(x)
        plan.call("hdiff_gn_swish_apply", x.data_ptr(), one.data_ptr(), zero.data_ptr(), y.data_ptr(), 1, 1, n)
        plan.run()
        return y


class TimeEmbedding(nn.Module, _EagerMixin):
    """Sinusoidal table (trainable) -> Linear -> Swish -> Linear (reference ModelCondition.py:27-49)."""

    def __init__(self, T, d_model, dim):
        assert d_model % 2 == 0
        super().__init__()
        freqs = torch.exp(-(torch.arange(0, d_model, step=2) / d_model * math.log(10000)))
        ang = torch.arange(T).float()[:, None] * freqs[None, :]
        table = torch.stack([torch.sin(ang), torch.cos(ang)], dim=-1).view(T, d_model)
        self.timembedding = nn.Sequential(
            nn.Embedding.from_pretrained(table, freeze=False),
            nn.Linear(d_model, dim),
            Swish(),
            nn.Linear(dim, dim),
        )

    def forward(self, t):
        t = _inputs(t=t)
        plan = E.Plan(t.device)
        out = E.emit_embed_mlp(plan, _params_of(self.timembedding, "e."), "e", t, int(t.shape[0]))
        return self._finish(plan, out)


class ConditionalEmbedding(nn.Module, _EagerMixin):
    """Embedding(num_labels+1, padding_idx=0) -> Linear -> Swish -> Linear (reference ModelCondition.py:52-65)."""

    def __init__(self, num_labels, d_model, dim):
        assert d_model % 2 == 0
        super().__init__()
        self.condEmbedding = nn.Sequential(
            nn.Embedding(num_embeddings=num_labels + 1, embedding_dim=d_model, padding_idx=0),
            nn.Linear(d_model, dim),
            Swish(),
            nn.Linear(dim, dim),
        )

    def forward(self, t):
        t = _inputs(labels=t)
        plan = E.Plan(t.device)
        out = E.emit_embed_mlp(plan, _params_of(self.condEmbedding, "e."), "e", t, int(t.shape[0]))
        return self._finish(plan, out)


class DownSample(nn.Module, _EagerMixin):
    """Conv3x3/s2 + Conv5x5/s2, summed (reference ModelCondition.py:68-76); temb/cemb are ignored."""

    def __init__(self, in_ch):
        super().__init__()
        self.c1 = nn.Conv2d(in_ch, in_ch, 3, stride=2, padding=1)
        self.c2 = nn.Conv2d(in_ch, in_ch, 5, stride=2, padding=2)

    def forward(self, x, temb, cemb):
        x = _inputs(x=x)
        B, Cc, H, W = (int(v) for v in x.shape)
        plan = E.Plan(x.device)
        return self._finish(plan, E.emit_downsample(plan, _params_of(self, "m."), "m", x, B, Cc, H, W))


class UpSample(nn.Module, _EagerMixin):
    """ConvTranspose2d(5, 2, 2, 1) then Conv3x3 (reference ModelCondition.py:79-89)."""

    def __init__(self, in_ch):
        super().__init__()
        self.c = nn.Conv2d(in_ch, in_ch, 3, stride=1, padding=1)
        self.t = nn.ConvTranspose2d(in_ch, in_ch, 5, 2, 2, 1)

    def forward(self, x, temb, cemb):
        x = _inputs(x=x)
        B, Cc, H, W = (int(v) for v in x.shape)
        plan = E.Plan(x.device)
        return self._finish(plan, E.emit_upsample(plan, _params_of(self, "m."), "m", x, B, Cc, H, W))


class AttnBlock(nn.Module, _EagerMixin):
    """GroupNorm -> 1x1 q, k, v -> softmax(q k^T * C^-1/2) v (ONE head of width in_ch) -> 1x1 proj -> x + h
    (reference ModelCondition.py:92-120).  The reference's UNet never instantiates it (ResBlock uses
    nn.MultiheadAttention instead, :189); it is provided because the module is part of the file's surface.  Heads up to 64
    channels wide run on the flash kernel, wider ones on a plain row-per-workgroup kernel; with gradients enabled the block
    goes through autograd.attn_block (hand-written backward: hdiff_mha_wide_bwd, hdiff_gn_affine_bwd, the conv backwards)."""

    def __init__(self, in_ch):
        super().__init__()
        self.group_norm = nn.GroupNorm(32, in_ch)
        self.proj_q = nn.Conv2d(in_ch, in_ch, 1, stride=1, padding=0)
        self.proj_k = nn.Conv2d(in_ch, in_ch, 1, stride=1, padding=0)
        self.proj_v = nn.Conv2d(in_ch, in_ch, 1, stride=1, padding=0)
        self.proj = nn.Conv2d(in_ch, in_ch, 1, stride=1, padding=0)

    def forward(self, x):
        x = _inputs(x=x)
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            from ..autograd import attn_block
            return attn_block(self, x)
        B, Cc, H, W = (int(v) for v in x.shape)
        with torch.cuda.device(x.device):
            plan = E.Plan(x.device)
            return self._finish(plan, E.emit_attn_block(plan, _params_of(self, "m."), "m", x, B, Cc, H, W))


class ResBlock(nn.Module, _EagerMixin):
    """GN-Swish-Conv (+temb +cemb) -> GN-Swish-Dropout-Conv -> + shortcut(x) -> optional 8-head self-attention
    with no pre-norm and no residual (reference ModelCondition.py:166-211)."""

    def __init__(self, in_ch, out_ch, tdim, dropout, attn=True):
        super().__init__()
        self.block1 = nn.Sequential(nn.GroupNorm(32, in_ch), Swish(), nn.Conv2d(in_ch, out_ch, 3, stride=1, padding=1))
        self.temb_proj = nn.Sequential(Swish(), nn.Linear(tdim, out_ch))
        self.cond_proj = nn.Sequential(Swish(), nn.Linear(tdim, out_ch))
        self.block2 = nn.Sequential(nn.GroupNorm(32, out_ch), Swish(), nn.Dropout(dropout),
                                    nn.Conv2d(out_ch, out_ch, 3, stride=1, padding=1))
        self.attn = nn.MultiheadAttention(out_ch, num_heads=8) if attn else nn.Identity()
        self.shortcut = nn.Conv2d(in_ch, out_ch, 1, stride=1, padding=0) if in_ch != out_ch else nn.Identity()
        self.out_ch = out_ch

    def forward(self, x, temb, cemb=None):
        x, temb = _inputs(x=x, temb=temb)
        if cemb is not None:
            cemb = _inputs(cemb=cemb)
        if _dropout_active(self):
            # nn.Dropout in train mode runs whether or not autograd records (ModelCondition.py:185): the eager launch path
            # carries the dropout kernels (keep-mask from torch's generator), the static plan does not
            from ..autograd import res_block
            return res_block(self, x, None, temb, cemb, True)
        B, _, H, W = (int(v) for v in x.shape)
        plan = E.Plan(x.device)
        has_attn = isinstance(self.attn, nn.MultiheadAttention)
        out = E.emit_resblock(plan, _params_of(self, "m."), "m", x, None, temb, cemb, self.out_ch, B, H, W, has_attn)
        plan.flush_block_vecs(0, temb, cemb, B)
        return self._finish(plan, out)


def _dropout_active(mod: nn.Module) -> bool:
    return any(isinstance(m, nn.Dropout) and m.training and m.p > 0 for m in mod.modules())


class UNet(nn.Module):
    """Conditional U-Net denoiser (reference ModelCondition.py:213-276): forward(x[B,3,H,W], t[B], labels[B]) -> eps."""

    MAX_CACHED_PLANS = 2     # launch plans (with their preallocated activations) kept per model, least recently used evicted

    def __init__(self, T, num_labels, ch, ch_mult, num_res_blocks, dropout):
        super().__init__()
        tdim = ch * 4
        self.time_embedding = TimeEmbedding(T, ch, tdim)
        self.cond_embedding = ConditionalEmbedding(num_labels, ch, tdim)
        self.head = nn.Conv2d(3, ch, kernel_size=3, stride=1, padding=1)
        self.downblocks = nn.ModuleList()
        widths = [ch]
        now = ch
        last_level = len(ch_mult) - 1
        for level, mult in enumerate(ch_mult):
            for _ in range(num_res_blocks):
                self.downblocks.append(ResBlock(in_ch=now, out_ch=ch * mult, tdim=tdim, dropout=dropout))
                now = ch * mult
                widths.append(now)
            if level != last_level:
                self.downblocks.append(DownSample(now))
                widths.append(now)
        self.middleblocks = nn.ModuleList([ResBlock(now, now, tdim, dropout, attn=True),
                                           ResBlock(now, now, tdim, dropout, attn=False)])
        self.upblocks = nn.ModuleList()
        for level in range(last_level, -1, -1):
            for _ in range(num_res_blocks + 1):
                self.upblocks.append(ResBlock(in_ch=widths.pop() + now, out_ch=ch * ch_mult[level], tdim=tdim,
                                              dropout=dropout, attn=False))
                now = ch * ch_mult[level]
            if level != 0:
                self.upblocks.append(UpSample(now))
        assert len(widths) == 0
        self.tail = nn.Sequential(nn.GroupNorm(32, now), Swish(), nn.Conv2d(now, 3, 3, stride=1, padding=1))
        self._shape = E.UNetShape(T=T, num_labels=num_labels, ch=ch, ch_mult=tuple(ch_mult), num_res_blocks=num_res_blocks)
        self._plans: Dict[tuple, E.UNetPlan] = {}
        self._plan_ptrs: Optional[tuple] = None
        self._packed_versions: Dict[tuple, tuple] = {}

    # -- plan cache ---------------------------------------------------------------------------------------------------
    def _param_signature(self):
        ps = list(self.parameters())
        return tuple(p.data_ptr() for p in ps), tuple(p._version for p in ps)

    def invalidate_packed(self) -> None:
        """Force a repack on the next forward.  Needed only after a write that bypasses autograd's version counter
        (``p.data.copy_``, ``dist.broadcast(p.data)``) -- to a convolution weight OR to a GroupNorm weight / bias (the fp16-pair
        convolutions stage their activations with a power of two computed from the GroupNorm parameters at pack time);
        ``p.copy_`` under ``no_grad``, optimizer steps and ``load_state_dict`` are seen automatically, and
        ``GaussianDiffusionSampler.forward`` repacks on every call."""
        self._packed_versions.clear()

    def plan_for(self, B: int, H: int, W: int, device) -> E.UNetPlan:
        """Launch plan for (B, H, W) with up-to-date packed weights (repacked whenever a parameter changed)."""
        ptrs, versions = self._param_signature()
        if ptrs != self._plan_ptrs:
            self._plans.clear()
            self._packed_versions.clear()
            self._plan_ptrs = ptrs
        key = (B, H, W, str(device))
        up = self._plans.pop(key, None)
        if up is None:
            while len(self._plans) >= self.MAX_CACHED_PLANS:        # a plan owns all its activations: keep few
                old = next(iter(self._plans))
                self._plans.pop(old)
                self._packed_versions.pop(old, None)
            up = E.UNetPlan(_params_of(self), self._shape, B, H, W, device)
        self._plans[key] = up                                       # most recently used last
        if self._packed_versions.get(key) != versions:
            up.plan.pack_weights()
            self._packed_versions[key] = versions
        return up

    def check_indices(self, t, labels) -> None:
        """nn.Embedding raises IndexError on an out-of-range index (reference ModelCondition.py:38,56); the kernels clamp
        instead of faulting, so the range is validated here (one device->host read)."""
        lim = torch.stack([t.min(), t.max(), labels.min(), labels.max()]).tolist()
        if lim[0] < 0 or lim[1] >= self._shape.T or lim[2] < 0 or lim[3] > self._shape.num_labels:
            raise IndexError("index out of range in self")

    def forward(self, x, t, labels):
        x, t, labels = _inputs(x=x, t=t, labels=labels)
        B, Cx, H, W = (int(v) for v in x.shape)
        if Cx != 3:
            raise RuntimeError(f"expected input[{B}, {Cx}, {H}, {W}] to have 3 channels")
        with torch.cuda.device(x.device):
            records = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))
            if records or _dropout_active(self):
                # training, or train-mode dropout under no_grad (the reference applies nn.Dropout there too,
                # ModelCondition.py:185): eager launches through the autograd Functions, which carry the dropout kernels
                from ..autograd import unet_forward_with_grad
                return unet_forward_with_grad(self, x, t, labels)
            self.check_indices(t, labels)
            up = self.plan_for(B, H, W, x.device)
            up.x.copy_(x)
            up.t.copy_(t)
            up.labels.copy_(labels)
            up.plan.run()
            return up.out.clone()
