"""Drop-in for the live classes of the reference's ``diffusion/Model.py``: ``DynamicUNet`` (lines 382-517) and the blocks it
is built from -- same class names, constructor / ``forward`` signatures, parameter names and shapes (the 319-entry
``state_dict`` of the default configuration) and initialisation order (``torch.manual_seed`` gives the reference's weights).

As in ``DiffusionFreeGuidence/ModelCondition.py`` the ``torch.nn`` leaf modules are parameter containers only; every
``forward`` issues hand-written gfx950 kernels through ``libhdiff.so``.  Inference only: this tree's trainer
(``diffusion/Diffusion.py:26-180``) is built on pretrained VGG / DINO perceptual losses and is out of scope, so a forward that
would have to record an autograd graph raises.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
from torch import nn
from torch.nn import init

from .. import engine as E
from ..DiffusionFreeGuidence.ModelCondition import (DownSample, Swish, TimeEmbedding, UpSample, _dropout_active, _EagerMixin,
                                                    _inputs, _params_of)

__all__ = ["Swish", "TimeEmbedding", "ConditionalEmbedding", "DownSample", "UpSample", "ResBlock", "DynamicUNet"]


class ConditionalEmbedding(nn.Module, _EagerMixin):
    """Image label -> three stride-2 3x3 convs (no activation) -> global average pool -> Linear -> Swish -> Linear
    (reference diffusion/Model.py:110-168)."""

    def __init__(self, d_model, dim):
        super().__init__()
        channels = d_model // 16
        self.conv1 = nn.Conv2d(in_channels=3, out_channels=channels, kernel_size=3, stride=2, padding=1)
        self.conv2 = nn.Conv2d(in_channels=channels, out_channels=channels * 2, kernel_size=3, stride=2, padding=1)
        self.conv3 = nn.Conv2d(in_channels=channels * 2, out_channels=channels * 4, kernel_size=3, stride=2, padding=1)
        self.pool = nn.AdaptiveAvgPool2d((1, 1))
        self.linear1 = nn.Linear(channels * 4, dim)
        self.activation = Swish()
        self.linear2 = nn.Linear(dim, dim)

    def forward(self, label_tensor):
        label_tensor = _inputs(label_tensor=label_tensor)
        B, _, H, W = (int(v) for v in label_tensor.shape)
        plan = E.Plan(label_tensor.device)
        return self._finish(plan, E.emit_cond_image_embedding(plan, _params_of(self, "m."), "m", label_tensor, B, H, W))


class ResBlock(nn.Module, _EagerMixin):
    """Same block as the first tree's (GN-Swish-Conv +temb +cemb, GN-Swish-Dropout-Conv, + shortcut, optional 8-head
    self-attention replacing h), with ``attn=False`` by default and ``cemb`` optional (reference diffusion/Model.py:267-311)."""

    def __init__(self, in_ch, out_ch, tdim, dropout, attn=False):
        super().__init__()
        self.block1 = nn.Sequential(nn.GroupNorm(32, in_ch), Swish(), nn.Conv2d(in_ch, out_ch, 3, stride=1, padding=1))
        self.temb_proj = nn.Sequential(Swish(), nn.Linear(tdim, out_ch))
        self.cond_proj = nn.Sequential(Swish(), nn.Linear(tdim, out_ch))
        self.block2 = nn.Sequential(nn.GroupNorm(32, out_ch), Swish(), nn.Dropout(dropout),
                                    nn.Conv2d(out_ch, out_ch, 3, stride=1, padding=1))
        self.activate_attn = attn
        self.attn = nn.MultiheadAttention(out_ch, num_heads=8) if attn else nn.Identity()
        self.shortcut = nn.Conv2d(in_ch, out_ch, 1, stride=1, padding=0) if in_ch != out_ch else nn.Identity()
        self.out_ch = out_ch

    def forward(self, x, temb, cemb=None):
        x, temb = _inputs(x=x, temb=temb)
        if cemb is not None:
            cemb = _inputs(cemb=cemb)
        if _dropout_active(self):                  # train-mode nn.Dropout (Model.py:286): the eager path carries its kernels
            from ..autograd import res_block
            return res_block(self, x, None, temb, cemb, True)
        B, _, H, W = (int(v) for v in x.shape)
        plan = E.Plan(x.device)
        out = E.emit_resblock(plan, _params_of(self, "m."), "m", x, None, temb, cemb, self.out_ch, B, H, W, self.activate_attn)
        plan.flush_block_vecs(0, temb, cemb, B)
        return self._finish(plan, out)


class DynamicUNet(nn.Module):
    """Image-conditioned denoiser (reference diffusion/Model.py:382-517): forward(x[B,6,H,W], t[B], labels=None,
    context_zero=True) -> eps[B,3,H,W].  x is the channel concat [conditioning image | noisy image]."""

    MAX_CACHED_PLANS = 2

    def __init__(self, T, ch, ch_mult, num_res_blocks, dropout):
        super().__init__()
        tdim = ch * 4
        self.time_embedding = TimeEmbedding(T, ch, tdim)
        self.cond_embedding = ConditionalEmbedding(ch, tdim)
        self.head = nn.Conv2d(6, ch, kernel_size=3, stride=1, padding=1)
        self.downblocks, self.chs, self.now_ch = self.create_downblocks(ch, ch_mult, num_res_blocks, tdim, dropout)
        self.middleblocks = self.create_middleblocks(self.now_ch, tdim, dropout)
        self.upblocks = self.create_upblocks(ch, ch_mult, num_res_blocks, tdim, dropout)
        self.tail = nn.Sequential(nn.GroupNorm(32, ch), Swish(), nn.Conv2d(ch, 3, kernel_size=3, stride=1, padding=1))
        self.initialize()
        self._shape = E.DynUNetShape(T=T, ch=ch, ch_mult=tuple(ch_mult), num_res_blocks=num_res_blocks)
        self._plans: Dict[tuple, E.DynUNetPlan] = {}
        self._plan_ptrs: Optional[tuple] = None
        self._packed_versions: Dict[tuple, tuple] = {}

    def initialize(self):
        init.xavier_uniform_(self.head.weight)
        init.zeros_(self.head.bias)
        init.xavier_uniform_(self.tail[-1].weight, gain=1e-5)
        init.zeros_(self.tail[-1].bias)

    def create_downblocks(self, ch, ch_mult, num_res_blocks, tdim, dropout):
        blocks, widths, now = nn.ModuleList(), [ch], ch
        for level, mult in enumerate(ch_mult):
            for _ in range(num_res_blocks):
                blocks.append(ResBlock(in_ch=now, out_ch=ch * mult, tdim=tdim, dropout=dropout, attn=False))
                now = ch * mult
                widths.append(now)
            if level != len(ch_mult) - 1:
                blocks.append(DownSample(now))
                widths.append(now)
        return blocks, widths, now

    def create_middleblocks(self, now_ch, tdim, dropout):
        return nn.ModuleList([ResBlock(now_ch, now_ch, tdim, dropout, attn=True) for _ in range(4)])

    def create_upblocks(self, ch, ch_mult, num_res_blocks, tdim, dropout):
        blocks, widths, now = nn.ModuleList(), self.chs.copy(), self.now_ch
        for level in range(len(ch_mult) - 1, -1, -1):
            for _ in range(num_res_blocks):          # not +1: some skip tensors are never consumed (as in the reference)
                blocks.append(ResBlock(in_ch=widths.pop() + now, out_ch=ch * ch_mult[level], tdim=tdim, dropout=dropout,
                                       attn=False))
                now = ch * ch_mult[level]
            if level != 0:
                blocks.append(UpSample(now))
        return blocks

    def dynamic_forward(self, x):
        """Reference :446-474: by the input's mean red vs blue, the even ("underwater") or odd ("atmospheric") middle blocks are
        made trainable and the others frozen.  It only toggles ``requires_grad`` -- the output does not depend on it."""
        underwater = bool(x[:, 2, :, :].mean() > x[:, 0, :, :].mean())
        for i, layer in enumerate(self.middleblocks):
            on = (i % 2 == 0) if underwater else (i % 2 != 0)
            for prm in layer.parameters():
                prm.requires_grad = on

    # -- plan cache (same policy as the first tree's UNet) ---------------------------------------------------------------
    def _param_signature(self):
        ps = list(self.parameters())
        return tuple(p.data_ptr() for p in ps), tuple(p._version for p in ps)

    def plan_for(self, B: int, H: int, W: int, device, context_zero: bool = True) -> E.DynUNetPlan:
        ptrs, versions = self._param_signature()
        if ptrs != self._plan_ptrs:
            self._plans.clear()
            self._packed_versions.clear()
            self._plan_ptrs = ptrs
        key = (B, H, W, str(device), bool(context_zero))
        up = self._plans.pop(key, None)
        if up is None:
            while len(self._plans) >= self.MAX_CACHED_PLANS:
                old = next(iter(self._plans))
                self._plans.pop(old)
                self._packed_versions.pop(old, None)
            up = E.DynUNetPlan(_params_of(self), self._shape, B, H, W, device, bool(context_zero))
        self._plans[key] = up
        if self._packed_versions.get(key) != versions:
            up.plan.pack_weights()
            self._packed_versions[key] = versions
        return up

    def forward(self, x, t, labels=None, context_zero=True):
        x, t = _inputs(x=x, t=t)
        self.dynamic_forward(x)
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            raise NotImplementedError("hdiff: DynamicUNet runs inference only on the HIP path (call it under torch.no_grad()); "
                                      "this tree's trainer depends on pretrained perceptual networks and is out of scope")
        if _dropout_active(self):
            raise NotImplementedError("hdiff: DynamicUNet runs inference only on the HIP path: call .eval() (train-mode dropout "
                                      "belongs to this tree's trainer, which is out of scope)")
        B, Cx, H, W = (int(v) for v in x.shape)
        if Cx != 6:
            raise RuntimeError(f"expected input[{B}, {Cx}, {H}, {W}] to have 6 channels")
        lim = torch.stack([t.min(), t.max()]).tolist()
        if lim[0] < 0 or lim[1] >= self._shape.T:
            raise IndexError("index out of range in self")
        if not context_zero:
            if labels is None:
                raise AttributeError("'NoneType' object has no attribute 'shape'")      # what the reference's conv would hit
            labels = _inputs(labels=labels)
        up = self.plan_for(B, H, W, x.device, context_zero)
        up.cond.copy_(x[:, :3])
        up.y.copy_(x[:, 3:])
        up.t.copy_(t)
        if not context_zero:
            up.label.copy_(labels)
        up.plan.run()
        return up.out.clone()
