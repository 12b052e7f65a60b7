"""TEST INFRASTRUCTURE ONLY -- CPU (torch fp32) restatement of the reference's second tree: the image-conditioned
``DynamicUNet`` (diffusion/Model.py:382-517) and its ancestral / DDIM sampler (diffusion/Diffusion.py:182-269).

Parity status: PINNED by golden vectors produced by the real reference (``oracle/gen_golden_b.py``: ``diffusion/Model.py``
imports by file path; ``diffusion/Diffusion.py`` needs cv2 / lpips / Loss.loss, so only the text of ``extract`` and
``class GaussianDiffusionSampler`` is compiled, in place -- see ``oracle/reference_loader.py``).  Fixtures: ``tests/golden/dyn_*``;
checks: ``tests/test_oracle_golden_b.py``.  All ``file:line`` citations are relative to ``/root/reference/``.

Behaviour restated exactly as written there, including what looks unintended:
  * the sampler always calls ``model(input, t)`` -- ``labels=None, context_zero=True`` -- so the conditional embedding
    is a zero vector (Model.py:482-483) and the classifier-free-guidance combine of the DDIM branch (Diffusion.py:254-257)
    mixes two evaluations of the same function: eps_u + s * (eps - eps_u) with eps == eps_u;
  * the up path has ``num_res_blocks`` (not +1) blocks per level, so skip tensors of the wrong resolution are popped and
    resized with nearest-neighbour interpolation (Model.py:500-506); some skips are never used;
  * DDIM time steps are laid out over a hard-coded 1000 (Diffusion.py:243-245) and ``alphas_bar`` is read at ``t + 1``
    (Diffusion.py:250-251), with ``alphas_bar[0]`` as the final "next" value.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from . import cpu_path as O

Tensor = torch.Tensor
SD = Dict[str, Tensor]


@dataclass(frozen=True)
class DynUNetConfig:
    """Constructor arguments of DynamicUNet (Model.py:383)."""
    T: int
    ch: int
    ch_mult: Tuple[int, ...]
    num_res_blocks: int
    dropout: float = 0.0
    num_heads: int = 8          # Model.py:291
    gn_groups: int = 32         # Model.py:272,284,396
    gn_eps: float = 1e-5


def layout(cfg: DynUNetConfig) -> Tuple[List[Tuple[str, str, int]], List[str], List[Tuple[str, str, int]]]:
    """(down, middle, up) block lists as (kind, prefix, out_ch), Model.py:409-444."""
    down, up = [], []
    now = cfg.ch
    n = 0
    for i, mult in enumerate(cfg.ch_mult):
        for _ in range(cfg.num_res_blocks):
            down.append(("res", f"downblocks.{n}", cfg.ch * mult)); n += 1
            now = cfg.ch * mult
        if i != len(cfg.ch_mult) - 1:
            down.append(("down", f"downblocks.{n}", now)); n += 1
    middle = [f"middleblocks.{i}" for i in range(4)]
    n = 0
    for i, mult in reversed(list(enumerate(cfg.ch_mult))):
        for _ in range(cfg.num_res_blocks):
            up.append(("res", f"upblocks.{n}", cfg.ch * mult)); n += 1
        if i != 0:
            up.append(("up", f"upblocks.{n}", cfg.ch * mult)); n += 1
    return down, middle, up


def cond_image_embedding(sd: SD, img: Tensor) -> Tensor:
    """ConditionalEmbedding.forward, Model.py:135-166: three stride-2 3x3 convs (no activation), global mean, MLP."""
    p = "cond_embedding"
    x = F.conv2d(img, sd[f"{p}.conv1.weight"], sd[f"{p}.conv1.bias"], stride=2, padding=1)
    x = F.conv2d(x, sd[f"{p}.conv2.weight"], sd[f"{p}.conv2.bias"], stride=2, padding=1)
    x = F.conv2d(x, sd[f"{p}.conv3.weight"], sd[f"{p}.conv3.bias"], stride=2, padding=1)
    x = x.mean(dim=(2, 3))
    h = O.swish(x @ sd[f"{p}.linear1.weight"].t() + sd[f"{p}.linear1.bias"])
    return h @ sd[f"{p}.linear2.weight"].t() + sd[f"{p}.linear2.bias"]


def resize_nearest(x: Tensor, size: Sequence[int]) -> Tensor:
    """F.interpolate(mode='nearest') spelled out: src = min(floor(dst * in / out), in - 1), scale formed in fp32."""
    H, W = x.shape[2:]
    OH, OW = int(size[0]), int(size[1])
    sy, sx = torch.tensor(H, dtype=torch.float32) / OH, torch.tensor(W, dtype=torch.float32) / OW
    iy = torch.clamp(torch.floor(torch.arange(OH, dtype=torch.float32) * sy).long(), max=H - 1)
    ix = torch.clamp(torch.floor(torch.arange(OW, dtype=torch.float32) * sx).long(), max=W - 1)
    return x[:, :, iy][:, :, :, ix]


def dyn_unet_forward(sd: SD, cfg: DynUNetConfig, x: Tensor, t: Tensor, labels: Optional[Tensor] = None,
                     context_zero: bool = True, taps: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """DynamicUNet.forward (eval), Model.py:475-515.  ``dynamic_forward`` (:446-474) only toggles requires_grad."""
    rc = O.UNetConfig(T=cfg.T, num_labels=1, ch=cfg.ch, ch_mult=cfg.ch_mult, num_res_blocks=cfg.num_res_blocks,
                      num_heads=cfg.num_heads, gn_groups=cfg.gn_groups, gn_eps=cfg.gn_eps)
    down, middle, up = layout(cfg)
    tp = "time_embedding.timembedding"
    temb = O.embed_mlp(t, sd[f"{tp}.0.weight"], sd[f"{tp}.1.weight"], sd[f"{tp}.1.bias"], sd[f"{tp}.3.weight"], sd[f"{tp}.3.bias"])
    cemb = torch.zeros_like(temb) if context_zero else cond_image_embedding(sd, labels)
    if taps is not None:
        taps["temb"], taps["cemb"] = temb, cemb
    h = F.conv2d(x, sd["head.weight"], sd["head.bias"], stride=1, padding=1)
    hs = [h]
    if taps is not None:
        taps["head"] = h
    for kind, p, _ in down:
        h = O.res_block(sd, p, h, temb, cemb, rc, attn=False) if kind == "res" else O.down_sample(sd, p, h)
        hs.append(h)
        if taps is not None:
            taps[p] = h
    for p in middle:
        h = O.res_block(sd, p, h, temb, cemb, rc, attn=True)
        if taps is not None:
            taps[p] = h
    for kind, p, _ in up:
        if kind == "res":
            skip = hs.pop()
            if skip.shape[2:] != h.shape[2:]:
                skip = resize_nearest(skip, h.shape[2:])
            h = O.res_block(sd, p, torch.cat([h, skip], dim=1), temb, cemb, rc, attn=False)
        else:
            h = O.up_sample(sd, p, h)
        if taps is not None:
            taps[p] = h
    h = O.swish(O.group_norm(h, cfg.gn_groups, sd["tail.0.weight"], sd["tail.0.bias"], cfg.gn_eps))
    if taps is not None:
        taps["tail_in"] = h
    return F.conv2d(h, sd["tail.2.weight"], sd["tail.2.bias"], stride=1, padding=1)


# ----------------------------------------------------------------------------------------
# Sampler (Diffusion.py:182-269)
# ----------------------------------------------------------------------------------------
def sampler_schedule(beta_1: float, beta_T: float, T: int) -> Dict[str, Tensor]:
    """GaussianDiffusionSampler.__init__, Diffusion.py:189-200 (float64 throughout)."""
    betas = torch.linspace(beta_1, beta_T, T).double()
    alphas = 1.0 - betas
    alphas_bar = torch.cumprod(alphas, dim=0)
    alphas_bar_prev = F.pad(alphas_bar, [1, 0], value=1)[:T]
    coeff1 = torch.sqrt(1.0 / alphas)
    return {"betas": betas, "alphas_bar": alphas_bar, "coeff1": coeff1,
            "coeff2": coeff1 * (1.0 - alphas) / torch.sqrt(1.0 - alphas_bar),
            "posterior_var": betas * (1.0 - alphas_bar_prev) / (1.0 - alphas_bar)}


def ddim_sequence(ddim_step: int) -> List[Tuple[int, int]]:
    """(t, t_next) pairs in sampling order, Diffusion.py:243-247 (the 1000 is literal there)."""
    step = int(1000 / ddim_step)
    seq = range(0, 1000, step)
    seq_next = [-1] + list(seq[:-1])
    return list(zip(reversed(seq), reversed(seq_next)))


def ddim_coefficients(sched: Dict[str, Tensor], ddim_step: int) -> Tensor:
    """Per step [sqrt(1-at), sqrt(at), sqrt(at_next), c2] in fp32, formed with the reference's ops (Diffusion.py:250-262)."""
    rows = []
    for i, j in ddim_sequence(ddim_step):
        at = sched["alphas_bar"][i + 1].float()
        at_next = sched["alphas_bar"][j + 1].float()
        c1 = 0 * ((1 - at / at_next) * (1 - at_next) / (1 - at)).sqrt()
        c2 = ((1 - at_next) - c1 ** 2).sqrt()
        rows.append(torch.stack([(1 - at).sqrt(), at.sqrt(), at_next.sqrt(), c2]))
    return torch.stack(rows)


def sampler_forward(sd: SD, cfg: DynUNetConfig, beta_1: float, beta_T: float, T: int, input_image: Tensor, y_T: Tensor,
                    noise_by_step: Optional[List[Tensor]] = None, ddim: bool = False, unconditional_guidance_scale: float = 1,
                    ddim_step: Optional[int] = None, trajectory: Optional[List[Tensor]] = None) -> Tensor:
    """GaussianDiffusionSampler.forward, Diffusion.py:217-269, with the random draws (``y_T`` :226/:239 and the per-step
    ``randn_like`` :232) handed in.  ``trajectory`` collects the pre-clip y_t after every step."""
    sched = sampler_schedule(beta_1, beta_T, T)
    img = input_image.float() / 255.0
    B = img.shape[0]
    y = y_T
    if not ddim:
        var_tab = torch.cat([sched["posterior_var"][1:2], sched["betas"][1:]])
        for k, time_step in enumerate(reversed(range(T))):
            t = torch.full((B,), time_step, dtype=torch.long)
            eps = dyn_unet_forward(sd, cfg, torch.cat([img, y], dim=1).float(), t)
            mean = O.extract(sched["coeff1"], t, y.shape) * y - O.extract(sched["coeff2"], t, y.shape) * eps
            if time_step > 0:
                y = mean + torch.sqrt(O.extract(var_tab, t, y.shape)) * noise_by_step[k]
            else:
                y = mean + torch.sqrt(O.extract(var_tab, t, y.shape)) * 0
            if trajectory is not None:
                trajectory.append(y)
        return torch.clip(y, -1, 1)
    for i, j in ddim_sequence(ddim_step):
        t = torch.full((B,), i, dtype=torch.long)
        at = O.extract(sched["alphas_bar"], t + 1, y.shape)
        at_next = O.extract(sched["alphas_bar"], torch.full((B,), j, dtype=torch.long) + 1, y.shape)
        eps = dyn_unet_forward(sd, cfg, torch.cat([img, y], dim=1).float(), t)
        if unconditional_guidance_scale != 1:
            eps_u = dyn_unet_forward(sd, cfg, torch.cat([img, y], dim=1).float(), t, context_zero=True)
            eps = eps_u + unconditional_guidance_scale * (eps - eps_u)
        y0 = (y - eps * (1 - at).sqrt()) / at.sqrt()
        c1 = 0 * ((1 - at / at_next) * (1 - at_next) / (1 - at)).sqrt()
        c2 = ((1 - at_next) - c1 ** 2).sqrt()
        y = at_next.sqrt() * y0 + c2 * eps          # + c1 * randn == + 0
        if trajectory is not None:
            trajectory.append(y)
    return torch.clip(y, -1, 1)
