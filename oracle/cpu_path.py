"""TEST INFRASTRUCTURE ONLY -- CPU (torch fp32) restatement of the reference CFG-DDPM hot path.

Parity status: PINNED.  Every function here is checked against golden vectors
produced by running the real reference (``oracle/gen_golden.py``; fixtures in
``tests/golden/``; checks in ``tests/test_oracle_golden.py``).

The reference's arithmetic lives in third-party PyTorch operators (reference pin
``pytorch=2.2.1``, ``CLEDiff_bkp.yaml:275``; this container: torch 2.10 CPU kernels).
This restatement is *functional*: it takes a plain ``state_dict`` (the reference's
366-key layout) and spells out each step of the reference's forward with elementary
tensor ops, so that the HIP path can be compared against it op by op.

All ``file:line`` citations are relative to ``/root/reference/``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


@dataclass(frozen=True)
class UNetConfig:
    """Constructor arguments of the reference UNet (ModelCondition.py:214)."""
    T: int
    num_labels: int
    ch: int
    ch_mult: Tuple[int, ...]
    num_res_blocks: int
    dropout: float = 0.0
    num_heads: int = 8          # hard-coded at ModelCondition.py:189
    gn_groups: int = 32         # hard-coded at ModelCondition.py:170,184,249
    gn_eps: float = 1e-5        # nn.GroupNorm default


# ----------------------------------------------------------------------------------------
# Architecture walk (ModelCondition.py:213-252) -- the layer list both the oracle and the
# product's planner are checked against.
# ----------------------------------------------------------------------------------------
@dataclass(frozen=True)
class BlockSpec:
    kind: str            # "res" | "down" | "up"
    prefix: str          # state_dict prefix, e.g. "downblocks.3"
    in_ch: int
    out_ch: int
    attn: bool = False
    takes_skip: bool = False


def architecture(cfg: UNetConfig) -> Tuple[List[BlockSpec], List[BlockSpec], List[BlockSpec], int]:
    """Return (down, middle, up, final_ch) exactly as UNet.__init__ builds them (ModelCondition.py:220-246)."""
    ch = cfg.ch
    chs = [ch]
    now = ch
    down: List[BlockSpec] = []
    for i, mult in enumerate(cfg.ch_mult):
        out = ch * mult
        for _ in range(cfg.num_res_blocks):
            # down ResBlocks do not pass attn= -> default True (ModelCondition.py:226)
            down.append(BlockSpec("res", f"downblocks.{len(down)}", now, out, attn=True))
            now = out
            chs.append(now)
        if i != len(cfg.ch_mult) - 1:
            down.append(BlockSpec("down", f"downblocks.{len(down)}", now, now))
            chs.append(now)
    middle = [BlockSpec("res", "middleblocks.0", now, now, attn=True),
              BlockSpec("res", "middleblocks.1", now, now, attn=False)]
    up: List[BlockSpec] = []
    for i, mult in reversed(list(enumerate(cfg.ch_mult))):
        out = ch * mult
        for _ in range(cfg.num_res_blocks + 1):
            up.append(BlockSpec("res", f"upblocks.{len(up)}", chs.pop() + now, out, attn=False, takes_skip=True))
            now = out
        if i != 0:
            up.append(BlockSpec("up", f"upblocks.{len(up)}", now, now))
    assert not chs
    return down, middle, up, now


# ----------------------------------------------------------------------------------------
# Elementary pieces
# ----------------------------------------------------------------------------------------
def swish(x: Tensor) -> Tensor:
    """Swish.forward, ModelCondition.py:22-24."""
    return x * torch.sigmoid(x)


def sinusoidal_table(T: int, d_model: int) -> Tensor:
    """Initial value of the (trainable) time-embedding table, ModelCondition.py:29-38.

    Row p = [sin(p*f0), cos(p*f0), sin(p*f1), cos(p*f1), ...] with f_i = exp(-(2i/d_model)*ln 1e4).
    """
    assert d_model % 2 == 0
    freqs = torch.exp(-(torch.arange(0, d_model, step=2) / d_model * math.log(10000)))
    ang = torch.arange(T).float()[:, None] * freqs[None, :]
    return torch.stack([torch.sin(ang), torch.cos(ang)], dim=-1).view(T, d_model)


def embed_mlp(idx: Tensor, table: Tensor, w1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor,
              padding_idx: Optional[int] = None) -> Tensor:
    """Embedding -> Linear -> Swish -> Linear (TimeEmbedding ModelCondition.py:40-49, ConditionalEmbedding :56-65).

    For ConditionalEmbedding the table's row 0 is the padding row (kept at zero by nn.Embedding(padding_idx=0), which
    also gives that row no gradient: pass padding_idx=0 when differentiating).
    """
    e = torch.nn.functional.embedding(idx, table, padding_idx=padding_idx)   # gather rows
    h = e @ w1.t() + b1
    h = swish(h)
    return h @ w2.t() + b2


def group_norm(x: Tensor, groups: int, weight: Tensor, bias: Tensor, eps: float) -> Tensor:
    """nn.GroupNorm(32, C) (ModelCondition.py:170,184,249): biased variance over (C/G, H, W) per sample and group."""
    B, C, H, W = x.shape
    xg = x.reshape(B, groups, -1)
    mean = xg.mean(dim=2, keepdim=True)
    var = xg.var(dim=2, unbiased=False, keepdim=True)
    xn = ((xg - mean) * torch.rsqrt(var + eps)).reshape(B, C, H, W)
    return xn * weight[None, :, None, None] + bias[None, :, None, None]


def group_norm_stats(x: Tensor, groups: int, eps: float) -> Tuple[Tensor, Tensor]:
    """(mean, rstd) per (sample, group) -- what the product's gn_stats kernel produces."""
    B = x.shape[0]
    xg = x.reshape(B, groups, -1).double()
    mean = xg.mean(dim=2)
    var = xg.var(dim=2, unbiased=False)
    return mean.float(), torch.rsqrt(var + eps).float()


def mha_self_attention(x_lbc: Tensor, in_w: Tensor, in_b: Tensor, out_w: Tensor, out_b: Tensor,
                       num_heads: int, q_chunk: int = 1024) -> Tensor:
    """nn.MultiheadAttention(C, 8) called as attn(h,h,h) on (L,B,C) (ModelCondition.py:189,204-208).

    Packed in-projection rows are [Wq; Wk; Wv]; per head softmax(q k^T / sqrt(d)) v; out-projection.
    No mask, dropout 0; the averaged attention weights the reference also returns are discarded there.
    Queries are processed in chunks so the L x L score matrix is never held in full (same math).
    """
    L, B, C = x_lbc.shape
    d = C // num_heads
    qkv = x_lbc @ in_w.t() + in_b                      # (L,B,3C)
    q, k, v = qkv.split(C, dim=2)

    def heads(z: Tensor) -> Tensor:                    # (L,B,C) -> (B*heads, L, d)
        return z.reshape(L, B * num_heads, d).transpose(0, 1)

    q, k, v = heads(q) * (1.0 / math.sqrt(d)), heads(k), heads(v)
    outs = []
    for s in range(0, L, q_chunk):
        w = torch.softmax(torch.bmm(q[:, s:s + q_chunk], k.transpose(1, 2)), dim=-1)
        outs.append(torch.bmm(w, v))
    o = torch.cat(outs, dim=1)                          # (B*heads, L, d)
    o = o.transpose(0, 1).reshape(L, B, C)
    return o @ out_w.t() + out_b


def attn_block(sd: SD, p: str, x: Tensor) -> Tensor:
    """AttnBlock.forward, ModelCondition.py:102-120 (dead code in the reference's UNet): GroupNorm(32) WITHOUT Swish, 1x1
    q / k / v, w = softmax(q^T k * C^-1/2) over all positions (one head of width C), h = w v, 1x1 proj, x + h."""
    B, C, H, W = x.shape
    h = group_norm(x, 32, sd[f"{p}.group_norm.weight"], sd[f"{p}.group_norm.bias"], 1e-5)
    conv1 = lambda name: torch.nn.functional.conv2d(h, sd[f"{p}.{name}.weight"], sd[f"{p}.{name}.bias"])
    q = conv1("proj_q").permute(0, 2, 3, 1).reshape(B, H * W, C)
    k = conv1("proj_k").reshape(B, C, H * W)
    v = conv1("proj_v").permute(0, 2, 3, 1).reshape(B, H * W, C)
    w = torch.softmax(torch.bmm(q, k) * (int(C) ** (-0.5)), dim=-1)
    o = torch.bmm(w, v).reshape(B, H, W, C).permute(0, 3, 1, 2)
    return x + torch.nn.functional.conv2d(o, sd[f"{p}.proj.weight"], sd[f"{p}.proj.bias"])


def res_block(sd: SD, p: str, x: Tensor, temb: Tensor, cemb: Tensor, cfg: UNetConfig, attn: bool,
              training: bool = False, drop_mask: Optional[Tensor] = None) -> Tensor:
    """ResBlock.forward, ModelCondition.py:196-211.

    ``drop_mask`` (already scaled by 1/(1-p)) injects the nn.Dropout mask (ModelCondition.py:185) for training parity.
    """
    g, eps = cfg.gn_groups, cfg.gn_eps
    h = swish(group_norm(x, g, sd[f"{p}.block1.0.weight"], sd[f"{p}.block1.0.bias"], eps))
    h = F.conv2d(h, sd[f"{p}.block1.2.weight"], sd[f"{p}.block1.2.bias"], stride=1, padding=1)
    h = h + (swish(temb) @ sd[f"{p}.temb_proj.1.weight"].t() + sd[f"{p}.temb_proj.1.bias"])[:, :, None, None]
    h = h + (swish(cemb) @ sd[f"{p}.cond_proj.1.weight"].t() + sd[f"{p}.cond_proj.1.bias"])[:, :, None, None]
    h = swish(group_norm(h, g, sd[f"{p}.block2.0.weight"], sd[f"{p}.block2.0.bias"], eps))
    if training and drop_mask is not None:
        h = h * drop_mask
    h = F.conv2d(h, sd[f"{p}.block2.3.weight"], sd[f"{p}.block2.3.bias"], stride=1, padding=1)
    if f"{p}.shortcut.weight" in sd:
        sc = F.conv2d(x, sd[f"{p}.shortcut.weight"], sd[f"{p}.shortcut.bias"])
    else:
        sc = x
    h = h + sc
    if attn:
        B, C, H, W = h.shape
        seq = h.reshape(B, C, H * W).permute(2, 0, 1)           # (L,B,C); no pre-norm, no residual (:204-208)
        seq = mha_self_attention(seq, sd[f"{p}.attn.in_proj_weight"], sd[f"{p}.attn.in_proj_bias"],
                                 sd[f"{p}.attn.out_proj.weight"], sd[f"{p}.attn.out_proj.bias"], cfg.num_heads)
        h = seq.permute(1, 2, 0).reshape(B, C, H, W)
    return h


def down_sample(sd: SD, p: str, x: Tensor) -> Tensor:
    """DownSample.forward, ModelCondition.py:74-76: Conv3x3 s2 p1 + Conv5x5 s2 p2, summed."""
    return (F.conv2d(x, sd[f"{p}.c1.weight"], sd[f"{p}.c1.bias"], stride=2, padding=1)
            + F.conv2d(x, sd[f"{p}.c2.weight"], sd[f"{p}.c2.bias"], stride=2, padding=2))


def up_sample(sd: SD, p: str, x: Tensor) -> Tensor:
    """UpSample.forward, ModelCondition.py:85-89: ConvTranspose2d(5, s2, p2, op1) then Conv3x3."""
    x = F.conv_transpose2d(x, sd[f"{p}.t.weight"], sd[f"{p}.t.bias"], stride=2, padding=2, output_padding=1)
    return F.conv2d(x, sd[f"{p}.c.weight"], sd[f"{p}.c.bias"], stride=1, padding=1)


def unet_forward(sd: SD, cfg: UNetConfig, x: Tensor, t: Tensor, labels: Tensor,
                 taps: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """UNet.forward (eval mode), ModelCondition.py:255-276.  ``taps`` collects per-layer outputs for op-level parity."""
    down, middle, up, final_ch = architecture(cfg)
    temb = embed_mlp(t, sd["time_embedding.timembedding.0.weight"],
                     sd["time_embedding.timembedding.1.weight"], sd["time_embedding.timembedding.1.bias"],
                     sd["time_embedding.timembedding.3.weight"], sd["time_embedding.timembedding.3.bias"])
    cemb = embed_mlp(labels, sd["cond_embedding.condEmbedding.0.weight"],
                     sd["cond_embedding.condEmbedding.1.weight"], sd["cond_embedding.condEmbedding.1.bias"],
                     sd["cond_embedding.condEmbedding.3.weight"], sd["cond_embedding.condEmbedding.3.bias"], padding_idx=0)
    if taps is not None:
        taps["temb"], taps["cemb"] = temb, cemb
    h = F.conv2d(x, sd["head.weight"], sd["head.bias"], stride=1, padding=1)
    if taps is not None:
        taps["head"] = h
    hs = [h]
    for b in down:
        h = res_block(sd, b.prefix, h, temb, cemb, cfg, b.attn) if b.kind == "res" else down_sample(sd, b.prefix, h)
        hs.append(h)
        if taps is not None:
            taps[b.prefix] = h
    for b in middle:
        h = res_block(sd, b.prefix, h, temb, cemb, cfg, b.attn)
        if taps is not None:
            taps[b.prefix] = h
    for b in up:
        if b.kind == "res":
            h = torch.cat([h, hs.pop()], dim=1)
            h = res_block(sd, b.prefix, h, temb, cemb, cfg, b.attn)
        else:
            h = up_sample(sd, b.prefix, h)
        if taps is not None:
            taps[b.prefix] = h
    assert not hs
    h = swish(group_norm(h, cfg.gn_groups, sd["tail.0.weight"], sd["tail.0.bias"], cfg.gn_eps))
    return F.conv2d(h, sd["tail.2.weight"], sd["tail.2.bias"], stride=1, padding=1)


# ----------------------------------------------------------------------------------------
# Diffusion process (DiffusionCondition.py)
# ----------------------------------------------------------------------------------------
def extract(v: Tensor, t: Tensor, x_shape: Sequence[int]) -> Tensor:
    """extract, DiffusionCondition.py:9-16: gather on the float64 buffer, THEN cast to fp32, view [B,1,1,...]."""
    out = torch.gather(v, index=t, dim=0).float()
    return out.view([t.shape[0]] + [1] * (len(x_shape) - 1))


def trainer_schedule(beta_1: float, beta_T: float, T: int) -> Dict[str, Tensor]:
    """GaussianDiffusionTrainer.__init__ buffers, DiffusionCondition.py:26-35 (fp32 linspace, then float64)."""
    betas = torch.linspace(beta_1, beta_T, T).double()
    alphas_bar = torch.cumprod(1.0 - betas, dim=0)
    return {"betas": betas, "sqrt_alphas_bar": torch.sqrt(alphas_bar),
            "sqrt_one_minus_alphas_bar": torch.sqrt(1.0 - alphas_bar)}


def sampler_schedule(beta_1: float, beta_T: float, T: int) -> Dict[str, Tensor]:
    """GaussianDiffusionSampler.__init__ buffers, DiffusionCondition.py:60-66."""
    betas = torch.linspace(beta_1, beta_T, T).double()
    alphas = 1.0 - betas
    alphas_bar = torch.cumprod(alphas, dim=0)
    alphas_bar_prev = F.pad(alphas_bar, [1, 0], value=1)[:T]
    coeff1 = torch.sqrt(1.0 / alphas)
    return {"betas": betas, "coeff1": coeff1,
            "coeff2": coeff1 * (1.0 - alphas) / torch.sqrt(1.0 - alphas_bar),
            "posterior_var": betas * (1.0 - alphas_bar_prev) / (1.0 - alphas_bar)}


def sampler_variance_table(sched: Dict[str, Tensor]) -> Tensor:
    """The 'fixed-large' variance row used at every step, DiffusionCondition.py:74."""
    return torch.cat([sched["posterior_var"][1:2], sched["betas"][1:]])


def q_sample(sched: Dict[str, Tensor], x_0: Tensor, t: Tensor, noise: Tensor) -> Tensor:
    """x_t of Algorithm 1, DiffusionCondition.py:43-44."""
    return (extract(sched["sqrt_alphas_bar"], t, x_0.shape) * x_0
            + extract(sched["sqrt_one_minus_alphas_bar"], t, x_0.shape) * noise)


def trainer_loss(sd: SD, cfg: UNetConfig, sched: Dict[str, Tensor], x_0: Tensor, labels: Tensor,
                 t: Tensor, noise: Tensor) -> Tensor:
    """GaussianDiffusionTrainer.forward with injected (t, noise), DiffusionCondition.py:37-46 (eval-mode model)."""
    x_t = q_sample(sched, x_0, t, noise)
    return (unet_forward(sd, cfg, x_t, t, labels) - noise) ** 2


def cfg_eps(eps_cond: Tensor, eps_uncond: Tensor, w: float) -> Tensor:
    """Classifier-free-guidance combine, DiffusionCondition.py:78."""
    return (1.0 + w) * eps_cond - w * eps_uncond


def posterior_mean(sched: Dict[str, Tensor], x_t: Tensor, t: Tensor, eps: Tensor) -> Tensor:
    """predict_xt_prev_mean_from_eps, DiffusionCondition.py:68-70."""
    assert x_t.shape == eps.shape
    return extract(sched["coeff1"], t, x_t.shape) * x_t - extract(sched["coeff2"], t, x_t.shape) * eps


def denoise_step(sd: SD, cfg: UNetConfig, sched: Dict[str, Tensor], w: float, x_t: Tensor, time_step: int,
                 labels: Tensor, noise: Optional[Tensor]) -> Tensor:
    """One iteration of GaussianDiffusionSampler.forward's loop, DiffusionCondition.py:87-96 (pre-clip x_{t-1})."""
    B = x_t.shape[0]
    t = torch.full((B,), time_step, dtype=torch.long)
    var = extract(sampler_variance_table(sched), t, x_t.shape)
    eps = unet_forward(sd, cfg, x_t, t, labels)
    non_eps = unet_forward(sd, cfg, x_t, t, torch.zeros_like(labels))
    mean = posterior_mean(sched, x_t, t, cfg_eps(eps, non_eps, w))
    if time_step > 0:
        assert noise is not None
        x_prev = mean + torch.sqrt(var) * noise
    else:
        x_prev = mean
    assert torch.isnan(x_prev).int().sum() == 0, "nan in tensor."
    return x_prev


def sampler_forward(sd: SD, cfg: UNetConfig, beta_1: float, beta_T: float, T: int, w: float, x_T: Tensor,
                    labels: Tensor, noises: Sequence[Optional[Tensor]],
                    trajectory: Optional[List[Tensor]] = None) -> Tensor:
    """GaussianDiffusionSampler.forward with injected per-step noise, DiffusionCondition.py:82-98.

    ``noises[time_step]`` is the z used at that step (ignored at time_step 0, where the reference uses 0).
    """
    sched = sampler_schedule(beta_1, beta_T, T)
    x_t = x_T
    for time_step in reversed(range(T)):
        x_t = denoise_step(sd, cfg, sched, w, x_t, time_step, labels, noises[time_step])
        if trajectory is not None:
            trajectory.append(x_t)
    return torch.clip(x_t, -1, 1)


# ----------------------------------------------------------------------------------------
# Image-quality checks used for end-to-end parity (PSNR on x*0.5+0.5, data range 1)
# ----------------------------------------------------------------------------------------
def psnr(a: Tensor, b: Tensor, data_range: float = 1.0) -> float:
    mse = torch.mean((a.double() - b.double()) ** 2).item()
    if mse == 0.0:
        return float("inf")
    return 10.0 * math.log10(data_range ** 2 / mse)
