"""TEST INFRASTRUCTURE ONLY -- generate the golden vectors that pin the oracle and the HIP path.

Run in the build container (needs ``/root/reference``):

    python -m oracle.gen_golden            # writes tests/golden/*.npz + *.json

What it does: loads the REAL reference classes in place (``oracle/reference_loader.py``), runs them on
seeded inputs on the CPU, and stores inputs + expected outputs (never any reference source).  The fixtures are
data only; they travel to the GPU box, the reference does not.

Fixtures:
  G1 schedules.npz        Trainer/Sampler float64 buffers for three (beta_1, beta_T, T)
  G2 modules.npz          Swish, TimeEmbedding, ConditionalEmbedding, DownSample, UpSample, ResBlock x3 (+weights)
  G2b attnblock.npz       AttnBlock (dead code in the reference's UNet) at 64 channels (flash path) and 128 (wide-head path)
                          [python -m oracle.gen_golden g2b]
  G3 unet_small.npz       small UNet (ch=32, ch_mult=[1,2], nrb=1): state_dict, inputs, eps @16^2 and @32^2, per-layer taps
  G3c unet_wide.npz       four levels, ch=32 ch_mult=[1,2,3,4] (attention heads of 4 / 8 / 12 / 16 channels) @32^2 B=2: seed recipe,
                          weight checksums, eps; trainer loss + gradients of ten tensors      [python -m oracle.gen_golden g3c]
  G4 unet_default64.npz   default UNet (ch=128,[1,2,2,2],nrb=2) @64^2 B=1: seed recipe, weight checksums, input, eps
  G4b unet_default128.npz default UNet @128^2 B=1 (BASELINE config C2's shape; the reference materialises 8 x 16384^2 scores
                          = 8.6 GB per attention block): input, eps for label 1 and label 0   [python -m oracle.gen_golden g4b]
  G5b sampler_default128.npz  the REAL reference sampler on the default UNet at 128x128 (config C2's shape), T = 3, w = 1.8,
                          two independent B = 1 runs (the reference's batch entries do not interact), every draw recorded
                          [python -m oracle.gen_golden g5b; ~8 min, ~25 GB]
  G5 sampler_small.npz    T=8 sampler on the small UNet, w in {0, 1.8}: x_T, per-step noise, pre-clip trajectory, output
  G6 trainer_small.npz    Trainer loss with recorded (t, noise); grads of named params; one clipped AdamW step
  G6b trainer_default64.npz  the DEFAULT model's trainer pass at 64x64, B = 2 (attention backward over L = 4096, d_head 16 / 32):
                          loss, 19 small gradient tensors + 4 row slices, total norm         [python -m oracle.gen_golden g6b]
  G7 state_dict_default.json   the 366 (name, shape) pairs of the default UNet
  G8 lr_schedule.json     GradualWarmupScheduler + CosineAnnealingLR learning-rate sequence (Scheduler.py)
"""
from __future__ import annotations

import contextlib
import io
import json
import os
import sys

import numpy as np
import torch

from . import reference_loader as RL

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

SMALL = dict(T=8, num_labels=3, ch=32, ch_mult=[1, 2], num_res_blocks=1, dropout=0.0)
DEFAULT = dict(T=1000, num_labels=10, ch=128, ch_mult=[1, 2, 2, 2], num_res_blocks=2, dropout=0.15)
WIDE = dict(T=16, num_labels=3, ch=32, ch_mult=[1, 2, 3, 4], num_res_blocks=1, dropout=0.0)   # head widths 4, 8, 12, 16
SMALL_SEED = 1234
DEFAULT_SEED = 0
WIDE_SEED = 31


def _np(t):
    return t.detach().cpu().numpy().copy()      # copy: later in-place updates (clip_grad_norm_, optimizer) must not leak in


def _sd_np(sd, prefix="sd/"):
    return {prefix + k: _np(v) for k, v in sd.items()}


class _Recorder:
    """Record the tensors the reference draws from torch's RNG / checks for NaN, without touching its code."""

    def __init__(self):
        self.randn, self.randint, self.isnan_in = [], [], []
        self._orig = {}

    def __enter__(self):
        self._orig = dict(randn_like=torch.randn_like, randint=torch.randint, isnan=torch.isnan)

        def randn_like(x, *a, **k):
            r = self._orig["randn_like"](x, *a, **k)
            self.randn.append(r.clone())
            return r

        def randint(*a, **k):
            r = self._orig["randint"](*a, **k)
            self.randint.append(r.clone())
            return r

        def isnan(x):
            self.isnan_in.append(x.detach().clone())
            return self._orig["isnan"](x)

        torch.randn_like, torch.randint, torch.isnan = randn_like, randint, isnan
        return self

    def __exit__(self, *exc):
        torch.randn_like = self._orig["randn_like"]
        torch.randint = self._orig["randint"]
        torch.isnan = self._orig["isnan"]


def gen_schedules(RD):
    out = {}
    for i, (b1, bT, T) in enumerate([(1e-4, 0.028, 50), (1e-4, 0.028, 500), (1e-4, 0.02, 1000)]):
        tr = RD.GaussianDiffusionTrainer(torch.nn.Identity(), b1, bT, T)
        sa = RD.GaussianDiffusionSampler(torch.nn.Identity(), b1, bT, T, w=1.8)
        out[f"cfg{i}"] = np.array([b1, bT, T], dtype=np.float64)
        for n in ("betas", "sqrt_alphas_bar", "sqrt_one_minus_alphas_bar"):
            out[f"cfg{i}/trainer/{n}"] = _np(getattr(tr, n))
        for n in ("betas", "coeff1", "coeff2", "posterior_var"):
            out[f"cfg{i}/sampler/{n}"] = _np(getattr(sa, n))
        # extract(): float64 gather -> fp32 -> view
        t = torch.tensor([0, 1, T // 2, T - 1])
        out[f"cfg{i}/extract_t"] = _np(t)
        out[f"cfg{i}/extract_coeff2"] = _np(RD.extract(sa.coeff2, t, (4, 3, 8, 8)))
    np.savez_compressed(os.path.join(OUT, "schedules.npz"), **out)


def gen_modules(RM):
    g = torch.Generator().manual_seed(7)
    out = {}
    x = torch.randn(3, 5, 7, generator=g) * 3
    out["swish/x"], out["swish/y"] = _np(x), _np(RM.Swish()(x))

    torch.manual_seed(11)
    te = RM.TimeEmbedding(20, 32, 128).eval()
    t = torch.tensor([0, 1, 7, 19])
    out["temb/t"], out["temb/y"] = _np(t), _np(te(t))
    out["temb/table_T20_d32"] = _np(te.timembedding[0].weight)
    out.update(_sd_np(te.state_dict(), "temb/sd/"))

    ce = RM.ConditionalEmbedding(4, 32, 128).eval()
    lab = torch.tensor([0, 1, 4, 2, 0])
    out["cemb/labels"], out["cemb/y"] = _np(lab), _np(ce(lab))
    out.update(_sd_np(ce.state_dict(), "cemb/sd/"))

    ds = RM.DownSample(32).eval()
    x = torch.randn(2, 32, 12, 12, generator=g)
    out["down/x"], out["down/y"] = _np(x), _np(ds(x, None, None))
    out.update(_sd_np(ds.state_dict(), "down/sd/"))

    us = RM.UpSample(32).eval()
    x = torch.randn(2, 32, 6, 6, generator=g)
    out["up/x"], out["up/y"] = _np(x), _np(us(x, None, None))
    out.update(_sd_np(us.state_dict(), "up/sd/"))

    # ResBlocks: (in, out, attn); tdim 64; C multiple of 32 (GroupNorm) and of 8 (heads)
    for name, (cin, cout, attn, hw) in {"rb_attn": (32, 32, True, 8), "rb_sc": (32, 64, False, 8),
                                         "rb_sc_attn": (96, 64, True, 6)}.items():
        rb = RM.ResBlock(cin, cout, 64, 0.0, attn=attn).eval()
        # MHA biases are zero-initialised by torch; make them non-trivial so the fixture exercises them
        with torch.no_grad():
            if attn:
                rb.attn.in_proj_bias.copy_(torch.randn(3 * cout, generator=g) * 0.1)
                rb.attn.out_proj.bias.copy_(torch.randn(cout, generator=g) * 0.1)
        x = torch.randn(2, cin, hw, hw, generator=g)
        temb = torch.randn(2, 64, generator=g)
        cemb = torch.randn(2, 64, generator=g)
        out[f"{name}/x"], out[f"{name}/temb"], out[f"{name}/cemb"] = _np(x), _np(temb), _np(cemb)
        out[f"{name}/y"] = _np(rb(x, temb, cemb))
        out[f"{name}/meta"] = np.array([cin, cout, int(attn), hw])
        out.update(_sd_np(rb.state_dict(), f"{name}/sd/"))
    np.savez_compressed(os.path.join(OUT, "modules.npz"), **out)


def gen_attnblock(RM):
    g = torch.Generator().manual_seed(17)
    out = {}
    for name, (cin, hw) in {"c64": (64, 12), "c128": (128, 10)}.items():
        torch.manual_seed(40 + cin)
        ab = RM.AttnBlock(cin).eval()
        with torch.no_grad():
            ab.group_norm.weight.add_(torch.randn(cin, generator=g) * 0.2)      # default affine is the identity: exercise it
            ab.group_norm.bias.add_(torch.randn(cin, generator=g) * 0.2)
        x = torch.randn(2, cin, hw, hw + 1, generator=g) * 1.5
        with torch.no_grad():
            out[f"{name}/x"], out[f"{name}/y"] = _np(x), _np(ab(x))
        out[f"{name}/meta"] = np.array([cin, hw, hw + 1])
        out.update(_sd_np(ab.state_dict(), f"{name}/sd/"))
    np.savez_compressed(os.path.join(OUT, "attnblock.npz"), **out)


def _small_model(RM):
    torch.manual_seed(SMALL_SEED)
    m = RM.UNet(**SMALL).eval()
    g = torch.Generator().manual_seed(SMALL_SEED + 1)
    with torch.no_grad():
        # torch zero-initialises MHA biases: perturb them (and GN affine) so they are exercised
        for n, p in m.named_parameters():
            if n.endswith("in_proj_bias") or n.endswith("out_proj.bias"):
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)
            if ".block1.0." in n or ".block2.0." in n or n.startswith("tail.0."):
                p.add_(torch.randn(p.shape, generator=g) * 0.1)
    return m


def gen_unet_small(RM):
    m = _small_model(RM)
    out = _sd_np(m.state_dict())
    out["cfg_json"] = np.frombuffer(json.dumps(SMALL).encode(), dtype=np.uint8)
    g = torch.Generator().manual_seed(99)
    for S in (16, 32):
        x = torch.randn(2, 3, S, S, generator=g)
        t = torch.tensor([3, 7])
        labels = torch.tensor([2, 0])
        taps = {}
        hooks = []
        if S == 16:
            for name, mod in list(m.downblocks.named_children()):
                hooks.append(mod.register_forward_hook(lambda _m, _i, o, n=f"downblocks.{name}": taps.__setitem__(n, o)))
            for name, mod in list(m.middleblocks.named_children()):
                hooks.append(mod.register_forward_hook(lambda _m, _i, o, n=f"middleblocks.{name}": taps.__setitem__(n, o)))
            for name, mod in list(m.upblocks.named_children()):
                hooks.append(mod.register_forward_hook(lambda _m, _i, o, n=f"upblocks.{name}": taps.__setitem__(n, o)))
            hooks.append(m.head.register_forward_hook(lambda _m, _i, o: taps.__setitem__("head", o)))
            hooks.append(m.time_embedding.register_forward_hook(lambda _m, _i, o: taps.__setitem__("temb", o)))
            hooks.append(m.cond_embedding.register_forward_hook(lambda _m, _i, o: taps.__setitem__("cemb", o)))
        with torch.no_grad():
            y = m(x, t, labels)
        for h in hooks:
            h.remove()
        out[f"s{S}/x"], out[f"s{S}/t"], out[f"s{S}/labels"], out[f"s{S}/eps"] = _np(x), _np(t), _np(labels), _np(y)
        for k, v in taps.items():
            out[f"s{S}/tap/{k}"] = _np(v)
    np.savez_compressed(os.path.join(OUT, "unet_small.npz"), **out)
    return m


def weight_checksums(sd):
    """Exact, order-independent pins for a regenerated state_dict: per tensor the int64 sum of the fp32 bit patterns
    (integer arithmetic: no dependence on reduction order or thread count), the element count, first and last bits."""
    names = sorted(sd.keys())
    rows = []
    for n in names:
        bits = sd[n].detach().float().contiguous().reshape(-1).view(torch.int32).to(torch.int64)
        rows.append([int(bits.sum().item()), bits.numel(), int(bits[0].item()), int(bits[-1].item())])
    return names, np.array(rows, dtype=np.int64)


def gen_unet_default64(RM):
    torch.manual_seed(DEFAULT_SEED)
    m = RM.UNet(**DEFAULT).eval()
    names, sums = weight_checksums(m.state_dict())
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(1, 3, 64, 64, generator=g)
    t = torch.tensor([417])
    # the sinusoidal table is computed with vectorised sin/cos/exp whose last bit differs between CPU generations:
    # it is pinned as data (the row the forward uses), not through the seed recipe
    out_row = _np(m.time_embedding.timembedding[0].weight[417])
    out = {"temb_row_417": out_row, "seed": np.array([DEFAULT_SEED]), "cfg_json": np.frombuffer(json.dumps(DEFAULT).encode(), dtype=np.uint8),
           "weight_names": np.array(names), "weight_checksums": sums, "x": _np(x), "t": _np(t)}
    with torch.no_grad():
        for lab in (1, 0):
            out[f"eps_label{lab}"] = _np(m(x, t, torch.tensor([lab])))
    np.savez_compressed(os.path.join(OUT, "unet_default64.npz"), **out)
    with open(os.path.join(OUT, "state_dict_default.json"), "w") as fh:
        json.dump({"n_params": sum(p.numel() for p in m.parameters()),
                   "entries": [[k, list(v.shape)] for k, v in m.state_dict().items()]}, fh, indent=0)


def gen_unet_default128(RM):
    """G4b: the REAL reference at BASELINE config C2's shape (128x128, L = 16 384 at the first level), B = 1.
    Same seed recipe as G4 (weights are pinned by G4's checksums); peak host memory ~25 GB."""
    torch.manual_seed(DEFAULT_SEED)
    m = RM.UNet(**DEFAULT).eval()
    g = torch.Generator().manual_seed(4242)
    x = torch.randn(1, 3, 128, 128, generator=g)
    t = torch.tensor([133])
    out = {"temb_row_133": _np(m.time_embedding.timembedding[0].weight[133]), "seed": np.array([DEFAULT_SEED]),
           "cfg_json": np.frombuffer(json.dumps(DEFAULT).encode(), dtype=np.uint8), "x": _np(x), "t": _np(t)}
    with torch.no_grad():
        for lab in (2, 0):
            out[f"eps_label{lab}"] = _np(m(x, t, torch.tensor([lab])))
    np.savez_compressed(os.path.join(OUT, "unet_default128.npz"), **out)


def gen_sampler_small(RM, RD):
    m = _small_model(RM)
    # random default-init weights saturate the trajectory; shrink the tail conv so x_t stays O(1) (SURVEY 8c note)
    out = {}
    g = torch.Generator().manual_seed(4321)
    x_T = torch.randn(2, 3, 16, 16, generator=g)
    labels = torch.tensor([1, 3])
    out["x_T"], out["labels"] = _np(x_T), _np(labels)
    out["beta"] = np.array([1e-4, 0.028])
    for w in (0.0, 1.8):
        samp = RD.GaussianDiffusionSampler(m, 1e-4, 0.028, SMALL["T"], w=w)
        torch.manual_seed(555)
        with _Recorder() as rec, torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
            y = samp(x_T, labels)
        tag = f"w{w}"
        # randn_like is drawn at time_step = T-1 ... 1 (not at 0): store indexed by time_step
        T = SMALL["T"]
        noise = np.zeros((T,) + tuple(x_T.shape), dtype=np.float32)
        for i, r in enumerate(rec.randn):
            noise[T - 1 - i] = _np(r)
        out[f"{tag}/noise_by_step"] = noise
        out[f"{tag}/traj_preclip"] = np.stack([_np(v) for v in rec.isnan_in])   # order: time_step T-1 ... 0
        out[f"{tag}/x_0"] = _np(y)
    np.savez_compressed(os.path.join(OUT, "sampler_small.npz"), **out)


def gen_sampler_default128(RM, RD):
    """G5b: three ancestral steps of the reference's own sampler at 128x128 on the seed-recipe default UNet, one sample at a
    time (B = 2 at once would hold 2 x 17 GB of scores per attention block); the GPU test runs both as one batch."""
    torch.manual_seed(DEFAULT_SEED)
    m = RM.UNet(**dict(DEFAULT, T=3)).eval()
    T = 3
    out = {"beta": np.array([1e-4, 0.028]), "w": np.array([1.8]), "T": np.array([T]), "seed": np.array([DEFAULT_SEED]),
           "temb_table_T3": _np(m.time_embedding.timembedding[0].weight)}
    g = torch.Generator().manual_seed(31337)
    samp = RD.GaussianDiffusionSampler(m, 1e-4, 0.028, T, w=1.8)
    xs, labs, noises, trajs, outs = [], [], [], [], []
    for i, lab in enumerate((1, 2)):
        x_T = torch.randn(1, 3, 128, 128, generator=g)
        labels = torch.tensor([lab])
        torch.manual_seed(600 + i)
        with _Recorder() as rec, torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
            y = samp(x_T, labels)
        noise = np.zeros((T, 1, 3, 128, 128), dtype=np.float32)
        for k, r in enumerate(rec.randn):
            noise[T - 1 - k] = _np(r)
        xs.append(_np(x_T)); labs.append(lab); noises.append(noise)
        trajs.append(np.stack([_np(v) for v in rec.isnan_in])); outs.append(_np(y))
    out["x_T"] = np.concatenate(xs, 0)
    out["labels"] = np.array(labs, dtype=np.int64)
    out["noise_by_step"] = np.concatenate(noises, 1)            # [T][2][3][128][128], indexed by time_step
    out["traj_preclip"] = np.concatenate(trajs, 1)              # order: time_step T-1 ... 0
    out["x_0"] = np.concatenate(outs, 0)
    np.savez_compressed(os.path.join(OUT, "sampler_default128.npz"), **out)


def gen_trainer_small(RM, RD):
    m = _small_model(RM)
    m.train()                      # dropout p=0 in SMALL -> no RNG inside the model
    trainer = RD.GaussianDiffusionTrainer(m, 1e-4, 0.028, SMALL["T"])
    g = torch.Generator().manual_seed(777)
    x_0 = torch.rand(4, 3, 16, 16, generator=g) * 2 - 1
    labels = torch.tensor([1, 2, 0, 3])
    torch.manual_seed(888)
    with _Recorder() as rec:
        loss = trainer(x_0, labels)
    out = {"x_0": _np(x_0), "labels": _np(labels), "t": _np(rec.randint[0]), "noise": _np(rec.randn[0]),
           "loss": _np(loss), "beta": np.array([1e-4, 0.028])}
    b = x_0.shape[0]
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4, weight_decay=1e-4)
    opt.zero_grad()
    (loss.sum() / b ** 2.).backward()            # TrainCondition.py:59-60
    grad_names = ["head.weight", "tail.2.weight", "downblocks.0.attn.in_proj_weight", "downblocks.0.block1.0.weight",
                  "downblocks.1.c2.weight", "upblocks.2.t.weight", "time_embedding.timembedding.0.weight",
                  "cond_embedding.condEmbedding.0.weight", "middleblocks.0.temb_proj.1.weight",
                  "upblocks.0.shortcut.weight", "downblocks.2.attn.out_proj.bias"]
    params = dict(m.named_parameters())
    for n in grad_names:
        out[f"grad/{n}"] = _np(params[n].grad)
    out["grad_total_norm"] = np.array([torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0).item()])  # :61-62
    opt.step()                                                                                      # :63
    for n in grad_names:
        out[f"after_step/{n}"] = _np(params[n])
    np.savez_compressed(os.path.join(OUT, "trainer_small.npz"), **out)


def gen_unet_wide(RM, RD):
    """G3c: four resolution levels with widths 32 / 64 / 96 / 128 (8 heads of 4 / 8 / 12 / 16 channels) from the REAL
    reference: eval forward at 32x32, and one trainer pass (loss, a few gradients, total norm).  Weights come from the seed
    recipe (pinned by checksums), plus a seeded perturbation of the zero-initialised MHA biases."""
    def build():
        torch.manual_seed(WIDE_SEED)
        m = RM.UNet(**WIDE)
        g = torch.Generator().manual_seed(WIDE_SEED + 1)
        with torch.no_grad():
            for n, p in sorted(m.named_parameters()):
                if n.endswith("in_proj_bias") or n.endswith("out_proj.bias"):
                    p.copy_(torch.randn(p.shape, generator=g) * 0.05)
        return m
    m = build().eval()
    names, sums = weight_checksums(m.state_dict())
    g = torch.Generator().manual_seed(515)
    x = torch.randn(2, 3, 32, 32, generator=g)
    t, labels = torch.tensor([5, 11]), torch.tensor([3, 0])
    out = {"seed": np.array([WIDE_SEED]), "cfg_json": np.frombuffer(json.dumps(WIDE).encode(), dtype=np.uint8),
           "weight_names": np.array(names), "weight_checksums": sums, "x": _np(x), "t": _np(t), "labels": _np(labels),
           "temb_rows": _np(m.time_embedding.timembedding[0].weight)}
    with torch.no_grad():
        out["eps"] = _np(m(x, t, labels))
    m.train()
    trainer = RD.GaussianDiffusionTrainer(m, 1e-4, 0.02, WIDE["T"])
    x_0 = torch.rand(2, 3, 32, 32, generator=g) * 2 - 1
    tl = torch.tensor([1, 2])
    torch.manual_seed(616)
    with _Recorder() as rec:
        loss = trainer(x_0, tl)
    (loss.sum() / 2 ** 2.).backward()
    out.update({"x_0": _np(x_0), "train_labels": _np(tl), "train_t": _np(rec.randint[0]), "noise": _np(rec.randn[0]),
                "loss": _np(loss)})
    params = dict(m.named_parameters())
    for n in ["head.weight", "downblocks.0.attn.in_proj_weight", "downblocks.2.attn.in_proj_weight",
              "downblocks.4.attn.in_proj_weight", "downblocks.4.attn.out_proj.weight", "downblocks.6.attn.in_proj_weight",
              "middleblocks.0.attn.in_proj_bias", "downblocks.4.block1.2.weight", "upblocks.3.shortcut.weight",
              "tail.2.weight"]:
        out[f"grad/{n}"] = _np(params[n].grad)
    out["grad_total_norm"] = np.array([torch.nn.utils.clip_grad_norm_(m.parameters(), 1e9).item()])
    np.savez_compressed(os.path.join(OUT, "unet_wide.npz"), **out)


def gen_trainer_default64(RM, RD):
    """G6b: one trainer pass of the DEFAULT model (ch=128, [1,2,2,2], d_head 16 / 32; attention over L = 4096 at the first
    level) from the REAL reference at 64x64, B = 2: loss, gradients of small tensors from every part of the network, the
    total gradient norm.  dropout = 0 (torch's CPU dropout stream cannot be reproduced by a device generator); weights from
    the seed recipe + the seeded perturbation of the zero-initialised MHA biases used by G3c."""
    cfg = dict(DEFAULT, dropout=0.0)
    torch.manual_seed(DEFAULT_SEED)
    m = RM.UNet(**cfg)
    g = torch.Generator().manual_seed(DEFAULT_SEED + 1)
    with torch.no_grad():
        for n, p in sorted(m.named_parameters()):
            if n.endswith("in_proj_bias") or n.endswith("out_proj.bias"):
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)
    m.train()
    trainer = RD.GaussianDiffusionTrainer(m, 1e-4, 0.02, cfg["T"])
    gx = torch.Generator().manual_seed(2024)
    x_0 = torch.rand(2, 3, 64, 64, generator=gx) * 2 - 1
    labels = torch.tensor([1, 2])
    torch.manual_seed(99)
    with _Recorder() as rec:
        loss = trainer(x_0, labels)
    (loss.sum() / 2 ** 2.).backward()
    out = {"seed": np.array([DEFAULT_SEED]), "cfg_json": np.frombuffer(json.dumps(cfg).encode(), dtype=np.uint8),
           "x_0": _np(x_0), "labels": _np(labels), "t": _np(rec.randint[0]), "noise": _np(rec.randn[0]), "loss": _np(loss),
           "temb_rows": _np(m.time_embedding.timembedding[0].weight[rec.randint[0]])}
    params = dict(m.named_parameters())
    names = ["head.bias", "tail.2.weight", "tail.0.weight", "time_embedding.timembedding.3.bias",
             "cond_embedding.condEmbedding.3.bias", "downblocks.0.block1.0.weight", "downblocks.0.attn.in_proj_bias",
             "downblocks.0.attn.out_proj.bias", "downblocks.1.block2.3.bias", "downblocks.2.c1.bias",
             "downblocks.3.shortcut.bias", "downblocks.3.attn.in_proj_bias", "downblocks.4.temb_proj.1.bias",
             "middleblocks.0.attn.in_proj_bias", "middleblocks.1.cond_proj.1.bias", "upblocks.0.block1.0.bias",
             "upblocks.3.t.bias", "upblocks.14.block2.0.weight", "upblocks.14.shortcut.bias"]
    for n in names:
        out[f"grad/{n}"] = _np(params[n].grad)
    # slices of large tensors: the first 4 output rows
    for n in ["downblocks.0.attn.in_proj_weight", "downblocks.3.attn.in_proj_weight", "downblocks.0.block1.2.weight",
              "upblocks.14.block1.2.weight"]:
        out[f"gradrows/{n}"] = _np(params[n].grad[:4])
    out["grad_total_norm"] = np.array([torch.nn.utils.clip_grad_norm_(m.parameters(), 1e9).item()])
    np.savez_compressed(os.path.join(OUT, "trainer_default64.npz"), **out)


def gen_lr_schedule():
    RS = RL.load_scheduler()
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=1e-4, weight_decay=1e-4)
    epochs, mult = 70, 2.5
    cos = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer=opt, T_max=epochs, eta_min=0, last_epoch=-1)
    warm = RS.GradualWarmupScheduler(optimizer=opt, multiplier=mult, warm_epoch=epochs // 10, after_scheduler=cos)
    lrs = []
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for _ in range(epochs):
            lrs.append(opt.param_groups[0]["lr"])
            opt.step()
            warm.step()
    with open(os.path.join(OUT, "lr_schedule.json"), "w") as fh:
        json.dump({"epochs": epochs, "multiplier": mult, "base_lr": 1e-4, "lr_by_epoch": lrs}, fh)


def main():
    if not RL.available():
        sys.exit("reference not present at " + RL.REFERENCE_ROOT)
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    RD, RM = RL.load_diffusion(), RL.load_model()
    if len(sys.argv) > 1 and sys.argv[1] == "g4b":      # the large fixture alone (minutes, ~25 GB of host memory)
        gen_unet_default128(RM)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "g2b":
        gen_attnblock(RM)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "g6b":
        gen_trainer_default64(RM, RD)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "g3c":
        gen_unet_wide(RM, RD)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "g5b":
        gen_sampler_default128(RM, RD)
        return
    gen_schedules(RD)
    gen_modules(RM)
    gen_unet_small(RM)
    gen_unet_default64(RM)
    gen_sampler_small(RM, RD)
    gen_trainer_small(RM, RD)
    gen_lr_schedule()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
