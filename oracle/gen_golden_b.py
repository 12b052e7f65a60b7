"""TEST INFRASTRUCTURE ONLY -- golden vectors for the reference's second tree (image-conditioned DynamicUNet + DDIM sampler).

Run in the build container (needs ``/root/reference``):

    python -m oracle.gen_golden_b          # writes tests/golden/dyn_*.npz + state_dict_dyn_default.json

Loads the REAL reference classes in place (``reference_loader.load_model_b`` / ``load_sampler_b``), runs them on seeded CPU
inputs and stores inputs + expected outputs only.

  GB1 dyn_unet_small.npz      small DynamicUNet (ch=32, ch_mult=[1,2,2], nrb=1, T=1000; tail conv scaled up so eps is O(1)):
                              seed recipe + weight checksums + time table, x[.,6,.,.], t, eps with context_zero=True and with an image label, @16^2 and @32^2
  GB2 dyn_unet_default64.npz  default DynamicUNet (ch=128,[1,2,2,2],nrb=2) @64^2 B=1: seed recipe, weight checksums, inputs,
                              the activation entering the tail conv (O(1) pin; eps itself is ~1e-5 by the xavier gain), eps
  GB3 dyn_sampler_small.npz   sampler on the small model: ancestral T=6, DDIM (5 steps of 1000) with guidance scale 1 and 1.8:
                              input image, every tensor the reference drew from torch's RNG, outputs
  GB4 state_dict_dyn_default.json   the 319 (name, shape) pairs of the default DynamicUNet
"""
from __future__ import annotations

import json
import os

import numpy as np
import torch

from . import reference_loader as RL
from .gen_golden import OUT, _Recorder, _np, _sd_np, weight_checksums

SMALL_B = dict(T=1000, ch=32, ch_mult=[1, 2, 2], num_res_blocks=1, dropout=0.0)
DEFAULT_B = dict(T=1000, ch=128, ch_mult=[1, 2, 2, 2], num_res_blocks=2, dropout=0.0)
SMALL_B_SEED = 4242
DEFAULT_B_SEED = 0
TAIL_GAIN = 2.0e4        # DynamicUNet.initialize() gives the tail conv xavier gain 1e-5 (Model.py:406)


def small_model(RMB):
    torch.manual_seed(SMALL_B_SEED)
    m = RMB.DynamicUNet(**SMALL_B).eval()
    with torch.no_grad():
        m.tail[2].weight.mul_(TAIL_GAIN)
        m.tail[2].bias.add_(0.05)
    return m


def gen_dyn_unet_small(RMB):
    """The weights are NOT stored (5 MB): the build's DynamicUNet reproduces the reference's seeded init bit for bit (checked
    through the integer checksums below), so the fixture carries the seed, the two tail edits of small_model() and the
    sinusoidal table (whose last bit depends on the CPU generation) as data."""
    m = small_model(RMB)
    names, sums = weight_checksums(m.state_dict())
    out = {"seed": np.array([SMALL_B_SEED]), "tail_gain": np.array([TAIL_GAIN]), "tail_bias_add": np.array([0.05]),
           "temb_table": _np(m.time_embedding.timembedding[0].weight), "weight_names": np.array(names), "weight_checksums": sums}
    out["cfg_json"] = np.frombuffer(json.dumps(SMALL_B).encode(), dtype=np.uint8)
    g = torch.Generator().manual_seed(77)
    for tag, (B, S) in {"s16": (2, 16), "s32": (1, 32)}.items():
        x = torch.randn(B, 6, S, S, generator=g)
        lab = torch.rand(B, 3, S, S, generator=g)
        t = torch.randint(0, 1000, (B,), generator=g)
        with torch.no_grad():
            e0 = m(x, t)                                     # labels=None, context_zero=True: what the sampler calls
            e1 = m(x, t, lab, context_zero=False)
        out[f"{tag}/x"], out[f"{tag}/t"], out[f"{tag}/label_image"] = _np(x), _np(t), _np(lab)
        out[f"{tag}/eps_context_zero"], out[f"{tag}/eps_image_label"] = _np(e0), _np(e1)
    np.savez_compressed(os.path.join(OUT, "dyn_unet_small.npz"), **out)


def gen_dyn_unet_default64(RMB):
    torch.manual_seed(DEFAULT_B_SEED)
    m = RMB.DynamicUNet(**DEFAULT_B).eval()
    names, sums = weight_checksums(m.state_dict())
    g = torch.Generator().manual_seed(4321)
    x = torch.randn(1, 6, 64, 64, generator=g)
    lab = torch.rand(1, 3, 64, 64, generator=g)
    t = torch.tensor([417])
    grabbed = {}
    hook = m.tail[1].register_forward_hook(lambda mod, inp, res: grabbed.__setitem__("tail_in", res.detach().clone()))
    out = {"temb_row_417": _np(m.time_embedding.timembedding[0].weight[417]), "seed": np.array([DEFAULT_B_SEED]),
           "cfg_json": np.frombuffer(json.dumps(DEFAULT_B).encode(), dtype=np.uint8),
           "weight_names": np.array(names), "weight_checksums": sums, "x": _np(x), "t": _np(t), "label_image": _np(lab)}
    with torch.no_grad():
        out["eps_context_zero"] = _np(m(x, t))
        out["tail_in_context_zero_ch8"] = _np(grabbed["tail_in"][:, ::8])     # every 8th channel keeps the fixture small
        out["eps_image_label"] = _np(m(x, t, lab, context_zero=False))
        out["tail_in_image_label_ch8"] = _np(grabbed["tail_in"][:, ::8])
    hook.remove()
    np.savez_compressed(os.path.join(OUT, "dyn_unet_default64.npz"), **out)
    with open(os.path.join(OUT, "state_dict_dyn_default.json"), "w") as fh:
        json.dump({"n_params": sum(p.numel() for p in m.parameters()),
                   "entries": [[k, list(v.shape)] for k, v in m.state_dict().items()]}, fh, indent=0)


def gen_dyn_sampler_small(RMB, RDB):
    m = small_model(RMB)
    g = torch.Generator().manual_seed(99)
    img = torch.randint(0, 256, (2, 3, 16, 16), generator=g).float()       # the sampler divides by 255 itself (Diffusion.py:220)
    out = {"input_image": _np(img), "beta_ancestral": np.array([1e-4, 0.028]), "beta_ddim": np.array([1e-4, 0.02])}
    runs = {"ancestral": (dict(), 6, (1e-4, 0.028)),
            "ddim_s1": (dict(ddim=True, unconditional_guidance_scale=1, ddim_step=5), 1000, (1e-4, 0.02)),
            "ddim_s1.8": (dict(ddim=True, unconditional_guidance_scale=1.8, ddim_step=5), 1000, (1e-4, 0.02))}
    for tag, (kw, T, beta) in runs.items():
        samp = RDB.GaussianDiffusionSampler(m, beta[0], beta[1], T)
        torch.manual_seed(31337)
        with _Recorder() as rec, torch.no_grad():
            y = samp(img, **kw)
        out[f"{tag}/T"] = np.array([T])
        out[f"{tag}/y_T"] = _np(rec.randn[0])                               # first draw: the start noise (:226 / :239)
        out[f"{tag}/randn_after"] = np.stack([_np(r) for r in rec.randn[1:]])  # per-step draws, in call order
        out[f"{tag}/y_0"] = _np(y)
    np.savez_compressed(os.path.join(OUT, "dyn_sampler_small.npz"), **out)


def main():
    torch.set_num_threads(8)
    RMB, RDB = RL.load_model_b(), RL.load_sampler_b()
    gen_dyn_unet_small(RMB)
    gen_dyn_unet_default64(RMB)
    gen_dyn_sampler_small(RMB, RDB)
    for f in sorted(os.listdir(OUT)):
        if f.startswith("dyn_") or "dyn" in f:
            print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
