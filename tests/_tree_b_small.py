"""Shared by the tree-B tests: rebuild the small DynamicUNet of tests/golden/dyn_unet_small.npz from its seed recipe.

The fixture does not store the 1.3 M weights.  The build's DynamicUNet reproduces the reference's seeded initialisation bit
for bit; the recipe (seed, the two edits of the tail conv, the sinusoidal table as data) is applied here and verified against
the integer checksums the real reference produced (oracle/gen_golden_b.py)."""
import json
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_small_dyn_unet():
    """-> (fixture npz, constructor kwargs, DynamicUNet on the CPU in eval mode, its state_dict as plain tensors)."""
    import hdiff_amd  # noqa: F401
    from hdiff_amd.diffusion.Model import DynamicUNet
    d = np.load(os.path.join(GOLDEN, "dyn_unet_small.npz"))
    cfg = json.loads(bytes(d["cfg_json"]).decode())
    torch.manual_seed(int(d["seed"][0]))
    m = DynamicUNet(**cfg).eval()
    with torch.no_grad():
        m.tail[2].weight.mul_(float(d["tail_gain"][0]))
        m.tail[2].bias.add_(float(d["tail_bias_add"][0]))
        m.time_embedding.timembedding[0].weight.copy_(torch.from_numpy(d["temb_table"]))
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    names = sorted(sd.keys())
    assert names == list(d["weight_names"])
    for n, want in zip(names, d["weight_checksums"]):
        bits = sd[n].float().contiguous().reshape(-1).view(torch.int32).to(torch.int64)
        got = [int(bits.sum()), bits.numel(), int(bits[0]), int(bits[-1])]
        assert got == list(want), f"seed recipe no longer reproduces the reference's weights for {n}"
    return d, cfg, m, sd
