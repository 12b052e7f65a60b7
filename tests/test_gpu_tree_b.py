"""GPU parity of the second tree's inference path (DynamicUNet + ancestral / DDIM sampler) against the golden vectors
produced by the real reference and against the CPU oracle (oracle/cpu_path_b.py)."""
import ctypes as C
import json
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import hdiff_amd  # noqa: E402
from hdiff_amd import _capi  # noqa: E402
from hdiff_amd.diffusion.Diffusion import GaussianDiffusionSampler  # noqa: E402
from hdiff_amd.diffusion.Model import DynamicUNet  # noqa: E402
from oracle import cpu_path_b as OB  # noqa: E402

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(ROOT, "tests", "golden")
DEV = "cuda:0"


def T(a):
    return torch.from_numpy(np.asarray(a))


def small_model():
    from _tree_b_small import load_small_dyn_unet
    d, cfg, m, sd = load_small_dyn_unet()
    ocfg = OB.DynUNetConfig(T=cfg["T"], ch=cfg["ch"], ch_mult=tuple(cfg["ch_mult"]), num_res_blocks=cfg["num_res_blocks"])
    return d, m.to(DEV), ocfg, sd


def stream():
    return torch.cuda.current_stream().cuda_stream


def test_small_ops_match_torch():
    lib = _capi.lib()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(6, 5, 7, generator=g)
    dx = x.to(DEV)
    for OH, OW in [(10, 14), (9, 9), (5, 7), (3, 4), (16, 8)]:
        y = torch.empty(6, OH, OW, device=DEV)
        _capi.check(lib.hdiff_resize_nearest(dx.data_ptr(), y.data_ptr(), 6, 5, 7, OH, OW, stream()))
        assert torch.equal(y.cpu(), F.interpolate(x[None], size=(OH, OW), mode="nearest")[0])
    x = torch.randn(12, 333, generator=g)
    dx, y = x.to(DEV), torch.empty(12, device=DEV)
    _capi.check(lib.hdiff_avgpool_global(dx.data_ptr(), y.data_ptr(), 12, 333, stream()))
    assert (y.cpu() - x.mean(dim=1)).abs().max() < 1e-6
    # DDIM update: bit-exact with the reference's separate fp32 tensor ops
    sched = OB.sampler_schedule(1e-4, 0.02, 1000)
    tab = OB.ddim_coefficients(sched, 5)
    yv, ev = torch.randn(1000, generator=g), torch.randn(1000, generator=g)
    d_y, d_e, d_tab = yv.to(DEV), ev.to(DEV), tab.to(DEV)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    for k in range(5):
        step = torch.tensor([k], dtype=torch.int32, device=DEV)
        out = torch.empty(1000, device=DEV)
        _capi.check(lib.hdiff_ddim_step(d_y.data_ptr(), d_e.data_ptr(), out.data_ptr(), d_tab.data_ptr(), step.data_ptr(),
                                        5, flag.data_ptr(), 1000, stream()))
        y0 = (yv - ev * tab[k, 0]) / tab[k, 1]
        want = tab[k, 2] * y0 + tab[k, 3] * ev
        assert torch.equal(out.cpu(), want), k
    assert int(flag.item()) == 0
    tt = torch.empty(3, dtype=torch.int64, device=DEV)
    table = torch.tensor([0, 200, 400], dtype=torch.int32, device=DEV)
    _capi.check(lib.hdiff_fill_from_table(tt.data_ptr(), table.data_ptr(), torch.tensor([2], dtype=torch.int32, device=DEV).data_ptr(),
                                          3, 3, stream()))
    assert tt.tolist() == [400, 400, 400]
    _capi.check(lib.hdiff_fill_from_table(tt.data_ptr(), table.data_ptr(), torch.tensor([9], dtype=torch.int32, device=DEV).data_ptr(),
                                          3, 3, stream()))
    assert tt.tolist() == [400, 400, 400]                        # an index past the table is clamped, not read


def test_dyn_unet_small_matches_reference_golden():
    d, m, _, _ = small_model()
    for tag in ("s16", "s32"):
        x, t, lab = T(d[f"{tag}/x"]).to(DEV), T(d[f"{tag}/t"]).to(DEV), T(d[f"{tag}/label_image"]).to(DEV)
        with torch.no_grad():
            e0 = m(x, t)
            e1 = m(x, t, lab, context_zero=False)
        for got, key in ((e0, "eps_context_zero"), (e1, "eps_image_label")):
            ref = T(d[f"{tag}/{key}"])
            err = (got.cpu() - ref).abs().max().item()
            assert err <= 1e-4 * max(1.0, ref.abs().max().item()), (tag, key, err)
    # dynamic_forward only toggles requires_grad of the middle blocks (underwater: even blocks train)
    x = torch.zeros(1, 6, 16, 16, device=DEV)
    x[:, 2] = 1.0
    with torch.no_grad():
        m(x, torch.zeros(1, dtype=torch.long, device=DEV))
    flags = [all(p.requires_grad for p in blk.parameters()) for blk in m.middleblocks]
    assert flags == [True, False, True, False]


def test_dyn_unet_odd_sizes_match_oracle():
    """24x40: the skip tensors popped by the short up path have the wrong resolution and are resized (nearest)."""
    d, m, ocfg, sd = small_model()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 6, 24, 40, generator=g)
    t = torch.tensor([3, 977])
    with torch.no_grad():
        want = OB.dyn_unet_forward(sd, ocfg, x, t)
        got = m(x.to(DEV), t.to(DEV))
    assert got.shape == want.shape
    assert (got.cpu() - want).abs().max().item() <= 1e-4 * max(1.0, want.abs().max().item())


def test_dyn_unet_default64_matches_reference_golden():
    d = np.load(os.path.join(GOLDEN, "dyn_unet_default64.npz"))
    cfg = json.loads(bytes(d["cfg_json"]).decode())
    torch.manual_seed(int(d["seed"][0]))
    m = DynamicUNet(**cfg).eval()
    with torch.no_grad():
        m.time_embedding.timembedding[0].weight[417] = T(d["temb_row_417"])
    m = m.to(DEV)
    x, t, lab = T(d["x"]).to(DEV), T(d["t"]).to(DEV), T(d["label_image"]).to(DEV)
    with torch.no_grad():
        for key, kw in (("context_zero", {}), ("image_label", dict(labels=lab, context_zero=False))):
            eps = m(x, t, **kw)
            up = m.plan_for(1, 64, 64, x.device, "labels" not in kw)
            h, (scale, shift) = up.tail_in_src
            a = h * scale[:, :, None, None] + shift[:, :, None, None]
            tail_in = (a * torch.sigmoid(a))[:, ::8].cpu()
            ref_in = T(d[f"tail_in_{key}_ch8"])
            assert (tail_in - ref_in).abs().max().item() <= 3e-4 * max(1.0, ref_in.abs().max().item())
            ref = T(d[f"eps_{key}"])
            assert (eps.cpu() - ref).abs().max().item() <= 3e-4 * ref.abs().max().item()


def test_dyn_sampler_matches_reference_golden():
    d, m, _, _ = small_model()
    s = np.load(os.path.join(GOLDEN, "dyn_sampler_small.npz"))
    img = T(s["input_image"]).to(DEV)
    with torch.no_grad():
        b = s["beta_ancestral"]
        Tn = int(s["ancestral/T"][0])
        samp = GaussianDiffusionSampler(m, float(b[0]), float(b[1]), Tn).to(DEV)
        noise = [T(n).to(DEV) for n in s["ancestral/randn_after"]]
        y = samp(img, y_T=T(s["ancestral/y_T"]).to(DEV), noise_by_step=noise)
        err = (y.cpu() - T(s["ancestral/y_0"])).abs().max().item()
        assert err <= 2e-4, err
        b = s["beta_ddim"]
        samp = GaussianDiffusionSampler(m, float(b[0]), float(b[1]), 1000).to(DEV)
        for tag, scale in (("ddim_s1", 1), ("ddim_s1.8", 1.8)):
            traj = []
            y_eager = samp(img, ddim=True, unconditional_guidance_scale=scale, ddim_step=5, y_T=T(s[f"{tag}/y_T"]).to(DEV),
                           trajectory=traj)
            err = (y_eager.cpu() - T(s[f"{tag}/y_0"])).abs().max().item()
            assert err <= 2e-4, (tag, err)
            assert len(traj) == 5
            y_graph = samp(img, ddim=True, unconditional_guidance_scale=scale, ddim_step=5, y_T=T(s[f"{tag}/y_T"]).to(DEV))
            assert torch.equal(y_graph, y_eager), "hipGraph replay and eager launches must agree bit for bit"
        with pytest.raises(RuntimeError, match="out of bounds"):
            samp(img, ddim=True, ddim_step=1000)          # alphas_bar[t + 1] with t = 999 (Diffusion.py:250)


def test_dyn_sampler_seeded_noise_is_reproducible():
    d, m, _, _ = small_model()
    img = torch.randint(0, 256, (1, 3, 16, 16), generator=torch.Generator().manual_seed(1)).float().to(DEV)
    samp = GaussianDiffusionSampler(m, 1e-4, 0.028, 6).to(DEV)
    with torch.no_grad():
        torch.manual_seed(7)
        a = samp(img)
        torch.manual_seed(7)
        b = samp(img)
        torch.manual_seed(8)
        c = samp(img)
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert torch.isfinite(a).all() and a.abs().max() <= 1.0
