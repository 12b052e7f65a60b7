"""SURVEY section 8 rows f2 / f4 on the GPU box: the quality measures are CPU numpy by design (the reference
evaluates them on the host: utils/rotinas.py:922, 926, metrics/metrics.py:282-299), but until this file they had no record under `-m gpu`.
Here they run on what the HIP sampler produced, the golden comparisons of tests/test_uw_metrics.py run on the box too, and SSIM is pinned
against values worked out BY HAND from the published definition with scikit-image's documented defaults (7x7 uniform window, K1 = 0.01,
K2 = 0.03, sample covariance, data_range 255).  scikit-image itself (reference pin 0.22.0, CLEDiff_bkp.yaml:293) is not in this image and
the reference records no metric values: rows f2 and the colour-space parts of f4 stay "parity unpinned" (DESIGN.md section 2)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import hdiff_amd  # noqa: E402
from hdiff_amd import metrics as M  # noqa: E402
from hdiff_amd import uw_metrics as U  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_ssim_psnr_hand_computed_single_window():
    """One 7x7 window (the image IS the window: after the (win - 1) / 2 border crop exactly one SSIM value is left), integers only.
    x = the ramp 0 .. 48: mean 24, sample variance sum((i - 24)^2) / 48 = 2 * 4900 / 48 = 9800 / 48.
      y = x + 10:   same variance, covariance = variance  -> SSIM = (2 * 24 * 34 + C1) / (24^2 + 34^2 + C1)          (structure term 1)
      y = 48 - x:   mean 24, covariance = -variance        -> SSIM = (2 v_neg + C2) / (2 v + C2) with v_neg = -v       (luminance term 1)
      y = 100:      variance 0, covariance 0               -> SSIM = (2 * 24 * 100 + C1) / (24^2 + 100^2 + C1) * C2 / (v + C2)
    C1 = (0.01 * 255)^2 = 6.5025, C2 = (0.03 * 255)^2 = 58.5225.  PSNR of y = x + 10: 10 log10(255^2 / 100)."""
    x = np.arange(49, dtype=np.float64).reshape(7, 7)
    v = 9800.0 / 48.0
    c1, c2 = 6.5025, 58.5225
    assert abs(M.ssim(x, x + 10.0, 255, channel_axis=None) - (2 * 24 * 34 + c1) / (24 ** 2 + 34 ** 2 + c1)) < 1e-12
    assert abs(M.ssim(x, 48.0 - x, 255, channel_axis=None) - (-2 * v + c2) / (2 * v + c2)) < 1e-12
    assert abs(M.ssim(x, np.full((7, 7), 100.0), 255, channel_axis=None)
               - (2 * 24 * 100 + c1) / (24 ** 2 + 100 ** 2 + c1) * c2 / (v + c2)) < 1e-12
    assert abs(M.psnr(x, x + 10.0, 255) - 10.0 * np.log10(255.0 ** 2 / 100.0)) < 1e-12
    # three channels = the mean over the channels (channel_axis = 2, as the reference calls it)
    rgb_a = np.stack([x, x, x], axis=2)
    rgb_b = np.stack([x + 10.0, 48.0 - x, np.full((7, 7), 100.0)], axis=2)
    want = ((2 * 24 * 34 + c1) / (24 ** 2 + 34 ** 2 + c1) + (-2 * v + c2) / (2 * v + c2)
            + (2 * 24 * 100 + c1) / (24 ** 2 + 100 ** 2 + c1) * c2 / (v + c2)) / 3.0
    assert abs(M.ssim(rgb_a, rgb_b, 255, channel_axis=2) - want) < 1e-12


def test_quality_measures_on_hip_sampler_output_against_the_oracle():
    """A short CFG sampling run of a small model on the GPU (the product path) and on the CPU oracle with the same weights and noise:
    PSNR / SSIM between the two on the reference's evaluation scale (uint8 HWC, data_range 255) and the underwater measures of both
    images -- the numbers a user of the reference's test() would print (utils/rotinas.py:922-930) agree between the two paths."""
    from hdiff_amd.DiffusionFreeGuidence.ModelCondition import UNet
    from hdiff_amd.DiffusionFreeGuidence.DiffusionCondition import GaussianDiffusionSampler
    from oracle import cpu_path as O
    T_, S = 6, 32
    cfgd = dict(T=T_, num_labels=2, ch=32, ch_mult=[1, 2], num_res_blocks=1, dropout=0.0)
    torch.manual_seed(3)
    model = UNet(**cfgd).eval()
    with torch.no_grad():
        model.tail[2].weight.mul_(0.1)           # keep the states inside (-1, 1): a saturated image would make the comparison vacuous
    g = torch.Generator().manual_seed(5)
    x_T = torch.randn(2, 3, S, S, generator=g)
    labels = torch.tensor([1, 2])
    noise = torch.randn(T_, 2, 3, S, S, generator=g)
    cfg = O.UNetConfig(T=T_, num_labels=2, ch=32, ch_mult=(1, 2), num_res_blocks=1)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        ref = O.sampler_forward(sd, cfg, 1e-4, 0.028, T_, 1.8, x_T, labels, list(noise))
        out = GaussianDiffusionSampler(model.to(DEV), 1e-4, 0.028, T_, w=1.8).to(DEV)(x_T.to(DEV), labels.to(DEV), noise_by_step=noise.to(DEV)).cpu()
    assert (ref.abs() < 1.0).float().mean() > 0.25
    p, s = M.batch_psnr_ssim(ref * 0.5 + 0.5, out * 0.5 + 0.5)
    print(f"GPU vs oracle sampler output: PSNR {p:.1f} dB, SSIM {s:.8f}")
    assert p >= 60.0 and s >= 0.99999, (p, s)           # uint8 quantisation: identical images give inf / 1; one grey level off in a few pixels ~ 70 dB
    for b in range(2):
        ia = np.asarray((ref[b].clamp(-1, 1) * 0.5 + 0.5).mul(255).permute(1, 2, 0), dtype=np.float32)
        ib = np.asarray((out[b].clamp(-1, 1) * 0.5 + 0.5).mul(255).permute(1, 2, 0), dtype=np.float32)
        qa, qb = U.getUIQM(ia), U.getUIQM(ib)
        ea, eb = U.eme(ia.mean(axis=2)), U.eme(ib.mean(axis=2))
        assert np.isfinite(qa) and abs(qa - qb) <= 1e-3 * max(1.0, abs(qa)), (qa, qb)
        assert np.isfinite(ea) and abs(ea - eb) <= 1e-3 * max(1.0, abs(ea)), (ea, eb)


def test_underwater_measures_golden_on_the_gpu_box():
    """tests/test_uw_metrics.py (values computed by the reference's own functions, tests/golden/uw_metrics.npz) under the gpu marker: the box's
    numpy / scipy give the container's numbers."""
    import test_uw_metrics as T
    T.test_uiqm_family_matches_reference_functions()
    T.test_block_measures_against_brute_force()
    T.test_unpinned_measures_behave()
