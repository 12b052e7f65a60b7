"""End-to-end parity at BASELINE.json's first configuration (64x64, T=50, batch 1, w=1.8, default U-Net): the HIP sampler
against the CPU oracle with shared weights, start noise and per-step noise, judged the way the reference's evaluation does
(PSNR / SSIM on x*0.5+0.5) -- SURVEY.md section 8d: target >= 40 dB.  The CPU side takes about a minute."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hdiff_amd  # noqa: E402
from hdiff_amd import metrics as M  # noqa: E402
from hdiff_amd.DiffusionFreeGuidence import DiffusionCondition as DC, ModelCondition as MC  # noqa: E402
from hdiff_amd.diffusion.Diffusion import GaussianDiffusionSampler as SamplerB  # noqa: E402
from hdiff_amd.diffusion.Model import DynamicUNet  # noqa: E402
from oracle import cpu_path as O, cpu_path_b as OB  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def quality(a, b):
    """PSNR (dB) and SSIM of two [-1, 1] image batches on the reference's evaluation scale."""
    ia = ((a.cpu().clamp(-1, 1) * 0.5 + 0.5) * 255.0).permute(0, 2, 3, 1).numpy()
    ib = ((b.cpu().clamp(-1, 1) * 0.5 + 0.5) * 255.0).permute(0, 2, 3, 1).numpy()
    ps = [M.psnr(x, y, 255) for x, y in zip(ia, ib)]
    ss = [M.ssim(x, y, 255, channel_axis=2) for x, y in zip(ia, ib)]
    return min(ps), min(ss)


def test_c1_sampling_64x64_T50_matches_cpu_path():
    T_, S, w, beta = 50, 64, 1.8, (1e-4, 0.028)
    cfg = dict(T=T_, num_labels=10, ch=128, ch_mult=[1, 2, 2, 2], num_res_blocks=2, dropout=0.15)
    torch.manual_seed(0)
    m = MC.UNet(**cfg).eval()
    with torch.no_grad():
        m.tail[2].weight.mul_(0.1)           # default init saturates x to +-1 within a few steps; keep x O(1)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    ocfg = O.UNetConfig(T=T_, num_labels=10, ch=128, ch_mult=(1, 2, 2, 2), num_res_blocks=2)
    g = torch.Generator().manual_seed(1234)
    x_T = torch.randn(1, 3, S, S, generator=g)
    labels = torch.tensor([1])
    noise = torch.randn(T_, 1, 3, S, S, generator=g)
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    ref_traj = []
    with torch.no_grad():
        ref = O.sampler_forward(sd, ocfg, beta[0], beta[1], T_, w, x_T, labels, noise, trajectory=ref_traj)
    assert (ref.abs() < 1.0).float().mean() > 0.25, "reference run saturated: the comparison would be vacuous"
    m = m.to(DEV)
    samp = DC.GaussianDiffusionSampler(m, beta[0], beta[1], T_, w=w).to(DEV)
    lib = hdiff_amd.lib()
    before = hdiff_amd.get_contraction_mode()
    try:
        for mode in ("f32", "bf16x3"):
            hdiff_amd.set_contraction_mode(mode)
            traj = []
            with torch.no_grad():
                out = samp(x_T.to(DEV), labels.to(DEV), noise_by_step=noise.to(DEV), trajectory=traj)
            worst = max((a.cpu() - b).abs().max().item() for a, b in zip(traj, ref_traj))
            psnr, ssim = quality(out, ref)
            print(f"C1 {mode}: worst pre-clip trajectory error {worst:.2e}, PSNR {psnr:.1f} dB, SSIM {ssim:.6f}")
            # measured: 1.9e-6 / 144 dB in both modes (SURVEY 8(d) asks for >= 40 dB)
            assert worst < 5e-5 and psnr >= 100.0 and ssim >= 0.999999, (mode, worst, psnr, ssim)
        # the plan was BUILT in the f32 mode and then run in bf16x3: its attention calls must carry the workspace all the same
        # (sized from the shape alone), or the bf16x3 leg above would have run the slower in-loop-split kernel
        plan = m.plan_for(2 * x_T.shape[0], x_T.shape[2], x_T.shape[3], torch.device(DEV))
        att = [args for name, _, args in plan.plan.ops if name == "hdiff_mha_flash_fwd_ws"]
        assert att and all(a[7] is not None and a[8].value > 0 for a in att if a[6] >= 512), [(a[6], a[7]) for a in att]
    finally:
        hdiff_amd.set_contraction_mode(before)


def test_tree_b_ddim_64x64_matches_cpu_path():
    cfg = dict(T=1000, ch=128, ch_mult=[1, 2, 2, 2], num_res_blocks=2, dropout=0.0)
    torch.manual_seed(0)
    m = DynamicUNet(**cfg).eval()
    with torch.no_grad():
        m.tail[2].weight.mul_(2.0e4)         # initialize() gives the tail xavier gain 1e-5: eps would be ~0
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    ocfg = OB.DynUNetConfig(T=1000, ch=128, ch_mult=(1, 2, 2, 2), num_res_blocks=2)
    g = torch.Generator().manual_seed(7)
    img = torch.randint(0, 256, (1, 3, 64, 64), generator=g).float()
    y_T = torch.randn(1, 3, 64, 64, generator=g)
    ref_traj = []
    with torch.no_grad():
        ref = OB.sampler_forward(sd, ocfg, 1e-4, 0.02, 1000, img, y_T, ddim=True, ddim_step=10, trajectory=ref_traj)
    samp = SamplerB(m.to(DEV), 1e-4, 0.02, 1000).to(DEV)
    traj = []
    with torch.no_grad():
        out = samp(img.to(DEV), ddim=True, unconditional_guidance_scale=1, ddim_step=10, y_T=y_T.to(DEV), trajectory=traj)
    # With untrained weights DDIM's y0 prediction divides by sqrt(alphas_bar) ~ 6e-3 and the state grows step by step, so
    # the clipped output is +-1 almost everywhere: compare the pre-clip states, relative to their own scale, and judge
    # PSNR / SSIM on the last state normalised into [-1, 1] by the reference's own maximum.
    rels = [((a.cpu() - b).abs().max() / b.abs().max()).item() for a, b in zip(traj, ref_traj)]
    scale = ref_traj[-1].abs().max()
    psnr, ssim = quality(traj[-1].cpu() / scale, ref_traj[-1] / scale)
    print(f"tree-B DDIM 10 steps: per-step relative error {['%.1e' % r for r in rels]}, final |y| max {scale:.1f}, "
          f"PSNR {psnr:.1f} dB, SSIM {ssim:.6f}")
    assert len(traj) == 10 and max(rels) < 5e-4, rels
    assert psnr >= 40.0 and ssim >= 0.99, (psnr, ssim)
    assert torch.equal(out.cpu() == 1.0, ref == 1.0) or ((out.cpu() - ref).abs() > 1e-3).float().mean() < 1e-3
