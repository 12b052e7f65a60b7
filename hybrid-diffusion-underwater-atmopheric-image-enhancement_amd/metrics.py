"""PSNR / SSIM as the reference's evaluation calls them (utils/rotinas.py:922, 926, 1170, 1174:
``skimage.metrics.peak_signal_noise_ratio(ref, img, data_range=255)`` and
``structural_similarity(ref, img, channel_axis=2, data_range=255)`` on HWC uint8 images).

scikit-image (reference pin 0.22.0, CLEDiff_bkp.yaml:293) is not installed in this image, so both are restated from the
published definitions with skimage's defaults; PARITY UNPINNED by the reference (it records no metric values):
  PSNR = 10 log10(R^2 / MSE)
  SSIM (Wang et al. 2004): 7x7 uniform window, K1 = 0.01, K2 = 0.03, sample covariance (N/(N-1)), borders of
  (win-1)//2 pixels cropped before averaging, channels averaged.
CPU / numpy: these are evaluation-side metrics, not part of the device hot path.
"""
from __future__ import annotations

import numpy as np
from scipy.ndimage import uniform_filter


def psnr(image_true, image_test, data_range: float = 255.0) -> float:
    a = np.asarray(image_true, dtype=np.float64)
    b = np.asarray(image_test, dtype=np.float64)
    mse = np.mean((a - b) ** 2)
    if mse == 0:
        return float("inf")
    return float(10.0 * np.log10(data_range ** 2 / mse))


def _ssim_plane(x: np.ndarray, y: np.ndarray, data_range: float, win: int, k1: float, k2: float) -> float:
    npix = win * win
    cov_norm = npix / (npix - 1.0)
    ux, uy = uniform_filter(x, size=win), uniform_filter(y, size=win)
    uxx, uyy, uxy = uniform_filter(x * x, size=win), uniform_filter(y * y, size=win), uniform_filter(x * y, size=win)
    vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
    c1, c2 = (k1 * data_range) ** 2, (k2 * data_range) ** 2
    s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2))
    pad = (win - 1) // 2
    return float(s[pad:-pad, pad:-pad].mean())


def ssim(image_true, image_test, data_range: float = 255.0, channel_axis: int | None = 2, win_size: int = 7,
         k1: float = 0.01, k2: float = 0.03) -> float:
    a = np.asarray(image_true, dtype=np.float64)
    b = np.asarray(image_test, dtype=np.float64)
    if a.shape != b.shape:
        raise ValueError("Input images must have the same dimensions.")
    if channel_axis is None:
        if min(a.shape) < win_size:
            raise ValueError("win_size exceeds image extent.")
        return _ssim_plane(a, b, data_range, win_size, k1, k2)
    a, b = np.moveaxis(a, channel_axis, -1), np.moveaxis(b, channel_axis, -1)
    if min(a.shape[:2]) < win_size:
        raise ValueError("win_size exceeds image extent.")
    return float(np.mean([_ssim_plane(a[..., c], b[..., c], data_range, win_size, k1, k2) for c in range(a.shape[-1])]))


def batch_psnr_ssim(ref, img):
    """Mean PSNR / SSIM over a batch of [N,3,H,W] tensors in [0,1], evaluated as the reference does: HWC uint8, range 255."""
    import torch
    r = (torch.as_tensor(ref).detach().float().cpu().clamp(0, 1) * 255.0 + 0.5).to(torch.uint8).permute(0, 2, 3, 1).numpy()
    t = (torch.as_tensor(img).detach().float().cpu().clamp(0, 1) * 255.0 + 0.5).to(torch.uint8).permute(0, 2, 3, 1).numpy()
    ps = [psnr(a, b, 255.0) for a, b in zip(r, t)]
    ss = [ssim(a, b, 255.0, channel_axis=2) for a, b in zip(r, t)]
    return float(np.mean(ps)), float(np.mean(ss))
