"""Underwater image-quality measures used by the reference's evaluation (``utils/rotinas.py:923-928`` calls ``nmetrics`` and
``getUIQM`` of ``metrics/metrics.py``): UIQM = c1*UICM + c2*UISM + c3*UIConM, UCIQE, EME and logAMEE.  CPU / numpy: this is
reporting code, not part of the GPU hot path.

Parity:
  * ``getUIQM`` and its parts, ``eme`` and ``logamee`` are PINNED by golden vectors produced by the reference's own functions
    (they need only numpy / scipy; ``oracle/gen_golden_uw.py``), including the details that look accidental and are kept:
    the trimmed mean skips one more sample on the left than it counts (metrics.py:89-94), the spread term uses ALL samples
    around the trimmed mean (:98-102), the blue weight of UISM is 0.144 (:190), and the module's second ``eme`` (:387) --
    ceil-sized edge blocks, zero extrema bumped to 1 -- is the one ``_uism`` ends up calling;
  * ``nmetrics`` and ``uciqe`` additionally need scikit-image / OpenCV colour conversions and edge filters, absent here:
    those are restated from the published definitions (CIE Lab D65, ITU-R 709 luma, 3x3 Sobel) and are UNPINNED.
"""
from __future__ import annotations

import math

import numpy as np
from scipy import ndimage

__all__ = ["eme", "logamee", "plipsum", "plipsub", "plipmult", "uicm", "uism", "uiconm", "getUIQM", "nmetrics", "uciqe"]


# ---------------------------------------------------------------------------------------------------------------------
# block extrema (metrics.py:387-425, 435-473): ceil(H/bs) x ceil(W/bs) blocks, the last ones smaller
# ---------------------------------------------------------------------------------------------------------------------
def _block_extrema(ch: np.ndarray, bs: int):
    H, W = ch.shape
    ys, xs = np.arange(0, H, bs), np.arange(0, W, bs)
    return np.minimum.reduceat(np.minimum.reduceat(ch, ys, axis=0), xs, axis=1).astype(np.float64), \
        np.maximum.reduceat(np.maximum.reduceat(ch, ys, axis=0), xs, axis=1).astype(np.float64)


def eme(ch, blocksize: int = 8) -> float:
    """Measure of enhancement: mean over blocks of 2*log(max/min), zero extrema replaced by 1 (metrics.py:387-425)."""
    lo, hi = _block_extrema(np.asarray(ch), blocksize)
    lo = np.where(lo == 0, lo + 1, lo)
    hi = np.where(hi == 0, hi + 1, hi)
    return float((2.0 / lo.size) * np.log(hi / lo).sum())


def plipsum(i, j, gamma=1026):
    return i + j - i * j / gamma


def plipsub(i, j, k=1026):
    return k * (i - j) / (k - j)


def plipmult(c, j, gamma=1026):
    return gamma - gamma * (1 - j / gamma) ** c


def logamee(ch, blocksize: int = 8) -> float:
    """Michelson-style contrast in the PLIP arithmetic (metrics.py:435-473)."""
    lo, hi = _block_extrema(np.asarray(ch), blocksize)
    top, bottom = plipsub(hi, lo), plipsum(hi, lo)
    with np.errstate(divide="ignore", invalid="ignore"):
        m = np.where(bottom == 0, 0.0, top / bottom)
        terms = np.where(m != 0, m * np.log(m), 0.0)
    return float(plipmult(1.0 / lo.size, terms.sum()))


# ---------------------------------------------------------------------------------------------------------------------
# UIQM as getUIQM computes it (metrics.py:77-299)
# ---------------------------------------------------------------------------------------------------------------------
def _trimmed_mean(v: np.ndarray, alpha_l: float = 0.1, alpha_r: float = 0.1):
    """metrics.py:77-95: sort, drop ceil(aL*K) + 1 samples on the left and floor(aR*K) on the right, but divide by
    K - ceil(aL*K) - floor(aR*K).  The reference adds the samples one by one in their own dtype (float32 for getUIQM):
    a running sum in that dtype reproduces it."""
    v = np.sort(v, kind="stable")
    K = v.size
    t_l, t_r = math.ceil(alpha_l * K), math.floor(alpha_r * K)
    kept = v[t_l + 1:K - t_r]
    total = np.cumsum(kept, dtype=v.dtype)[-1] if kept.size else v.dtype.type(0)
    return v.dtype.type(1 / (K - t_l - t_r)) * total


def uicm(x: np.ndarray) -> float:
    """Colourfulness (metrics.py:105-117)."""
    R, G, B = (x[:, :, c].flatten() for c in range(3))
    rg, yb = R - G, ((R + G) / 2) - B
    mu_rg, mu_yb = _trimmed_mean(rg), _trimmed_mean(yb)
    # spread around the trimmed mean over ALL samples (:98-102): differences in the image dtype, squares summed in double
    var_rg = _seq_sum((rg - mu_rg).astype(np.float64) ** 2) / rg.size
    var_yb = _seq_sum((yb - mu_yb).astype(np.float64) ** 2) / yb.size
    return (-0.0268 * math.sqrt(float(mu_rg) ** 2 + float(mu_yb) ** 2)) + (0.1586 * math.sqrt(var_rg + var_yb))


def _sobel_scaled(x: np.ndarray) -> np.ndarray:
    mag = np.hypot(ndimage.sobel(x, 0), ndimage.sobel(x, 1))
    mag *= 255.0 / np.max(mag)
    return mag


def uism(x: np.ndarray) -> float:
    """Sharpness: EME of each channel times its Sobel magnitude (metrics.py:164-193; weights 0.299 / 0.587 / 0.144)."""
    vals = [eme(_sobel_scaled(x[:, :, c]) * x[:, :, c], 8) for c in range(3)]
    return (0.299 * vals[0]) + (0.587 * vals[1]) + (0.144 * vals[2])


def uiconm(x: np.ndarray, window_size: int = 8) -> float:
    """Contrast: -mean over full blocks of (d/s) * log(d/s), d = max - min, s = max + min over all channels (:234-279)."""
    k1, k2 = x.shape[1] // window_size, x.shape[0] // window_size
    x = x[:window_size * k2, :window_size * k1]
    blocks = x.reshape(k2, window_size, k1, window_size, -1)
    hi, lo = blocks.max(axis=(1, 3, 4)), blocks.min(axis=(1, 3, 4))
    top, bot = hi - lo, hi + lo                      # in the image dtype, like the reference's scalars
    with np.errstate(divide="ignore", invalid="ignore"):
        r = (top / bot).astype(np.float64)
        ok = ~(np.isnan(top) | np.isnan(bot) | (bot == 0.0) | (top == 0.0))
        terms = np.where(ok, r * np.log(r), 0.0)
    return (-1.0 / (k1 * k2)) * _seq_sum(terms.T.flatten())     # accumulated column by column, as the reference loops


def _seq_sum(v: np.ndarray) -> float:
    """Left-to-right sum in double (the reference's Python accumulation loops)."""
    return float(np.cumsum(np.asarray(v, dtype=np.float64))[-1]) if v.size else 0.0


def getUIQM(x) -> float:
    """UIQM of an RGB image (H, W, 3) with values in [0, 255] (metrics.py:282-299)."""
    x = np.asarray(x).astype(np.float32)
    return (0.0282 * uicm(x)) + (0.2953 * uism(x)) + (3.5753 * uiconm(x, 8))


# ---------------------------------------------------------------------------------------------------------------------
# UNPINNED part: colour conversions / edge filter of scikit-image and OpenCV restated from their published definitions
# ---------------------------------------------------------------------------------------------------------------------
def _srgb_to_lab(rgb01: np.ndarray) -> np.ndarray:
    """CIE L*a*b* (D65, 2 degree observer) from sRGB in [0, 1]."""
    lin = np.where(rgb01 > 0.04045, ((rgb01 + 0.055) / 1.055) ** 2.4, rgb01 / 12.92)
    m = np.array([[0.412453, 0.357580, 0.180423], [0.212671, 0.715160, 0.072169], [0.019334, 0.119193, 0.950227]])
    xyz = lin @ m.T / np.array([0.95047, 1.0, 1.08883])
    f = np.where(xyz > 0.008856, np.cbrt(xyz), 7.787 * xyz + 16.0 / 116.0)
    return np.stack([116.0 * f[..., 1] - 16.0, 500.0 * (f[..., 0] - f[..., 1]), 200.0 * (f[..., 1] - f[..., 2])], axis=-1)


def _as_float01(a: np.ndarray) -> np.ndarray:
    a = np.asarray(a)
    return a.astype(np.float64) / 255.0 if a.dtype == np.uint8 else a.astype(np.float64)


def _sobel_normalised(x: np.ndarray) -> np.ndarray:
    """sqrt((Sobel_h^2 + Sobel_v^2) / 2) with kernels scaled by 1/4 (the normalisation scikit-image documents)."""
    return np.sqrt((ndimage.sobel(x, 0, mode="reflect") / 4.0) ** 2 + (ndimage.sobel(x, 1, mode="reflect") / 4.0) ** 2) / math.sqrt(2.0)


def nmetrics(a):
    """(uiqm, uciqe, uism, uicm, uiconm) as metrics.py:301-385 composes them.  UNPINNED (colour conversion / Sobel restated)."""
    rgb = _as_float01(a)
    lab = _srgb_to_lab(rgb)
    gray = rgb @ np.array([0.2125, 0.7154, 0.0721])
    lum = lab[:, :, 0]
    chroma = np.sqrt(lab[:, :, 1] ** 2 + lab[:, :, 2] ** 2)
    sc = float(np.sqrt(np.mean((chroma - chroma.mean()) ** 2)))
    top = int(np.round(0.01 * lum.size))
    sl = np.sort(lum, axis=None)
    conl = float(np.mean(sl[::-1][:top]) - np.mean(sl[:top]))
    with np.errstate(divide="ignore", invalid="ignore"):
        satur = np.where((chroma == 0) | (lum == 0), 0.0, chroma / lum)
    uciqe_v = 0.4680 * sc + 0.2745 * conl + 0.2576 * float(satur.mean())

    rg = np.sort(rgb[:, :, 0] - rgb[:, :, 1], axis=None)
    yb = np.sort((rgb[:, :, 0] + rgb[:, :, 1]) / 2 - rgb[:, :, 2], axis=None)
    t1 = int(0.1 * rg.size)
    rg, yb = rg[t1:-t1], yb[t1:-t1]
    uicm_v = -0.0268 * np.sqrt(rg.mean() ** 2 + yb.mean() ** 2) + 0.1586 * np.sqrt(rg.var() + yb.var())
    emes = [eme(np.round(rgb[:, :, c] * _sobel_normalised(rgb[:, :, c])).astype(np.uint8)) for c in range(3)]
    uism_v = 0.299 * emes[0] + 0.587 * emes[1] + 0.114 * emes[2]
    uiconm_v = logamee(gray)
    return 0.0282 * uicm_v + 0.2953 * uism_v + 3.5753 * uiconm_v, uciqe_v, uism_v, float(uicm_v), uiconm_v


def uciqe(nargin, loc):
    """metrics.py:40-76 (8-bit OpenCV Lab: L*255/100, a+128, b+128, rounded).  UNPINNED."""
    lab = _srgb_to_lab(_as_float01(loc))
    lab8 = np.clip(np.round(np.stack([lab[..., 0] * 255.0 / 100.0, lab[..., 1] + 128.0, lab[..., 2] + 128.0], axis=-1)), 0, 255)
    lum, a, b = (lab8[..., c] / 255 for c in range(3))
    chr_ = np.sqrt(np.square(a) + np.square(b))
    with np.errstate(divide="ignore", invalid="ignore"):
        sat = chr_ / np.sqrt(np.square(chr_) + np.square(lum))
        var_chr = np.sqrt(np.mean(abs(1 - np.square(np.mean(chr_) / chr_))))
    hist, _ = np.histogram(lum, 65536)
    cdf = np.cumsum(hist) / np.sum(hist)
    ilow, ihigh = np.where(cdf > 0.0100)[0][0], np.where(cdf >= 0.9900)[0][0]
    con_lum = (ihigh - 1) / 65535 - (ilow - 1) / 65535
    return 0.4680 * var_chr + 0.2745 * con_lum + 0.2576 * float(np.mean(sat))
