"""Underwater quality measures (hdiff_amd.uw_metrics) against golden values computed by the reference's own numpy/scipy
functions (oracle/gen_golden_uw.py); the scikit-image / OpenCV dependent parts are unpinned and only sanity-checked."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hdiff_amd  # noqa: E402,F401
from hdiff_amd import uw_metrics as U  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden", "uw_metrics.npz")


def test_uiqm_family_matches_reference_functions():
    d = np.load(GOLDEN)
    names = sorted({k.split("/")[0] for k in d.files})
    assert len(names) == 4
    for n in names:
        img = d[f"{n}/image"]
        x, gray = img.astype(np.float32), img.mean(axis=2)
        got = {"uicm": U.uicm(x), "uism": U.uism(x), "uiconm": U.uiconm(x, 8), "uiqm": U.getUIQM(img), "eme_gray": U.eme(gray),
               "eme_u8": U.eme(np.round(gray).astype(np.uint8), 8), "logamee_gray01": U.logamee(gray / 255.0),
               "logamee_gray255": U.logamee(gray)}
        for k, v in got.items():
            ref = float(d[f"{n}/{k}"])
            assert abs(v - ref) <= 1e-12 * max(1.0, abs(ref)), (n, k, v, ref)


def test_block_measures_against_brute_force():
    """eme / logamee restated with explicit loops (ceil-sized edge blocks, zero extrema -> 1, PLIP arithmetic)."""
    rng = np.random.RandomState(0)
    ch = rng.rand(21, 30) * 200
    ch[0:8, 0:8] = 0
    e, s, nb = 0.0, 0.0, 0
    for i in range(0, 21, 8):
        for j in range(0, 30, 8):
            blk = ch[i:i + 8, j:j + 8]
            lo, hi = float(blk.min()), float(blk.max())
            nb += 1
            e += np.log((hi if hi else 1.0) / (lo if lo else 1.0))
            top, bot = 1026 * (hi - lo) / (1026 - lo), hi + lo - hi * lo / 1026
            m = top / bot if bot else 0.0
            s += m * np.log(m) if m else 0.0
    assert abs(U.eme(ch) - 2.0 * e / nb) < 1e-12
    assert abs(U.logamee(ch) - (1026 - 1026 * (1 - s / 1026) ** (1.0 / nb))) < 1e-12


def test_unpinned_measures_behave():
    rng = np.random.RandomState(1)
    vivid = np.clip(rng.normal([200, 60, 40], 40, (48, 48, 3)), 0, 255).astype(np.float32)
    dull = np.clip(rng.normal([110, 120, 125], 6, (48, 48, 3)), 0, 255).astype(np.float32)
    for img in (vivid, dull):
        vals = U.nmetrics(img)
        assert len(vals) == 5 and all(np.isfinite(v) for v in vals)
    assert U.uciqe(1, vivid.astype(np.uint8)) > U.uciqe(1, dull.astype(np.uint8))      # more chroma spread and contrast
    assert U.getUIQM(vivid) > U.getUIQM(dull)
