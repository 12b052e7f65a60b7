"""TEST INFRASTRUCTURE ONLY -- golden values for the underwater quality measures (hdiff_amd.uw_metrics), produced by the
reference's own numpy/scipy functions (reference_loader.load_uw_metrics).  Run:  python -m oracle.gen_golden_uw"""
import os

import numpy as np

from . import reference_loader as RL
from .gen_golden import OUT


def images():
    rng = np.random.RandomState(7)
    yy, xx = np.mgrid[0:50, 0:37]
    grad = np.stack([xx * 6.0 + yy, 255.0 - yy * 4.0, (xx * yy) % 256.0], axis=-1).astype(np.float64)
    zeros = rng.randint(0, 256, (48, 64, 3)).astype(np.float64)
    zeros[8:24, 16:40] = 0.0                                     # whole blocks of zeros (log guards)
    return {"random_64x48": rng.randint(0, 256, (64, 48, 3)).astype(np.float64), "gradient_50x37": grad,
            "zero_blocks_48x64": zeros, "bluish_40x40": np.clip(rng.normal([60, 110, 170], 25, (40, 40, 3)), 0, 255)}


def main():
    R = RL.load_uw_metrics()
    out = {}
    for name, img in images().items():
        x = img.astype(np.float32)
        out[f"{name}/image"] = img
        out[f"{name}/uicm"] = np.array(R._uicm(x))
        out[f"{name}/uism"] = np.array(R._uism(x))
        out[f"{name}/uiconm"] = np.array(R._uiconm(x, 8))
        out[f"{name}/uiqm"] = np.array(R.getUIQM(img))
        gray = img.mean(axis=2)
        out[f"{name}/eme_gray"] = np.array(R.eme(gray))
        out[f"{name}/eme_u8"] = np.array(R.eme(np.round(gray).astype(np.uint8), 8))
        out[f"{name}/logamee_gray01"] = np.array(R.logamee(gray / 255.0))
        out[f"{name}/logamee_gray255"] = np.array(R.logamee(gray))
    np.savez_compressed(os.path.join(OUT, "uw_metrics.npz"), **out)
    print("uw_metrics.npz", os.path.getsize(os.path.join(OUT, "uw_metrics.npz")))


if __name__ == "__main__":
    main()
