"""MI355X-native classifier-free-guided DDPM hot path (gfx950 HIP kernels behind the reference's Python surface).

Import name: the directory name contains hyphens, so the repository root ships ``hdiff_amd.py``, which registers this
directory as the importable package ``hdiff_amd``:

    import hdiff_amd
    from hdiff_amd.DiffusionFreeGuidence.ModelCondition import UNet
    from hdiff_amd.DiffusionFreeGuidence.DiffusionCondition import GaussianDiffusionSampler, GaussianDiffusionTrainer

or, as a literal drop-in for the reference's imports, put this directory on ``sys.path`` and keep
``from DiffusionFreeGuidence.DiffusionCondition import ...`` unchanged.
"""
from . import _capi
from ._capi import build, lib

__all__ = ["build", "lib", "UNet", "GaussianDiffusionTrainer", "GaussianDiffusionSampler", "extract",
           "set_contraction_mode", "get_contraction_mode"]

_CONTRACT = {"f32": 0, "bf16x3": 1}


def set_contraction_mode(mode: str) -> None:
    """How the attention and 3x3-convolution contractions run (process-wide; graphs captured afterwards keep the mode they saw).

    ``"f32"`` (default): the fp32-input MFMA.  ``"bf16x3"``: every fp32 operand as three bf16 pieces, six products on the
    bf16 MFMA with fp32 accumulation -- fp32-class accuracy (tests/test_gpu_ops.py), about 1.5x faster kernels.
    The environment variable ``HDIFF_CONTRACT`` sets the initial value."""
    if mode not in _CONTRACT:
        raise ValueError(f"contraction mode must be one of {sorted(_CONTRACT)}, got {mode!r}")
    _capi.check(lib().hdiff_set_contraction_mode(_CONTRACT[mode]), "set_contraction_mode")


def get_contraction_mode() -> str:
    return {v: k for k, v in _CONTRACT.items()}[lib().hdiff_get_contraction_mode()]


def __getattr__(name):
    if name == "UNet":
        from .DiffusionFreeGuidence.ModelCondition import UNet
        return UNet
    if name in ("GaussianDiffusionTrainer", "GaussianDiffusionSampler", "extract"):
        from .DiffusionFreeGuidence import DiffusionCondition
        return getattr(DiffusionCondition, name)
    raise AttributeError(name)
