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

__all__ = ["build", "lib", "UNet", "GaussianDiffusionTrainer", "GaussianDiffusionSampler", "extract"]


def __getattr__(name):
    if name == "UNet":
        from .DiffusionFreeGuidence.ModelCondition import UNet
        return UNet
    if name in ("GaussianDiffusionTrainer", "GaussianDiffusionSampler", "extract"):
        from .DiffusionFreeGuidence import DiffusionCondition
        return getattr(DiffusionCondition, name)
    raise AttributeError(name)
