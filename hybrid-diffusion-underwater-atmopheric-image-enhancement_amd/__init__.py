"""MI355X-native classifier-free-guided DDPM hot path (gfx950 HIP kernels behind the reference's Python surface).

Import name: the directory name contains hyphens, so the repository root ships ``hdiff_amd.py``, which registers this
directory as the importable package ``hdiff_amd``:

    import hdiff_amd
    from hdiff_amd.DiffusionFreeGuidence.ModelCondition import UNet
    from hdiff_amd.DiffusionFreeGuidence.DiffusionCondition import GaussianDiffusionSampler, GaussianDiffusionTrainer

or, as a literal drop-in for the reference's own import lines (``MainCondition.py:1``, ``TrainCondition.py:15-17``):

    import hdiff_amd; hdiff_amd.install_dropin()
    from DiffusionFreeGuidence.TrainCondition import train, eval        # unchanged reference code from here on

(INTEGRATION.md section 1).
"""
import importlib
import sys

from . import _capi
from ._capi import build, lib

__all__ = ["build", "lib", "install_dropin", "UNet", "GaussianDiffusionTrainer", "GaussianDiffusionSampler", "extract",
           "set_contraction_mode", "get_contraction_mode", "reserve_split_workspace"]

_CONTRACT = {"f32": 0, "bf16x3": 1}


def set_contraction_mode(mode: str) -> None:
    """How the attention and 3x3-convolution contractions run (process-wide; graphs captured afterwards keep the mode they saw).

    ``"bf16x3"`` (default; the name is round 3's): the SPLIT-OPERAND mode -- every fp32 operand of a contraction is carried as 16-bit pieces
    and the product as a few piece products on the 16-bit MFMA, each exact in the fp32 accumulator.  As shipped: every operand of the
    attention contractions is an fp16 PAIR (x 2^s = h0 + h1: 22-23 of fp32's 24 bits) -- the scores with a balance per product term
    (four terms at d_head 16, three at d_head 32: attention_h2.hip / attention_x3p.hip), P.V three terms, the backward's five products
    likewise (attention_bwd_h2.hip) -- and so are both operands of the 3x3 convolutions behind GroupNorm + Swish; 3x3 convolutions
    without a known input range and the 1x1 convolutions run on bf16 TRIPLES (x = x0 + x1 + x2 exactly, the six products with
    i + j <= 2).  fp32-class accuracy: the error against float64 is <= 1.25x (attention) / 1.5x (conv) the fp32 kernels' rms
    (tests/test_gpu_ops.py, tests/test_gpu_backward.py), the whole golden / oracle suite passes in this mode at the fp32 tolerances
    (tests/conftest.py), mutants of each dropped-term class turn the gates red (tests/test_gpu_mutation.py).  ``"f32"``: the
    fp32-input MFMA (an exact k-ordered fma chain), about 2.1x slower per step at 256x256.  The environment variable
    ``HDIFF_CONTRACT`` sets the initial value."""
    if mode not in _CONTRACT:
        raise ValueError(f"contraction mode must be one of {sorted(_CONTRACT)}, got {mode!r}")
    _capi.check(lib().hdiff_set_contraction_mode(_CONTRACT[mode]), "set_contraction_mode")


def reserve_split_workspace(flag: bool) -> None:
    """Whether plans built from now on reserve the scratch of the split-operand attention kernels (default True: plans are keyed by shape,
    not by mode).  ``False`` is for deployments that stay in the ``"f32"`` mode: it saves 18 bytes per qkv element per attention block
    (engine.Plan.attention_workspace has the numbers); such a plan still runs correctly in the split-operand mode, on the slower
    in-loop-split kernel."""
    from . import engine
    engine.RESERVE_SPLIT_WORKSPACE = bool(flag)


def get_contraction_mode() -> str:
    return {v: k for k, v in _CONTRACT.items()}[lib().hdiff_get_contraction_mode()]


_DROPIN_MODULES = (
    # name the reference imports            module of this package that answers it
    ("DiffusionFreeGuidence", "DiffusionFreeGuidence"),
    ("DiffusionFreeGuidence.DiffusionCondition", "DiffusionFreeGuidence.DiffusionCondition"),
    ("DiffusionFreeGuidence.ModelCondition", "DiffusionFreeGuidence.ModelCondition"),
    ("DiffusionFreeGuidence.TrainCondition", "DiffusionFreeGuidence.TrainCondition"),
    ("Scheduler", "Scheduler"),                  # TrainCondition.py:17 `from Scheduler import GradualWarmupScheduler`
    ("MainCondition", "MainCondition"),
)
_DROPIN_MODULES_TREE_B = (
    ("diffusion", "diffusion"),                  # utils/rotinas.py:17-18 `from diffusion.Diffusion import ...`
    ("diffusion.Diffusion", "diffusion.Diffusion"),
    ("diffusion.Model", "diffusion.Model"),
)


def install_dropin(second_tree: bool = True, force: bool = False) -> dict:
    """Make the reference's OWN import statements resolve to this package, without touching the reference's files:

        from DiffusionFreeGuidence.TrainCondition import train, eval                         (MainCondition.py:1)
        from DiffusionFreeGuidence.DiffusionCondition import GaussianDiffusionSampler, GaussianDiffusionTrainer
        from DiffusionFreeGuidence.ModelCondition import UNet
        from Scheduler import GradualWarmupScheduler                                          (TrainCondition.py:15-17)
        from diffusion.Diffusion import GaussianDiffusionSampler; from diffusion.Model import DynamicUNet   (second tree)

    Every module is imported under its real name (``hdiff_amd.DiffusionFreeGuidence.TrainCondition`` ...: their relative
    imports into this package are resolved there) and the SAME module object is then registered under the reference's
    name.  Registering only the package would not do: the import system would re-load ``TrainCondition.py`` from the
    package's ``__path__`` as a top-level ``DiffusionFreeGuidence.TrainCondition`` and its ``from .. import parallel`` would
    fail.  Call it before the reference's modules are imported (``sys.modules`` wins over ``sys.path``, so it also works
    from inside the reference's checkout); a name already imported from somewhere else raises unless ``force``.
    Returns {reference name: module}."""
    table = _DROPIN_MODULES + (_DROPIN_MODULES_TREE_B if second_tree else ())
    mods = {ref: importlib.import_module(f"{__name__}.{own}") for ref, own in table}
    for ref, mod in mods.items():
        old = sys.modules.get(ref)
        if old is not None and old is not mod and not force:
            raise ImportError(f"install_dropin: '{ref}' is already imported from {getattr(old, '__file__', old)!r}; call "
                              "install_dropin() before the reference's modules are imported (or pass force=True)")
    for ref, mod in mods.items():
        sys.modules[ref] = mod
    return mods


def __getattr__(name):
    if name == "UNet":
        from .DiffusionFreeGuidence.ModelCondition import UNet
        return UNet
    if name in ("GaussianDiffusionTrainer", "GaussianDiffusionSampler", "extract"):
        from .DiffusionFreeGuidence import DiffusionCondition
        return getattr(DiffusionCondition, name)
    raise AttributeError(name)
