"""Drop-ins for the inference side of the reference's second tree (``diffusion/Model.py``, ``diffusion/Diffusion.py``):
the image-conditioned ``DynamicUNet`` and its ancestral / DDIM ``GaussianDiffusionSampler``, on the same gfx950 kernels."""
from . import Diffusion, Model  # noqa: F401
