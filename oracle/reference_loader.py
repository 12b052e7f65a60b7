"""TEST INFRASTRUCTURE ONLY.  Load the real reference classes for golden-vector generation.

The reference (``/root/reference``) exists only in the build container.  It is
read-only and never copied: this module loads it *in place*.

* ``DiffusionFreeGuidence/DiffusionCondition.py`` imports cleanly by file path.
* ``DiffusionFreeGuidence/ModelCondition.py`` has a ``SyntaxError`` at line 289
  (inside ``DynamicUNet``, a class the hot path never uses), so only the text of
  lines 1-277 (everything through ``class UNet``) is compiled, straight from the
  file where it lies.
"""
from __future__ import annotations

import importlib.util
import os
import types

REFERENCE_ROOT = os.environ.get("HDIFF_REFERENCE_ROOT", "/root/reference")
_MODEL_LIVE_LINES = 277


def available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "DiffusionFreeGuidence", "DiffusionCondition.py"))


def load_diffusion() -> types.ModuleType:
    path = os.path.join(REFERENCE_ROOT, "DiffusionFreeGuidence", "DiffusionCondition.py")
    spec = importlib.util.spec_from_file_location("_ref_DiffusionCondition", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_model() -> types.ModuleType:
    path = os.path.join(REFERENCE_ROOT, "DiffusionFreeGuidence", "ModelCondition.py")
    with open(path, "r") as fh:
        live = "".join(fh.readlines()[:_MODEL_LIVE_LINES])
    mod = types.ModuleType("_ref_ModelCondition")
    mod.__file__ = path
    exec(compile(live, path, "exec"), mod.__dict__)
    return mod


def load_scheduler() -> types.ModuleType:
    path = os.path.join(REFERENCE_ROOT, "Scheduler.py")
    spec = importlib.util.spec_from_file_location("_ref_Scheduler", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_model_b() -> types.ModuleType:
    """diffusion/Model.py (DynamicUNet and its blocks) imports cleanly by file path (torch only)."""
    path = os.path.join(REFERENCE_ROOT, "diffusion", "Model.py")
    spec = importlib.util.spec_from_file_location("_ref_diffusion_Model", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_sampler_b() -> types.ModuleType:
    """``extract`` and ``class GaussianDiffusionSampler`` of diffusion/Diffusion.py.

    The file's own imports need cv2, lpips and ``Loss.loss`` (absent here: ordinary ImportError) and its trainer needs
    pretrained perceptual networks, so only the text of the two live definitions is compiled, from the file where it
    lies: ``def extract`` up to ``class GaussianDiffusionTrainer``, and ``class GaussianDiffusionSampler`` up to the
    commented-out old code.  Nothing is copied into the repository."""
    path = os.path.join(REFERENCE_ROOT, "diffusion", "Diffusion.py")
    with open(path, "r") as fh:
        lines = fh.readlines()

    def find(prefix: str, start: int = 0) -> int:
        for i in range(start, len(lines)):
            if lines[i].startswith(prefix):
                return i
        raise RuntimeError(f"reference layout changed: no line starts with {prefix!r}")

    a0, a1 = find("def extract"), find("class GaussianDiffusionTrainer")
    b0 = find("class GaussianDiffusionSampler")
    b1 = find("#####", b0)
    # keep the original line numbers in tracebacks: blank out everything that is not compiled
    kept = ["\n"] * len(lines)
    kept[a0:a1] = lines[a0:a1]
    kept[b0:b1] = lines[b0:b1]
    mod = types.ModuleType("_ref_diffusion_Diffusion")
    mod.__file__ = path
    exec("import torch\nimport torch.nn as nn\nimport torch.nn.functional as F\n", mod.__dict__)
    exec(compile("".join(kept), path, "exec"), mod.__dict__)
    return mod


def load_uw_metrics() -> types.ModuleType:
    """The numpy/scipy-only quality measures of metrics/metrics.py (``mu_a`` ... ``getUIQM``, ``eme``, ``logamee``; lines from
    ``def mu_a`` up to ``class FID``).  The file's own imports need scikit-image, torchvision and OpenCV (absent: ordinary
    ImportError), so only that span is compiled, in place; ``nmetrics`` inside it is defined but never called here."""
    path = os.path.join(REFERENCE_ROOT, "metrics", "metrics.py")
    with open(path, "r") as fh:
        lines = fh.readlines()
    a = next(i for i, l in enumerate(lines) if l.startswith("def mu_a"))
    b = next(i for i, l in enumerate(lines) if l.startswith("class FID"))
    kept = ["\n"] * len(lines)
    kept[a:b] = lines[a:b]
    mod = types.ModuleType("_ref_metrics")
    mod.__file__ = path
    exec("import math\nimport numpy as np\nfrom scipy import ndimage\n", mod.__dict__)
    exec(compile("".join(kept), path, "exec"), mod.__dict__)
    return mod
