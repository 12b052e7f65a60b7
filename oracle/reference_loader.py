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
