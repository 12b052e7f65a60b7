"""Import alias: makes the directory ``hybrid-diffusion-underwater-atmopheric-image-enhancement_amd/`` (whose name is not a
valid Python identifier) importable as the package ``hdiff_amd``."""
import importlib.util
import os
import sys

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                    "hybrid-diffusion-underwater-atmopheric-image-enhancement_amd")
_spec = importlib.util.spec_from_file_location("hdiff_amd", os.path.join(_DIR, "__init__.py"),
                                               submodule_search_locations=[_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["hdiff_amd"] = _mod
_spec.loader.exec_module(_mod)
