"""Import shim: the product package lives in the directory ``gpu-sort_amd/``
(a name Python cannot import directly), so ``import gpu_sort_amd`` loads that
directory as the package ``gpu_sort_amd``."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpu-sort_amd")
_spec = importlib.util.spec_from_file_location(
    "gpu_sort_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["gpu_sort_amd"] = _mod
_spec.loader.exec_module(_mod)
