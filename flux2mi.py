"""Import alias: the package directory is named `flux-2-swift-mlx_amd` (not a valid
Python identifier), so `import flux2mi` / `from flux2mi.api import ...` resolve to it.
Submodules are registered under both names so that there is exactly one copy of each."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_REAL = "flux-2-swift-mlx_amd"
_pkg = importlib.import_module(_REAL)
for _sub in ("_lib", "synthetic", "api", "pipeline", "parallel", "textenc"):
    _m = importlib.import_module(f"{_REAL}.{_sub}")
    sys.modules[f"{__name__}.{_sub}"] = _m
    setattr(_pkg, _sub, _m)
sys.modules[__name__] = _pkg
