"""Import alias: the package directory is named `flux-2-swift-mlx_amd` (not a valid
Python identifier), so `import flux2mi` resolves to it."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("flux-2-swift-mlx_amd")
sys.modules[__name__] = _pkg
