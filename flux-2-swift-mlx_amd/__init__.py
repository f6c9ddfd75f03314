"""flux-2-swift-mlx_amd — MI355X-native Flux.2 denoise + VAE-decode hot path.

The compute lives in `libflux2mi.so` (hand-written HIP for gfx950 behind the C ABI of
`include/flux2mi.h`); this package is the host-side mirror of the reference's
`Flux2Pipeline` surface plus ctypes plumbing.  Importing the package does not load
the shared library; the first API call does, and raises if it was not built.
"""
from ._lib import Flux2Error, LIB_PATH, PROTOTYPES, lib  # noqa: F401
from . import synthetic  # noqa: F401


def __getattr__(name):
    # torch-dependent modules are imported lazily
    if name in ("api", "pipeline", "parallel", "textenc"):
        import importlib
        return importlib.import_module(f"{__name__}.{name}")
    raise AttributeError(name)
