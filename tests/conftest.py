import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def flux2mi():
    """The package (alias module); building the library first if it is missing."""
    import __graft_entry__ as g

    lib_path = os.path.join(ROOT, "flux-2-swift-mlx_amd", "libflux2mi.so")
    if not os.path.exists(lib_path):
        g.build()
    import flux2mi as pkg

    return pkg


@pytest.fixture(scope="session")
def ctx(flux2mi):
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from flux2mi.api import Context

    c = Context(0)
    yield c
    c.close()
