import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def host_threads() -> int:
    """Threads for the CPU oracle: the cgroup CPU quota when there is one (a GPU box shows 256 logical CPUs but
    grants 16 cores' worth of time — torch's default of 128 threads then runs f32 GEMMs at 0.9 TFLOP/s instead of
    1.6), else the visible CPUs."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, 2 * int(quota) // int(period)))   # 2 threads per granted core measured best
    except (OSError, ValueError):
        pass
    return n


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    import torch
    torch.set_num_threads(host_threads())


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
