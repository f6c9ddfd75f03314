"""CPU: the hand-scheduled kernels' ISA invariants (tools/check_isa.py) on the device assembly the product build left next to the
objects it linked (`-save-temps=obj`, csrc/Makefile: build/isa.ok is a prerequisite of libflux2mi.so).  gemm256s.hip / attention64b.hip
keep in-flight ds_read / LDS-DMA / MFMA results in registers the compiler believes ready and count vmcnt / lgkmcnt by hand: a register
copy, a spill, a vector-memory instruction or an M0 use of hipcc's own inside those loops silently breaks them."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "flux-2-swift-mlx_amd", "csrc", "build")
NAMES = ["gemm256s", "attention64b", "attention64b_doff", "attention64b_qk8", "attention64b_qk8_doff"]


def test_isa_invariants_of_the_linked_objects(flux2mi):
    missing = [n for n in NAMES if not os.path.exists(os.path.join(BUILD, n + "-hip-amdgcn-amd-amdhsa-gfx950.s"))]
    if missing:
        pytest.skip("no device assembly next to the objects (library built elsewhere): " + ", ".join(missing))
    lib = os.path.join(ROOT, "flux-2-swift-mlx_amd", "libflux2mi.so")
    for n in NAMES:   # the assembly belongs to the objects that were linked: not older than its source, the library not older than it
        s = os.path.join(BUILD, n + "-hip-amdgcn-amd-amdhsa-gfx950.s")
        assert os.path.getmtime(lib) >= os.path.getmtime(s) - 1.0
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_isa.py"), "--from", BUILD], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "all ISA invariants hold" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def test_makefile_links_only_behind_the_check():
    mk = open(os.path.join(ROOT, "flux-2-swift-mlx_amd", "csrc", "Makefile")).read()
    assert "$(OUT): $(OBJ) $(BUILD)/isa.ok" in mk and "check_isa.py --from $(BUILD)" in mk
