"""CPU: the C-ABI library loads and exports every symbol include/flux2mi.h declares; no
compute call is made (there is no GPU here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "flux2mi.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = set(re.findall(r"\b(flux2mi_[a-z0-9_]+)\s*\(", src))
    # typedef'd function-pointer types are not symbols
    return {n for n in names if not re.search(r"\(\*" + n + r"\)", src)}


def test_every_declared_symbol_is_exported_and_bound(flux2mi):
    names = declared_symbols()
    assert len(names) >= 40
    lib = flux2mi.lib()
    raw = ctypes.CDLL(flux2mi.LIB_PATH)
    for n in sorted(names):
        assert hasattr(raw, n), f"{n} declared in flux2mi.h but not exported by libflux2mi.so"
        assert n in flux2mi.PROTOTYPES, f"{n} has no ctypes prototype"
    for n in flux2mi.PROTOTYPES:
        assert n in names, f"{n} bound in _lib.py but not declared in flux2mi.h"
    assert lib.flux2mi_version() == 100


def test_host_functions_report_errors_not_abort(flux2mi):
    from flux2mi import api
    import pytest
    with pytest.raises(flux2mi.Flux2Error) as e:
        api.set_timesteps(0, 4096)
    assert e.value.code == 1 and "steps" in e.value.message


def test_presets_match_reference_config(flux2mi):
    """Flux2Config.swift:258-329, VAEConfig.swift:44-81."""
    import ctypes as C
    from flux2mi import _lib as L
    exp = {"klein-4b": (3072, 24, 5, 20, 7680, 0), "klein-9b": (4096, 32, 8, 24, 12288, 0), "dev": (6144, 48, 8, 48, 15360, 1)}
    for name, vals in exp.items():
        cfg = L.DitConfig()
        L.check(flux2mi.lib().flux2mi_dit_preset(name.encode(), C.byref(cfg)))
        assert (cfg.dim, cfg.heads, cfg.n_double, cfg.n_single, cfg.joint_dim, cfg.guidance_embeds) == vals
        p = flux2mi.synthetic.DIT_PRESETS[name]
        assert (p["dim"], p["heads"], p["n_double"], p["n_single"], p["joint_dim"], int(p["guidance"])) == vals
    for name, ch in {"standard": [128, 256, 512, 512], "small": [96, 192, 384, 384]}.items():
        v = L.VaeConfig()
        L.check(flux2mi.lib().flux2mi_vae_preset(name.encode(), C.byref(v)))
        assert list(v.channels) == ch and v.latent_channels == 32 and v.groups == 32 and v.layers_per_block == 2


def test_param_counts_match_published_sizes(flux2mi):
    """bf16 bytes: Klein-4B 7.8 GB, Klein-9B 18.2 GB, Dev 64.4 GB."""
    syn = flux2mi.synthetic
    for name, gb in {"klein-4b": 7.75, "klein-9b": 18.16, "dev": 64.4}.items():
        n = sum(int(__import__("numpy").prod(s)) for _, s, _ in syn.dit_param_shapes(syn.DIT_PRESETS[name]))
        assert abs(n * 2 / 1e9 - gb) / gb < 0.02, (name, n * 2 / 1e9)


def test_missing_library_fails_loudly():
    """No CPU / PyTorch fallback: with the shared object absent the first API call raises ImportError
    naming the build command (run in a child process with FLUX2MI_LIB pointing at a missing file)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); import flux2mi\n"
            "try:\n    flux2mi.lib()\n    print('loaded')\nexcept ImportError as e:\n    print('ImportError', 'no cpu fallback' in str(e).lower())\n") % root
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, FLUX2MI_LIB="libflux2mi_absent.so"),
                       capture_output=True, text=True, timeout=120)
    assert "ImportError True" in r.stdout, r.stdout + r.stderr
