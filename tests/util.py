"""Shared helpers for the parity tests."""
import torch


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def to_dev(d, device="cuda"):
    return {k: v.to(device) for k, v in d.items()}


def oracle_cached(name: str, fingerprint_tensors, compute):
    """The slow CPU-oracle results of seeded cases as committed fixtures (tests/golden/oracle_cache_<name>.npz): `compute()` -> dict of
    tensors is what the oracle returns on inputs that are fully determined by seeds, so its result is data — inputs' CRC32 + expected
    outputs, made by THIS function on a GPU box (the synthetic weights are drawn by the device generator) with FLUX2MI_WRITE_ORACLE_CACHE
    set to a directory.  Used only when the fingerprint of the inputs actually in hand (CRC32 over the given tensors' bytes) equals the
    recorded one; otherwise — another torch / generator, an edited recipe — the oracle runs live, exactly as before.  The oracle itself
    is pinned against drift by tests/golden/*.npz on the CPU side (tests/test_golden_oracle.py)."""
    import os
    import zlib

    import numpy as np
    crc = 0
    for t in fingerprint_tensors:
        crc = zlib.crc32(t.detach().cpu().contiguous().view(torch.uint8).numpy().tobytes(), crc)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"oracle_cache_{name}.npz")
    made_by = oracle_version()
    meta = ("fingerprint", "made_by")
    if os.path.exists(path) and not os.environ.get("FLUX2MI_ORACLE_LIVE"):
        z = np.load(path)
        # the fixture stands for a run of THIS oracle source under THIS torch: an edited oracle falls back to the live run
        if int(z["fingerprint"]) == crc and "made_by" in z.files and str(z["made_by"]) == made_by:
            return {k: torch.from_numpy(z[k]) for k in z.files if k not in meta}, True
    out = compute()
    wdir = os.environ.get("FLUX2MI_WRITE_ORACLE_CACHE")
    if wdir:
        os.makedirs(wdir, exist_ok=True)
        np.savez_compressed(os.path.join(wdir, f"oracle_cache_{name}.npz"), fingerprint=np.int64(crc), made_by=np.str_(made_by),
                            **{k: v.detach().cpu().numpy() for k, v in out.items()})
    return out, False


def oracle_version() -> str:
    """What an oracle-cache fixture was made by: sha256 of oracle/flux2_ref.py + the torch version that ran it."""
    import hashlib
    import os
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "flux2_ref.py")
    with open(src, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16] + "/torch-" + torch.__version__


def tapped_stream(taps: dict, layer: int, n_double: int) -> torch.Tensor:
    """Residual stream [S, d] (text rows first) BEHIND block `layer` of an oracle forward run with `taps=` (-1: behind the embedders);
    layers count the double blocks first, then the single blocks (the order of flux2mi_dit_forward_block)."""
    if layer < 0:
        return torch.cat([taps["txt_in"][0], taps["img_in"][0]], 0)
    if layer < n_double:
        return torch.cat([taps[f"double{layer}.txt"][0], taps[f"double{layer}.img"][0]], 0)
    return taps[f"single{layer - n_double}"][0]
