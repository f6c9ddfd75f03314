#!/usr/bin/env python3
"""Writes tests/golden/imaging.npz: Pillow's `Image.resize(size, BILINEAR, box=)` on seeded u8 inputs — the
external known answers `oracle/imaging_ref.resample_u8` (and through it the HIP path) is pinned on.
Usage: python tests/golden/make_golden_imaging.py   (needs Pillow; 12.2.0 here)"""
import os

import numpy as np
from PIL import Image

OUT = os.path.dirname(os.path.abspath(__file__))
# (name, input shape, (out_h, out_w), box or None)
CASES = [
    ("gray_down4", (100, 200), (25, 50), None),
    ("gray_down16", (128, 256), (8, 16), None),
    ("gray_ragged", (97, 131), (64, 64), None),
    ("rgb_up", (48, 40, 3), (96, 60), None),
    ("rgb_identity", (33, 35, 3), (33, 35), None),
    ("rgb_box", (120, 160, 3), (32, 40), (10.5, 20.25, 150.0, 100.75)),
    ("gray_box_up", (61, 77), (100, 150), (3.0, 5.0, 70.0, 50.0)),
    ("rgba_down", (64, 64, 4), (16, 16), None),
    ("gray_1px", (40, 40), (1, 1), None),
]


def pil_resize(a, out_hw, box):
    planes = a[:, :, None] if a.ndim == 2 else a
    outs = [np.asarray(Image.fromarray(np.ascontiguousarray(planes[:, :, c]), "L").resize((out_hw[1], out_hw[0]), Image.BILINEAR, box=box))
            for c in range(planes.shape[2])]
    r = np.stack(outs, -1)
    return r[:, :, 0] if a.ndim == 2 else r


def make_input(name, shape):
    rng = np.random.default_rng(sum(map(ord, name)))
    a = rng.integers(0, 256, shape, dtype=np.uint8)
    if name.startswith("gray_down"):   # a hard-edged mask, the realistic case
        a[:] = 0
        a[shape[0] // 5: shape[0] // 2, shape[1] // 3: shape[1] - 7] = 255
    return a


def main():
    d = {}
    for name, shape, out_hw, box in CASES:
        a = make_input(name, shape)
        d[name + "_in"] = a
        d[name + "_out"] = pil_resize(a, out_hw, box)
    import PIL
    d["pillow_version"] = np.array(PIL.__version__)
    np.savez_compressed(os.path.join(OUT, "imaging.npz"), **d)
    print("wrote imaging.npz", {k: v.shape for k, v in d.items() if k.endswith("_out")})


if __name__ == "__main__":
    main()
