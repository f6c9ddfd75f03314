"""Shared by the CPU (oracle) and GPU (product) imaging tests: the golden case table and the synthetic images of the
reference's own XCTest file (Tests/Flux2ChainsTests/Flux2InpaintCompositingTests.swift)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "imaging.npz")
# name -> ((out_h, out_w), box)
CASES = {
    "gray_down4": ((25, 50), None), "gray_down16": ((8, 16), None), "gray_ragged": ((64, 64), None),
    "rgb_up": ((96, 60), None), "rgb_identity": ((33, 35), None), "rgb_box": ((32, 40), (10.5, 20.25, 150.0, 100.75)),
    "gray_box_up": ((100, 150), (3.0, 5.0, 70.0, 50.0)), "rgba_down": ((16, 16), None), "gray_1px": ((1, 1), None),
}


def gray_mask(width, height, white_rect=None):
    """grayMask(width:height:whiteRect:) — rect = (x, y, w, h)."""
    m = np.zeros((height, width), dtype=np.uint8)
    if white_rect is not None:
        x, y, w, h = white_rect
        m[y:y + h, x:x + w] = 255
    return m


def solid_image(width, height, r, g, b):
    a = np.zeros((height, width, 3), dtype=np.uint8)
    a[:, :] = (r, g, b)
    return a
