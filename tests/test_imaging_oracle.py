"""oracle/imaging_ref.py pinned: the resampler bit-exactly on Pillow's outputs (tests/golden/imaging.npz, and live
against Pillow when it is importable), the bounding-box / crop-region / composite logic on the reference's own
XCTest cases (Tests/Flux2ChainsTests/Flux2InpaintCompositingTests.swift)."""
import numpy as np
import pytest

from oracle import imaging_ref as I
from tests.imaging_cases import CASES, GOLDEN, gray_mask, solid_image


@pytest.mark.parametrize("name", sorted(CASES))
def test_resample_matches_pillow_fixture(name):
    g = np.load(GOLDEN)
    (oh, ow), box = CASES[name]
    assert np.array_equal(I.resample_u8(g[name + "_in"], oh, ow, box), g[name + "_out"])


def test_resample_matches_pillow_live():
    PIL = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(7)
    for shape, (oh, ow), box in [((90, 70), (31, 17), None), ((64, 64), (200, 100), None), ((200, 300), (40, 60), (7.25, 9.5, 280.0, 190.0))]:
        a = rng.integers(0, 256, shape, dtype=np.uint8)
        ref = np.asarray(PIL.fromarray(a, "L").resize((ow, oh), PIL.BILINEAR, box=box))
        assert np.array_equal(I.resample_u8(a, oh, ow, box), ref)


# ---- the reference's XCTest cases -------------------------------------------------------------------------------
def test_mask_bounding_box_locates_white_region():
    bbox = I.mask_bounding_box(gray_mask(200, 100, (60, 20, 80, 40)), I.GRAYSCALE_WHITE_INPAINT, 200, 100)
    x, y, w, h = bbox
    assert abs(x - 60) <= 3 and abs(y - 20) <= 3 and abs(x + w - 140) <= 3 and abs(y + h - 60) <= 3


def test_mask_bounding_box_empty_mask_returns_none():
    assert I.mask_bounding_box(gray_mask(64, 64), I.GRAYSCALE_WHITE_INPAINT, 64, 64) is None


def test_mask_bounding_box_scales_to_image_space():
    x, y, w, h = I.mask_bounding_box(gray_mask(100, 50, (30, 10, 40, 20)), I.GRAYSCALE_WHITE_INPAINT, 200, 100)
    assert abs(x - 60) <= 6 and abs(x + w - 140) <= 6


def test_expand_crop_region_matches_image_aspect():
    x, y, w, h = I.expand_crop_region((500, 200, 100, 100), 32, 2000, 1000)
    assert abs(w / h - 2.0) <= 0.05
    assert x <= 500 - 32 and x + w >= 600 + 32 and y <= 200 - 32 and y + h >= 300 + 32


def test_expand_crop_region_clamps_to_image_bounds():
    x, y, w, h = I.expand_crop_region((5, 5, 50, 50), 64, 800, 600)
    assert x >= 0 and y >= 0 and x + w <= 800 and y + h <= 600 and w > 0 and h > 0


def test_expand_crop_region_huge_mask_caps_at_image_size():
    assert I.expand_crop_region((10, 10, 780, 580), 64, 800, 600)[2:] == (800, 600)


def test_composite_keeps_unmasked_pixels_and_pastes_masked():
    out = I.composite(solid_image(100, 100, 255, 0, 0), solid_image(50, 50, 0, 255, 0), (25, 25, 50, 50),
                      gray_mask(50, 50, (10, 10, 30, 30)))
    assert out.shape == (100, 100, 3)
    assert tuple(out[5, 5]) == (255, 0, 0)
    assert out[27, 27, 0] > 200 and out[27, 27, 1] < 50
    assert out[50, 50, 0] < 50 and out[50, 50, 1] > 200


def test_composite_soft_mask_blends():
    out = I.composite(solid_image(40, 40, 0, 0, 0), solid_image(40, 40, 255, 255, 255), (0, 0, 40, 40),
                      np.full((40, 40), 128, dtype=np.uint8))
    assert abs(int(out[20, 20, 0]) - 128) <= 6


def test_composite_rejects_out_of_bounds_crop():
    assert I.composite(solid_image(50, 50, 0, 0, 0), solid_image(20, 20, 255, 255, 255), (40, 40, 20, 20),
                       np.full((20, 20), 255, dtype=np.uint8)) is None


def test_alpha_convention_and_latent_mask():
    rgba = np.zeros((64, 64, 4), dtype=np.uint8)
    rgba[:, :, 3] = 255
    rgba[16:48, 16:48, 3] = 0           # erased square = inpaint
    w = I.render_inpaint_weights(rgba, 64, 64, I.ALPHA_TRANSPARENT_INPAINT)
    assert w[32, 32] == 1.0 and w[2, 2] == 0.0
    m = I.pack_mask_for_latent_blending(gray_mask(512, 512, (0, 0, 512, 256)), 512, 512)
    assert m.shape == (1, 1024, 1) and m[0, 0, 0] == 1.0 and m[0, -1, 0] == 0.0
    assert abs(float(m.sum()) - 512.0) < 1.0


# ---- outpainting chain: the reference's XCTest cases (Tests/Flux2ChainsTests/Flux2OutpaintingChainTests.swift) ------
def test_round_up_to_multiple_of_32():
    f = I.round_up_to_multiple_of_32
    assert [f(v) for v in (0, 32, 64, 480, 1, 31, 33, 63, 479)] == [0, 32, 64, 480, 32, 32, 64, 64, 480]


def test_outpaint_canvas_cases():
    src = np.full((32, 32, 3), 200, dtype=np.uint8)
    c = I.build_outpaint_canvas(np.full((32, 64, 3), 200, dtype=np.uint8), 96, 64, 16, 16, 42)
    assert c.shape == (64, 96, 3)
    c = I.build_outpaint_canvas(src, 96, 96, 32, 32, 42)
    assert abs(int(c[48, 48, 0]) - 200) <= 4 and c[48, 8, 0] != 200
    a, b = I.build_outpaint_canvas(src, 96, 96, 32, 32, 7), I.build_outpaint_canvas(src, 96, 96, 32, 32, 7)
    assert np.array_equal(a, b)
    a, b = I.build_outpaint_canvas(src, 96, 96, 32, 32, 1), I.build_outpaint_canvas(src, 96, 96, 32, 32, 2)
    assert any(a[y, x, 0] != b[y, x, 0] for x, y in [(2, 2), (2, 50), (90, 2), (90, 90)])
    strip = a[:32].astype(np.float64)                       # mean 128, sigma 32 grey noise
    assert abs(strip.mean() - 128) < 2 and abs(strip.std() - 32) < 2 and np.array_equal(a[..., 0], a[..., 1])


def smart_mask_cases(build):
    """Shared with the GPU test: `build(canvasW, canvasH, keepX, keepY, keepW, keepH, transition)` -> u8 [H,W]."""
    m = build(96, 32, 32, 0, 32, 32, 4)
    assert m[16, 4] == 255 and m[16, 48] == 0 and m[16, 92] == 255
    assert m[16, 32] > m[16, 33] > m[16, 34] > m[16, 35]                # monotonic ramp inside the keep
    m = build(96, 32, 0, 0, 64, 32, 4)
    assert m[16, 0] == 0 and m[16, 63] > 128 and m[16, 80] == 255       # ramp only on the padded side
    m = build(96, 32, 46, 0, 4, 32, 16)
    assert m[16, 45] == 255 and m[16, 50] == 255 and m[16, 46] > 200 and m[16, 47] < 200   # band capped at half the keep
    m = build(64, 64, 0, 32, 64, 32, 4)
    assert m[4, 32] == 255 and 0 < m[33, 32] < 255 and m[48, 32] == 0 and m[63, 32] == 0    # vertical ramps only


def test_smart_mask_cases():
    smart_mask_cases(I.build_smart_mask)
