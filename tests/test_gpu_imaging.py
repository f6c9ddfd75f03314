"""Pixel-space pre/post-processing on the device (csrc/imaging.hip) against the oracle (oracle/imaging_ref.py) —
bit-exact for the byte work — against Pillow's known answers (tests/golden/imaging.npz), on the reference's own
XCTest cases (Flux2InpaintCompositingTests.swift), and the inpainting chain from pixels end to end."""
import numpy as np
import pytest
import torch

from flux2mi import synthetic as syn
from oracle import imaging_ref as I
from tests.imaging_cases import CASES, GOLDEN, gray_mask, solid_image

pytestmark = pytest.mark.gpu
GRAY, ALPHA = 0, 1


def _np(t):
    return t.cpu().numpy()


@pytest.mark.parametrize("name", sorted(CASES))
def test_resample_matches_pillow_fixture(ctx, name):
    g = np.load(GOLDEN)
    (oh, ow), box = CASES[name]
    got = _np(ctx.image_resample_u8(g[name + "_in"], oh, ow, box))
    want = g[name + "_out"]
    assert np.array_equal(got.reshape(want.shape), want)


@pytest.mark.parametrize("shape,out,box", [
    ((1024, 1024), (64, 64), None),                           # mask -> latent grid of a 1024^2 image
    ((2048, 1536, 3), (1024, 768), None),                     # source image -> target size
    ((1024, 1024, 3), (700, 900), None),                      # generated crop -> crop size (non-integer ratio)
    ((777, 1333, 4), (512, 512), (100.0, 50.0, 1200.5, 700.25)),
    ((16, 16), (1024, 1024), None),                           # large upsample
])
def test_resample_matches_oracle_at_size(ctx, shape, out, box):
    rng = np.random.default_rng(sum(shape))
    a = rng.integers(0, 256, shape, dtype=np.uint8)
    got = _np(ctx.image_resample_u8(a, out[0], out[1], box))
    want = I.resample_u8(a, out[0], out[1], box)
    assert np.array_equal(got.reshape(want.shape), want)


def test_resample_rejects_bad_arguments(ctx, flux2mi):
    a = np.zeros((8, 8, 3), dtype=np.uint8)
    with pytest.raises(flux2mi.Flux2Error):
        ctx.image_resample_u8(a, 4, 4, (0, 0, 9, 8))            # box outside the image
    with pytest.raises(flux2mi.Flux2Error):
        ctx.image_resample_u8(a.astype(np.float32), 4, 4)


@pytest.mark.parametrize("conv", [GRAY, ALPHA])
def test_inpaint_weights_and_latent_mask_exact(ctx, conv):
    rng = np.random.default_rng(3 + conv)
    if conv == GRAY:
        mask = rng.integers(0, 256, (512, 768), dtype=np.uint8)
    else:
        mask = rng.integers(0, 256, (512, 768, 4), dtype=np.uint8)
    got = _np(ctx.mask_inpaint_weights(mask, 32, 48, conv))
    assert np.array_equal(got, I.render_inpaint_weights(mask, 48, 32, conv))
    box = (64.0, 32.0, 700.0, 500.0)
    got = _np(ctx.mask_inpaint_weights(mask, 20, 30, conv, box))
    assert np.array_equal(got, I.render_inpaint_weights(mask, 30, 20, conv, box))


def test_pack_mask_for_latent_blending(flux2mi, ctx):
    from flux2mi.pipeline import Flux2Pipeline
    pipe = Flux2Pipeline(model=syn.DIT_PRESETS["tiny"], vae_variant=syn.VAE_PRESETS["tiny"])
    mask = gray_mask(512, 512, (0, 0, 512, 256))
    m = pipe.pack_mask_for_latent_blending(mask, 512, 512)
    assert tuple(m.shape) == (1, 1024, 1)
    assert np.array_equal(_np(m), I.pack_mask_for_latent_blending(mask, 512, 512))
    # a float mask in [0,1] is quantised to 8 bits first
    mf = pipe.pack_mask_for_latent_blending(torch.from_numpy(mask).float() / 255.0, 512, 512)
    assert torch.equal(m, mf)
    with pytest.raises(flux2mi.Flux2Error):
        pipe.pack_mask_for_latent_blending(mask, 512, 512, ALPHA)    # alpha convention needs RGBA


# ---- the reference's XCTest cases through the product API ---------------------------------------------------------
def test_mask_bounding_box_cases(ctx):
    from flux2mi.pipeline import Flux2InpaintCompositing
    comp = Flux2InpaintCompositing(ctx)
    m = gray_mask(200, 100, (60, 20, 80, 40))
    x, y, w, h = comp.maskBoundingBox(m, GRAY, 200, 100)
    assert abs(x - 60) <= 3 and abs(y - 20) <= 3 and abs(x + w - 140) <= 3 and abs(y + h - 60) <= 3
    assert (x, y, w, h) == I.mask_bounding_box(m, GRAY, 200, 100)
    assert comp.maskBoundingBox(gray_mask(64, 64), GRAY, 64, 64) is None
    m = gray_mask(100, 50, (30, 10, 40, 20))
    x, y, w, h = comp.maskBoundingBox(m, GRAY, 200, 100)
    assert abs(x - 60) <= 6 and abs(x + w - 140) <= 6
    # larger than the 512-px scan cap, soft edge, alpha convention
    rgba = np.full((1500, 2000, 4), 255, dtype=np.uint8)
    rgba[400:900, 1000:1800, 3] = 0
    rgba[390:400, 1000:1800, 3] = 200
    got = comp.maskBoundingBox(rgba, ALPHA, 2000, 1500)
    assert got == I.mask_bounding_box(rgba, ALPHA, 2000, 1500)
    assert got[0] <= 1000 and got[0] + got[2] >= 1800 and got[1] <= 400 and got[1] + got[3] >= 900


def test_expand_crop_region_cases(ctx):
    from flux2mi.pipeline import Flux2InpaintCompositing as C_
    x, y, w, h = C_.expandCropRegion((500, 200, 100, 100), 32, 2000, 1000)
    assert abs(w / h - 2.0) <= 0.05 and x <= 468 and x + w >= 632 and y <= 168 and y + h >= 332
    x, y, w, h = C_.expandCropRegion((5, 5, 50, 50), 64, 800, 600)
    assert x >= 0 and y >= 0 and x + w <= 800 and y + h <= 600 and w > 0 and h > 0
    assert C_.expandCropRegion((10, 10, 780, 580), 64, 800, 600)[2:] == (800, 600)
    rng = np.random.default_rng(0)
    for _ in range(300):                                       # host integer logic == oracle on random boxes
        iw, ih = int(rng.integers(64, 4000)), int(rng.integers(64, 4000))
        bw, bh = int(rng.integers(1, iw + 1)), int(rng.integers(1, ih + 1))
        bx, by = int(rng.integers(0, iw - bw + 1)), int(rng.integers(0, ih - bh + 1))
        pad = int(rng.integers(0, 200))
        assert C_.expandCropRegion((bx, by, bw, bh), pad, iw, ih) == I.expand_crop_region((bx, by, bw, bh), pad, iw, ih)


def test_composite_cases(ctx):
    from flux2mi.pipeline import Flux2InpaintCompositing
    comp = Flux2InpaintCompositing(ctx)
    out = _np(comp.composite(solid_image(100, 100, 255, 0, 0), solid_image(50, 50, 0, 255, 0), (25, 25, 50, 50),
                             gray_mask(50, 50, (10, 10, 30, 30)), GRAY))
    assert out.shape == (100, 100, 3) and tuple(out[5, 5]) == (255, 0, 0)
    assert out[27, 27, 0] > 200 and out[27, 27, 1] < 50 and out[50, 50, 0] < 50 and out[50, 50, 1] > 200
    out = _np(comp.composite(solid_image(40, 40, 0, 0, 0), solid_image(40, 40, 255, 255, 255), (0, 0, 40, 40),
                             np.full((40, 40), 128, dtype=np.uint8), GRAY))
    assert abs(int(out[20, 20, 0]) - 128) <= 6
    assert comp.composite(solid_image(50, 50, 0, 0, 0), solid_image(20, 20, 255, 255, 255), (40, 40, 20, 20),
                          np.full((20, 20), 255, dtype=np.uint8), GRAY) is None


def test_composite_exact_random(ctx):
    from flux2mi.pipeline import Flux2InpaintCompositing
    comp = Flux2InpaintCompositing(ctx)
    rng = np.random.default_rng(11)
    base = rng.integers(0, 256, (1536, 2048, 3), dtype=np.uint8)
    gen = rng.integers(0, 256, (1024, 1024, 3), dtype=np.uint8)     # resampled to the 900x700 crop
    mask = rng.integers(0, 256, (450, 350), dtype=np.uint8)          # soft mask at half the crop resolution
    mask[:100] = 0
    mask[-100:] = 255
    rect = (333, 77, 700, 900)
    got = _np(comp.composite(base, gen, rect, mask, GRAY))
    want = I.composite(base, gen, rect, mask, GRAY)
    assert np.array_equal(got, want)
    assert np.array_equal(got[:77], base[:77]) and np.array_equal(got[:, :333], base[:, :333])


def test_preprocess_image_for_vae_exact(ctx):
    rng = np.random.default_rng(5)
    for c in (3, 4):
        img = rng.integers(0, 256, (96, 160, c), dtype=np.uint8)
        assert np.array_equal(_np(ctx.preprocess_image_for_vae(img)), I.preprocess_image_for_vae(img))


# ---- the chain from pixels ----------------------------------------------------------------------------------------
def test_inpainting_chain_from_pixels(flux2mi, ctx):
    from flux2mi.pipeline import Flux2MaskedInpaintingChain, Flux2Pipeline
    cfg, vcfg = syn.DIT_PRESETS["tiny"], syn.VAE_PRESETS["tiny"]
    pipe = Flux2Pipeline(model=cfg, vae_variant=vcfg).load_models(syn.make_dit_weights(cfg), syn.make_vae_weights(vcfg, encoder=True))
    txt = syn.make_inputs(cfg, 128, 128, s_txt=16)["txt"]
    rng = np.random.default_rng(2)
    image = rng.integers(0, 256, (200, 300, 3), dtype=np.uint8)
    mask = gray_mask(300, 200, (120, 60, 64, 48))
    # full canvas + composite on the original: 300x200 -> 288x192 generation, pasted back at 300x200
    res = Flux2MaskedInpaintingChain.run_on_image(pipe, image, mask, txt, steps=3, seed=3, compositeOnOriginal=True)
    assert res.image.shape == (200, 300, 3)
    keep = np.ones((200, 300), dtype=bool)
    keep[55:115, 112:192] = False
    assert np.array_equal(res.image[keep], image[keep])                  # untouched outside the (soft-edged) mask
    assert not np.array_equal(res.image[70:100, 130:170], image[70:100, 130:170])
    again = Flux2MaskedInpaintingChain.run_on_image(pipe, image, mask, txt, steps=3, seed=3, compositeOnOriginal=True)
    assert np.array_equal(res.image, again.image)                        # seed -> blend noise + initial noise, reproducible
    # crop-and-stitch: only the region around the mask is generated
    res2 = Flux2MaskedInpaintingChain.run_on_image(pipe, image, mask, txt, steps=3, seed=3, maskCropPadding=16)
    assert res2.image.shape == (200, 300, 3)
    assert np.array_equal(res2.image[keep], image[keep])
    assert not np.array_equal(res2.image[70:100, 130:170], image[70:100, 130:170])
    # without compositing the raw generated canvas comes back at the resolved size
    res3 = Flux2MaskedInpaintingChain.run_on_image(pipe, image, mask, txt, steps=3, seed=3)
    assert res3.image.shape == (192, 288, 3)


# ---- outpainting chain --------------------------------------------------------------------------------------------
def test_outpaint_canvas_and_mask_exact(ctx):
    from tests.test_imaging_oracle import smart_mask_cases
    rng = np.random.default_rng(9)
    src = rng.integers(0, 256, (160, 224, 3), dtype=np.uint8)
    for seed in (0, 42, (1 << 64) - 5):
        got = _np(ctx.outpaint_canvas(src, 160 + 64, 224 + 96, 32, 64, seed))
        want = I.build_outpaint_canvas(src, 224 + 96, 160 + 64, 32, 64, seed)
        # f64 log / cos of the device and of numpy may differ in the last place: a rounding boundary can flip one grey level
        diff = np.abs(got.astype(int) - want.astype(int))
        assert diff.max() <= 1 and (diff > 0).mean() < 1e-4
        assert np.array_equal(got[64:64 + 160, 32:32 + 224], src)
    smart_mask_cases(lambda cw, ch, kx, ky, kw, kh, t: _np(ctx.outpaint_smart_mask(ch, cw, kx, ky, kw, kh, t)))
    for args in [(1024, 768, 128, 0, 512, 768, 32), (640, 640, 64, 96, 512, 448, 32), (96, 96, 32, 32, 32, 32, 100), (64, 64, 0, 0, 64, 64, 8)]:
        cw, ch, kx, ky, kw, kh, t = args
        assert np.array_equal(_np(ctx.outpaint_smart_mask(ch, cw, kx, ky, kw, kh, t)), I.build_smart_mask(*args))


def test_outpainting_chain(flux2mi, ctx):
    from flux2mi.pipeline import Flux2OutpaintingChain, Flux2Pipeline
    cfg, vcfg = syn.DIT_PRESETS["tiny"], syn.VAE_PRESETS["tiny"]
    pipe = Flux2Pipeline(model=cfg, vae_variant=vcfg).load_models(syn.make_dit_weights(cfg), syn.make_vae_weights(vcfg, encoder=True))
    txt = syn.make_inputs(cfg, 128, 128, s_txt=16)["txt"]
    image = np.random.default_rng(4).integers(0, 256, (128, 96, 3), dtype=np.uint8)
    assert Flux2OutpaintingChain.roundUpToMultipleOf32(33) == 64
    with pytest.raises(flux2mi.Flux2Error):
        Flux2OutpaintingChain(pipe, image, left=-16).run(txt)                 # testRunRejectsNegativePadding
    with pytest.raises(flux2mi.Flux2Error):
        Flux2OutpaintingChain(pipe, image).run(txt)                           # testRunRejectsAllZeroPadding
    with pytest.raises(flux2mi.Flux2Error):
        Flux2OutpaintingChain(pipe, image[:100], left=32).run(txt)            # canvas not a multiple of 32
    res = Flux2OutpaintingChain(pipe, image, left=40, right=64, steps=3, seed=5, transitionPixels=8).run(txt)
    assert res.image.shape == (128, 96 + 64 + 64, 3)                          # 40 -> 64
    again = Flux2OutpaintingChain(pipe, image, left=40, right=64, steps=3, seed=5, transitionPixels=8).run(txt)
    assert np.array_equal(res.image, again.image)
