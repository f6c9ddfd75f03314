"""CPU restatement (numpy) of the chains' pixel-space pre/post-processing.  TEST INFRASTRUCTURE ONLY: imported
by tests/ and nothing else; the product path is `flux2mi_image_* / flux2mi_mask_* / flux2mi_composite_u8`
(csrc/imaging.hip).

What it follows in the reference:
  * mask -> inpaint weights / packed latent mask: `Flux2Pipeline.packMaskForLatentBlending`
    (Flux2Core/Pipeline/Flux2Pipeline+ChainHelpers.swift:148-236), `renderInpaintWeights`
    (Flux2Chains/Flux2InpaintCompositing.swift:196-227)
  * mask bounding box, crop region, pixel composite: Flux2InpaintCompositing.swift:23-65, 69-108, 112-162
  * u8 image -> VAE input: Flux2Pipeline.swift:2359-2418 (x / 127.5 - 1, planar)

The reference RESAMPLES with CoreGraphics (`CGContext.draw`, interpolationQuality .high), a closed filter that
cannot be restated.  SURVEY §8f row 2 replaces it with a documented open filter: the antialiased triangle
("bilinear") filter in Pillow's 8-bit fixed-point form (22-bit coefficients, horizontal pass then vertical pass,
u8 intermediate) — `resample_u8` below.  PARITY STATUS: `resample_u8` is pinned bit-exactly against Pillow 12.2.0
`Image.resize(..., BILINEAR, box=...)` (tests/test_imaging_oracle.py, fixtures in tests/golden/imaging.npz); the
integer/float logic of bounding box, crop region and composite is pinned on the reference's own XCTest cases
(Tests/Flux2ChainsTests/Flux2InpaintCompositingTests.swift), which hold for any reasonable resampling filter;
agreement with CoreGraphics' pixels is NOT claimed.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _coeffs(in_size, in0, in1, out_size):
    """Per output index: first tap, tap count, fixed-point coefficients (Pillow precompute_coeffs +
    normalize_coeffs_8bpc with the bilinear filter, support 1)."""
    scale = (in1 - in0) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = in0 + (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        x = np.arange(xmax, dtype=np.float64)
        w = 1.0 - np.abs((x + xmin - center + 0.5) * ss)
        w = np.where(w > 0.0, w, 0.0)
        ww = w.sum()
        if ww != 0.0:
            w = w / ww
        kk[xx, :xmax] = np.where(w < 0, (-0.5 + w * (1 << PRECISION_BITS)), (0.5 + w * (1 << PRECISION_BITS))).astype(np.int64)
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img, bounds, kk, axis):
    """One separable pass over `axis` (0 = rows / vertical, 1 = columns / horizontal); img u8 [H,W,C]."""
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.zeros((bounds.shape[0],) + src.shape[1:], dtype=np.uint8)
    for i, (lo, n) in enumerate(bounds):
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for t in range(n):
            acc += src[lo + t] * int(kk[i, t])
        out[i] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resample_u8(img, out_h, out_w, box=None):
    """u8 [H,W] or [H,W,C] -> u8 [out_h,out_w(,C)]; `box` = (x0, y0, x1, y1) source rectangle in pixels (floats
    allowed), default the whole image.  Channels are filtered independently (no alpha premultiplication)."""
    a = np.asarray(img, dtype=np.uint8)
    squeeze = a.ndim == 2
    if squeeze:
        a = a[:, :, None]
    H, W, _ = a.shape
    x0, y0, x1, y1 = (0.0, 0.0, float(W), float(H)) if box is None else [float(v) for v in box]
    bh, kh = _coeffs(W, x0, x1, out_w)
    bv, kv = _coeffs(H, y0, y1, out_h)
    tmp = _pass(a, bh, kh, axis=1)       # horizontal first, u8 intermediate
    out = _pass(tmp, bv, kv, axis=0)
    return out[:, :, 0] if squeeze else out


# ---- mask conventions (Flux2MaskConvention): 0 = grayscaleWhiteInpaint, 1 = alphaTransparentInpaint
GRAYSCALE_WHITE_INPAINT, ALPHA_TRANSPARENT_INPAINT = 0, 1


def render_inpaint_weights(mask, width, height, convention=GRAYSCALE_WHITE_INPAINT, box=None):
    """Flux2InpaintCompositing.renderInpaintWeights (:196-227): mask rasterised to width x height in 8 bits;
    grayscale -> v/255 (1 = inpaint); alpha -> 1 - a/255.  mask: u8 [H,W] (gray) or [H,W,4] (RGBA, straight)."""
    m = np.asarray(mask, dtype=np.uint8)
    if convention == GRAYSCALE_WHITE_INPAINT:
        g = m if m.ndim == 2 else m[:, :, 0]
        r = resample_u8(g, height, width, box)
        return (r.astype(np.float32) / np.float32(255.0)).astype(np.float32)
    r = resample_u8(m[:, :, 3], height, width, box)
    return (np.float32(1.0) - r.astype(np.float32) / np.float32(255.0)).astype(np.float32)


def pack_mask_for_latent_blending(mask, target_height, target_width, convention=GRAYSCALE_WHITE_INPAINT, box=None):
    """+ChainHelpers.swift:148-236 -> f32 [1, (H/16)(W/16), 1]."""
    w = render_inpaint_weights(mask, target_width // 16, target_height // 16, convention, box)
    return w.reshape(1, -1, 1)


def mask_bounding_box(mask, convention, image_width, image_height, threshold=np.float32(8.0 / 255.0)):
    """Flux2InpaintCompositing.maskBoundingBox (:23-65) -> (x, y, w, h) in image space, or None."""
    mh, mw = np.asarray(mask).shape[:2]
    scale = min(1.0, 512.0 / float(max(mw, mh)))
    scan_w, scan_h = max(1, int(mw * scale)), max(1, int(mh * scale))
    weights = render_inpaint_weights(mask, scan_w, scan_h, convention)
    ys, xs = np.nonzero(weights > np.float32(threshold))
    if len(xs) == 0:
        return None
    min_x, max_x, min_y, max_y = int(xs.min()), int(xs.max()), int(ys.min()), int(ys.max())
    sx, sy = image_width / scan_w, image_height / scan_h
    x0 = max(0, int((min_x - 1) * sx))
    y0 = max(0, int((min_y - 1) * sy))
    x1 = min(image_width, int((max_x + 2) * sx))
    y1 = min(image_height, int((max_y + 2) * sy))
    if not (x1 > x0 and y1 > y0):
        return None
    return (x0, y0, x1 - x0, y1 - y0)


def expand_crop_region(bbox, padding, image_width, image_height):
    """Flux2InpaintCompositing.expandCropRegion (:69-108), diffusers' get_crop_region."""
    bx, by, bw, bh = [int(v) for v in bbox]
    x0, y0, x1, y1 = bx - padding, by - padding, bx + bw + padding, by + bh + padding
    target = image_width / image_height
    w, h = x1 - x0, y1 - y0
    if (w / h) > target:
        grow = int(math.ceil(w / target)) - h
        y0 -= _sdiv(grow, 2)
        y1 += grow - _sdiv(grow, 2)
    else:
        grow = int(math.ceil(h * target)) - w
        x0 -= _sdiv(grow, 2)
        x1 += grow - _sdiv(grow, 2)
    if x0 < 0:
        x1 -= x0
        x0 = 0
    if y0 < 0:
        y1 -= y0
        y0 = 0
    if x1 > image_width:
        x0 -= x1 - image_width
        x1 = image_width
        x0 = max(0, x0)
    if y1 > image_height:
        y0 -= y1 - image_height
        y1 = image_height
        y0 = max(0, y0)
    return (x0, y0, x1 - x0, y1 - y0)


def _sdiv(a, b):
    """Swift integer division truncates toward zero."""
    return int(a / b) if a >= 0 else -int((-a) / b)


def composite(original, generated, crop_rect, mask_crop, convention=GRAYSCALE_WHITE_INPAINT):
    """Flux2InpaintCompositing.composite (:112-162): `generated` is resampled to the crop size, the mask crop is
    rendered to per-pixel weights m; m <= 0 keeps the original, m >= 1 pastes, otherwise
    u8(trunc(clamp(orig*(1-m) + gen*m)))  in f32.  u8 [H,W,3]; returns None when the crop leaves the image."""
    base = np.array(original, dtype=np.uint8, copy=True)
    H, W = base.shape[:2]
    cx, cy, cw, ch = [int(v) for v in crop_rect]
    if not (cw > 0 and ch > 0 and cx >= 0 and cy >= 0 and cx + cw <= W and cy + ch <= H):
        return None
    gen = np.asarray(generated, dtype=np.uint8)
    if gen.shape[:2] != (ch, cw):
        gen = resample_u8(gen, ch, cw)
    m = render_inpaint_weights(mask_crop, cw, ch, convention)[:, :, None].astype(np.float32)
    b = base[cy:cy + ch, cx:cx + cw, :3].astype(np.float32)
    g = gen[:, :, :3].astype(np.float32)
    inv = (np.float32(1.0) - m).astype(np.float32)
    mix = ((b * inv).astype(np.float32) + (g * m).astype(np.float32)).astype(np.float32)
    mix = np.clip(mix, 0.0, 255.0).astype(np.uint8)   # truncating cast
    out = np.where(m <= 0, base[cy:cy + ch, cx:cx + cw, :3], np.where(m >= 1, gen[:, :, :3], mix))
    base[cy:cy + ch, cx:cx + cw, :3] = out
    return base


def preprocess_image_for_vae(image_u8):
    """Flux2Pipeline.swift:2359-2418: u8 [H,W,3|4] -> f32 [1,3,H,W] = x / 127.5 - 1."""
    a = np.asarray(image_u8, dtype=np.uint8)[:, :, :3].astype(np.float32)
    return ((a / np.float32(127.5)).astype(np.float32) - np.float32(1.0)).transpose(2, 0, 1)[None]


# ---- outpainting chain (Flux2Chains/Flux2OutpaintingChain.swift) ---------------------------------------------------
def round_up_to_multiple_of_32(x):
    """:324-326."""
    return 0 if x == 0 else ((x + 31) // 32) * 32


def build_outpaint_canvas(source, canvas_w, canvas_h, offset_x, offset_y, noise_seed):
    """buildOutpaintCanvas (:331-395): SplitMix64 -> Box-Muller (f64), grey noise mean 128 sigma 32, two draws per
    pixel in row-major order, the source pasted at (offset_x, offset_y) from the top-left.  u8 [canvas_h,canvas_w,3]."""
    n = canvas_w * canvas_h
    G = np.uint64(0x9E3779B97F4A7C15)
    with np.errstate(over="ignore"):
        s0 = np.uint64(int(noise_seed) & 0xFFFFFFFFFFFFFFFF) + G
        j = np.arange(1, 2 * n + 1, dtype=np.uint64)
        z = s0 + j * G
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    u = ((z >> np.uint64(11)).astype(np.float64) + 1.0) / float(1 << 53)
    u1, u2 = u[0::2], u[1::2]
    g = 128.0 + (np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)) * 32.0
    v = np.clip(np.sign(g) * np.floor(np.abs(g) + 0.5), 0, 255).astype(np.uint8)   # .rounded(): half away from zero
    canvas = np.repeat(v.reshape(canvas_h, canvas_w, 1), 3, axis=2)
    src = np.asarray(source, dtype=np.uint8)
    canvas[offset_y:offset_y + src.shape[0], offset_x:offset_x + src.shape[1]] = src[:, :, :3]
    return canvas


def build_smart_mask(canvas_w, canvas_h, keep_x, keep_y, keep_w, keep_h, transition_pixels):
    """buildSmartMask (:397-474), written as the reference writes it (fill, then max-in the four ramps)."""
    m = np.full((canvas_h, canvas_w), 255, dtype=np.uint8)
    m[keep_y:keep_y + keep_h, keep_x:keep_x + keep_w] = 0
    band_w = max(1, min(transition_pixels, keep_w // 2))
    band_h = max(1, min(transition_pixels, keep_h // 2))
    ys, xs = slice(keep_y, keep_y + keep_h), slice(keep_x, keep_x + keep_w)
    if keep_x > 0:
        for x in range(band_w):
            v = (255 * (band_w - x) + band_w // 2) // band_w
            m[ys, keep_x + x] = np.maximum(m[ys, keep_x + x], v)
    if keep_x + keep_w < canvas_w:
        for x in range(band_w):
            v = (255 * (band_w - x) + band_w // 2) // band_w
            col = keep_x + keep_w - 1 - x
            m[ys, col] = np.maximum(m[ys, col], v)
    if keep_y > 0:
        for y in range(band_h):
            v = (255 * (band_h - y) + band_h // 2) // band_h
            m[keep_y + y, xs] = np.maximum(m[keep_y + y, xs], v)
    if keep_y + keep_h < canvas_h:
        for y in range(band_h):
            v = (255 * (band_h - y) + band_h // 2) // band_h
            row = keep_y + keep_h - 1 - y
            m[row, xs] = np.maximum(m[row, xs], v)
    return m
