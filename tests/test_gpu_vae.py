"""-m gpu: VAE decoder parity (HIP, NHWC bf16 activations, f32 accumulate / GroupNorm
statistics) vs the fp32 oracle.

Tolerance (stated): decoded f32 image rel-L2 <= 3e-2; u8 image mean |diff| <= 2 levels and
<= 1 % of the pixels off by more than 8 levels (bf16 activations through ~30 layers)."""
import pytest
import torch

from oracle import flux2_ref as R
from tests.util import rel_l2

pytestmark = pytest.mark.gpu


def _decode_both(flux2mi, ctx, vcfg, h, w, seed=4321):
    from flux2mi import synthetic as syn
    from flux2mi.api import VAE

    VP = syn.make_vae_weights(vcfg, seed=seed)
    g = torch.Generator().manual_seed(11)
    z = torch.randn((1, 32, h, w), generator=g)
    vae = VAE(ctx, vcfg).load(VP).finalize()
    of, ou = vae.decode(z.cuda())
    ref = R.vae_decode(VP, vcfg, z)
    vae.close()
    return of, ou, ref


def _check(of, ou, ref):
    assert torch.isfinite(of).all()
    assert rel_l2(of, ref) < 3e-2
    ref_u8 = R.postprocess_vae_output(ref)
    d = (ou.cpu().int() - ref_u8.int()).abs().float()
    assert d.mean() < 2.0
    assert (d > 8).float().mean() < 0.01
    # the u8 output is exactly the truncating conversion of the library's own f32 output
    assert torch.equal(ou.cpu(), R.postprocess_vae_output(of.cpu()))


def test_vae_tiny(flux2mi, ctx):
    from flux2mi import synthetic as syn
    _check(*_decode_both(flux2mi, ctx, syn.VAE_PRESETS["tiny"], 8, 8))


def test_vae_tiny_non_square(flux2mi, ctx):
    """SmallDecoderVAETests.swift:84-194 checks non-square shapes; here values too."""
    from flux2mi import synthetic as syn
    of, ou, ref = _decode_both(flux2mi, ctx, syn.VAE_PRESETS["tiny"], 6, 10)
    assert tuple(of.shape) == (1, 3, 48, 80) and tuple(ou.shape) == (48, 80, 3)
    _check(of, ou, ref)


def test_vae_small_decoder_channels(flux2mi, ctx):
    """Small-decoder widths (96,192,384,384: 3 channels per GroupNorm group) at 16x16 latents."""
    from flux2mi import synthetic as syn
    _check(*_decode_both(flux2mi, ctx, syn.VAE_PRESETS["small"], 16, 16))


def test_vae_standard_decoder_channels(flux2mi, ctx):
    from flux2mi import synthetic as syn
    _check(*_decode_both(flux2mi, ctx, syn.VAE_PRESETS["standard"], 16, 16))


def test_vae_standard_decoder_ragged_upsample_phases(flux2mi, ctx):
    """The upsampling convolutions of the standard decoder run as four 2x2 convolutions on the source image, one per output phase
    (vae_kernels.hip, conv2d_g128_kernel<64, true>; combined weights from launch_upsample_phase_weights): a 6x10 latent makes every
    source-pixel count (60, 240, 960) a partial 128-pixel tile and every phase touch all four borders — against the oracle's
    upsample-then-3x3 restatement (VAEDecoder.swift:91-121)."""
    from flux2mi import synthetic as syn
    of, ou, ref = _decode_both(flux2mi, ctx, syn.VAE_PRESETS["standard"], 6, 10)
    assert tuple(of.shape) == (1, 3, 48, 80)
    _check(of, ou, ref)


@pytest.mark.parametrize("variant,h,w", [("standard", 65, 67), ("small", 66, 70)])
def test_vae_decoder_gemm_convolutions_ragged(flux2mi, ctx, variant, h, w):
    """The 3x3 convolutions as implicit GEMMs over zero-haloed images (gemm256s.hip, GemmProblem::conv_*; vae.cpp conv_gemm /
    upsample_gemm; taken from 4096 pixels on) at sizes where nothing is aligned: an odd, non-square latent (65 x 67: padded pitch 69)
    — every level has a partial last tile, halo columns inside tiles, all four phases of the upsampling GEMMs touching all borders,
    GroupNorm statistics from partial tiles' epilogues, the 512 x 128 tile at the last level, the fused conv_out tail with ragged 8 x 32
    tiles — against the oracle's plain restatement (VAEDecoder.swift:91-121, ResnetBlock.swift:122-254)."""
    from flux2mi import synthetic as syn
    of, ou, ref = _decode_both(flux2mi, ctx, syn.VAE_PRESETS[variant], h, w)
    assert tuple(of.shape) == (1, 3, 8 * h, 8 * w)
    _check(of, ou, ref)


# ---- encoder (VAE encoder + encodeImageToPackedSequence) ---------------------
def _image(h, w, seed=3):
    g = torch.Generator().manual_seed(seed)
    # smooth-ish image in [-1, 1]
    x = torch.randn((1, 3, h // 8, w // 8), generator=g)
    x = torch.nn.functional.interpolate(x, size=(h, w), mode="bilinear", align_corners=False)
    return (x / x.abs().max()).contiguous()


@pytest.mark.parametrize("variant,h,w", [("tiny", 64, 64), ("tiny", 96, 160), ("small", 64, 64)])
def test_encode_vs_oracle(flux2mi, ctx, variant, h, w):
    """Posterior mean of AutoencoderKLFlux2.encode: rel-L2 <= 3e-2 vs the f32 oracle (bf16 NHWC
    activations, stride-2 convs with bottom/right padding, 3->32 channel padding of conv_in)."""
    from flux2mi import synthetic as syn
    from flux2mi.api import VAE
    vcfg = syn.VAE_PRESETS[variant]
    P = syn.make_vae_weights(vcfg, seed=21)
    vae = VAE(ctx, vcfg).load(P).finalize()
    assert vae.has_encoder
    img = _image(h, w)
    z = vae.encode(img.cuda()).cpu()
    ref = R.vae_encode(P, vcfg, img)
    assert z.shape == ref.shape == (1, 32, h // 8, w // 8)
    assert rel_l2(z, ref) < 3e-2
    packed = vae.encode_to_packed(img.cuda()).cpu()
    assert rel_l2(packed, R.encode_image_to_packed_sequence(P, vcfg, img)) < 3e-2
    vae.close()


def test_encode_without_encoder_weights_fails_loudly(flux2mi, ctx):
    from flux2mi import Flux2Error, synthetic as syn
    from flux2mi.api import VAE
    vcfg = syn.VAE_PRESETS["tiny"]
    vae = VAE(ctx, vcfg).load(syn.make_vae_weights(vcfg, encoder=False)).finalize()   # decode-only checkpoint
    assert not vae.has_encoder
    with pytest.raises(Flux2Error):
        vae.encode(torch.zeros((1, 3, 64, 64), device="cuda"))
    of, _ = vae.decode(torch.zeros((1, 32, 8, 8), device="cuda"))
    assert torch.isfinite(of).all()
    vae.close()


def test_encode_decode_roundtrip_shapes_via_pipeline(flux2mi, ctx):
    """encode_image_to_packed_sequence feeds the RePaint chain: u8 image in, [1,S,128] out."""
    from flux2mi import synthetic as syn
    from flux2mi.pipeline import Flux2Pipeline
    cfg, vcfg = syn.DIT_PRESETS["tiny"], syn.VAE_PRESETS["tiny"]
    pipe = Flux2Pipeline(model=cfg, vae_variant=vcfg).load_models(syn.make_dit_weights(cfg), syn.make_vae_weights(vcfg))
    img = (torch.rand((128, 128, 3), generator=torch.Generator().manual_seed(1)) * 255).to(torch.uint8)
    packed = pipe.encode_image_to_packed_sequence(img, 128, 128)
    assert packed.shape == (1, 64, 128) and torch.isfinite(packed).all()
    ref = R.encode_image_to_packed_sequence(syn.make_vae_weights(vcfg), vcfg, R.preprocess_image_for_vae(img))
    assert rel_l2(packed, ref) < 3e-2


def test_decode_is_bit_reproducible(flux2mi, ctx):
    """Same weights, same latent -> identical bytes, across instances and repeated calls (GroupNorm
    statistics are combined in a fixed order; no floating-point atomics anywhere in the decoder)."""
    from flux2mi import synthetic as syn
    from flux2mi.api import VAE
    vcfg = syn.VAE_PRESETS["small"]
    VP = syn.make_vae_weights(vcfg)
    z = torch.randn((1, 32, 16, 24), generator=torch.Generator().manual_seed(8)).cuda()
    a = VAE(ctx, vcfg).load(VP).finalize()
    b = VAE(ctx, vcfg).load(VP).finalize()
    ref_f, ref_u = a.decode(z)
    ref_f, ref_u = ref_f.clone(), ref_u.clone()
    for _ in range(5):
        f1, u1 = a.decode(z)
        f2, u2 = b.decode(z)
        assert torch.equal(f1, ref_f) and torch.equal(u1, ref_u)
        assert torch.equal(f2, ref_f) and torch.equal(u2, ref_u)
    a.close(); b.close()
