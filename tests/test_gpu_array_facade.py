"""-m gpu: the MLXArray façade of the boundary (include/flux2mi.h).

The reference's hook type is `(Flux2StepContext, MLXArray) -> MLXArray` and the UNCHANGED RePaint chain builds its blend
from MLX operators (Flux2Chains/Flux2MaskedInpaintingChain.swift:380-403).  These tests run that closure operator by
operator through flux2mi_array_* (one C-ABI call each, via api.MLXArray) and require bit-identity with the fused
flux2mi_repaint_blend and with a torch f32 evaluation of the same expression tree (separately rounded ops)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _inputs(S=96, seed=3):
    g = torch.Generator().manual_seed(seed)
    img = torch.randn((1, S, 128), generator=g)
    noise = torch.randn((1, S, 128), generator=g)
    x = torch.randn((1, S, 128), generator=g)
    mask = torch.rand((1, S, 1), generator=g)
    mask[0, :10] = 0.0
    mask[0, 10:20] = 1.0
    return img.cuda(), noise.cuda(), x.cuda(), mask.cuda()


@pytest.mark.parametrize("sigma_next", [0.0, 0.3125, 0.7371, 1.0])
def test_repaint_closure_op_by_op_equals_fused_kernel(flux2mi, ctx, sigma_next):
    from flux2mi._lib import check, lib
    from flux2mi.api import MLXArray
    img, noise, x, mask = _inputs()
    imageLatentsCaptured, blendNoise = MLXArray.wrap(ctx, img), MLXArray.wrap(ctx, noise)
    maskLatentsCaptured, latents = MLXArray.wrap(ctx, mask), MLXArray.wrap(ctx, x)
    # Flux2MaskedInpaintingChain.swift:399-403, verbatim
    sigmaNext = MLXArray(ctx, sigma_next)
    originalNoised = (1 - sigmaNext) * imageLatentsCaptured + sigmaNext * blendNoise
    out = (1 - maskLatentsCaptured) * originalNoised + maskLatentsCaptured * latents
    assert out.shape == [1, 96, 128] and sigmaNext.shape == [] and (1 - maskLatentsCaptured).shape == [1, 96, 1]
    got = out.eval().torch()
    fused = x.clone()
    ctx._enter()
    check(lib().flux2mi_repaint_blend(ctx.handle, fused.data_ptr(), img.data_ptr(), noise.data_ptr(), mask.data_ptr(), sigma_next, 96, 128))
    ctx._exit()
    assert torch.equal(got, fused)
    s = torch.tensor(sigma_next, dtype=torch.float32, device="cuda")
    ref = (1 - mask) * ((1 - s) * img + s * noise) + mask * x      # torch: one rounding per op as well
    assert torch.equal(got, ref)
    if sigma_next == 0.0:
        assert torch.equal(got[0, :10], img[0, :10])               # outside the mask the clean source comes back exactly


def test_broadcast_forms_and_errors(flux2mi, ctx):
    """[] x [S,C], [S,1] x [S,C], [1,S,1] x [1,S,C], [C] x [S,C], non-multiple-of-4 last dims (scalar path), scalars on
    either side, and a shape error for shapes that do not broadcast."""
    from flux2mi import Flux2Error
    from flux2mi.api import MLXArray
    g = torch.Generator().manual_seed(1)
    for shp_a, shp_b in [((), (7, 12)), ((7, 1), (7, 12)), ((1, 7, 1), (1, 7, 12)), ((12,), (7, 12)), ((5, 3), (5, 3)),
                         ((2, 1, 5), (4, 1)), ((2, 3, 4, 8), (3, 1, 8)), ((1,), (6, 10))]:
        a, b = torch.randn(shp_a, generator=g), torch.randn(shp_b, generator=g)
        A, B = MLXArray(ctx, a if a.dim() else float(a)), MLXArray(ctx, b)
        for f in (lambda p, q: p + q, lambda p, q: p - q, lambda p, q: p * q, lambda p, q: q - p):
            got = f(A, B).torch().cpu()
            ref = f(a if a.dim() else torch.tensor(float(a)), b)
            assert list(got.shape) == list(ref.shape)
            assert torch.equal(got, ref), (shp_a, shp_b)
    X = MLXArray(ctx, torch.arange(6.0).reshape(2, 3))
    assert (2 * X).tolist() == [0.0, 2.0, 4.0, 6.0, 8.0, 10.0] and (X - 1).tolist()[0] == -1.0 and (1 - X).tolist()[5] == -4.0
    with pytest.raises(Flux2Error):
        MLXArray(ctx, torch.zeros(3, 4)) + MLXArray(ctx, torch.zeros(3, 5))


def test_random_normal_array_is_the_mlx_stream(flux2mi, ctx):
    """`MLXRandom.seed(s); MLXRandom.normal(shape)` through the façade = the same draw as the tensor API."""
    from flux2mi.api import MLXRandom
    r1, r2 = MLXRandom(ctx, 42), MLXRandom(ctx, 42)
    a = r1.normal_array([1, 24, 128])
    b = r2.normal((1, 24, 128))
    assert a.shape == [1, 24, 128]
    assert torch.equal(a.torch(), b)


def test_generation_with_unfused_hook_is_bit_identical(flux2mi, ctx):
    """A whole RePaint generation (strength 0.7, hook on every step) with the closure run op by op through the façade
    and the result copied back into the latent buffer = the same generation with the fused kernel, bit for bit."""
    from flux2mi import synthetic as syn
    from flux2mi.pipeline import Flux2MaskedInpaintingChain, Flux2Pipeline
    cfg, vcfg = syn.DIT_PRESETS["tiny"], syn.VAE_PRESETS["tiny"]
    pipe = Flux2Pipeline(model=cfg, vae_variant=vcfg).load_models(syn.make_dit_weights(cfg), syn.make_vae_weights(vcfg))
    inp = syn.make_inputs(cfg, 128, 128, s_txt=16)
    g = torch.Generator().manual_seed(99)
    image_latents, blend_noise = torch.randn((1, 64, 128), generator=g), torch.randn((1, 64, 128), generator=g)
    mask = torch.zeros((8, 8))
    mask[2:6, 2:6] = 1.0
    mask[1, 2:6] = 0.5
    chain = Flux2MaskedInpaintingChain(pipe, image_latents, mask.reshape(1, 64, 1), blend_noise)
    kw = dict(steps=10, strength=0.7, noise=inp["noise"], return_latents=True)
    a = chain.run(inp["txt"], 128, 128, **kw)
    b = chain.run(inp["txt"], 128, 128, unfused_hook=True, **kw)
    assert a.effectiveSteps == b.effectiveSteps == 7
    assert torch.equal(a.latents, b.latents)
    assert (a.image == b.image).all()
    pipe.clearAll()
