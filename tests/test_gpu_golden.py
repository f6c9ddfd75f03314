"""-m gpu: the HIP path against the COMMITTED golden fixtures (tests/golden/*.npz, produced by the
oracle with tests/golden/make_golden.py) — nothing here runs the oracle, so this is what the GPU box
can check even without it.  Same tolerances as the live-oracle tests."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


@pytest.fixture(scope="module")
def tiny(flux2mi, ctx):
    from flux2mi import synthetic as syn
    from flux2mi.pipeline import Flux2Pipeline
    cfg, vcfg = syn.DIT_PRESETS["tiny"], syn.VAE_PRESETS["tiny"]
    P, VP = syn.make_dit_weights(cfg), syn.make_vae_weights(vcfg)
    pipe = Flux2Pipeline(model=cfg, vae_variant=vcfg).load_models(P, VP)
    return dict(pipe=pipe, inp=syn.make_inputs(cfg, 128, 128, s_txt=16), cfg=cfg)


def test_forward_vs_golden(tiny, flux2mi):
    from flux2mi import api
    g = np.load("tests/golden/tiny_dit.npz")
    inp = tiny["inp"]
    lat = tiny["pipe"].ctx.pack_patchified_to_sequence(inp["noise"].cuda())
    out = tiny["pipe"].transformer.forward(lat[0], inp["txt"].cuda(), 0.731, 0.0,
                                           torch.from_numpy(api.image_position_ids(128, 128)).cuda(),
                                           torch.from_numpy(api.text_position_ids(16)).cuda())
    assert _rel(out.cpu().numpy(), g["forward_out"]) < 2e-2


def test_t2i_and_repaint_vs_golden(tiny):
    from flux2mi.pipeline import Flux2MaskedInpaintingChain
    g = np.load("tests/golden/tiny_dit.npz")
    t = tiny
    res = t["pipe"].generate_with_result("textToImage", height=128, width=128, steps=4,
                                         precomputed_embeddings=t["inp"]["txt"], noise=t["inp"]["noise"],
                                         return_latents=True)
    assert _rel(res.latents[0].cpu().numpy(), g["t2i_final_latents"]) < 3e-2
    assert np.abs(res.image.astype(int) - g["t2i_image"].astype(int)).mean() < 3.0
    gen = torch.Generator().manual_seed(99)
    image_latents = torch.randn((1, 64, 128), generator=gen)
    blend_noise = torch.randn((1, 64, 128), generator=gen)
    mask = torch.zeros((8, 8))
    mask[2:6, 2:6] = 1.0
    chain = Flux2MaskedInpaintingChain(t["pipe"], image_latents, mask.reshape(1, 64, 1), blend_noise)
    res = chain.run(t["inp"]["txt"], 128, 128, steps=10, strength=0.7, noise=t["inp"]["noise"], return_latents=True)
    assert res.effectiveSteps == len(g["repaint_sigmas"]) - 1 == 7
    assert _rel(res.latents[0].cpu().numpy(), g["repaint_final_latents"]) < 3e-2
    assert np.abs(res.image.astype(int) - g["repaint_image"].astype(int)).mean() < 3.0


def test_vae_vs_golden(flux2mi, ctx):
    from flux2mi import synthetic as syn
    from flux2mi.api import VAE
    g = np.load("tests/golden/tiny_vae.npz")
    vcfg = syn.VAE_PRESETS["tiny"]
    vae = VAE(ctx, vcfg).load(syn.make_vae_weights(vcfg)).finalize()
    z = torch.randn((1, 32, 8, 8), generator=torch.Generator().manual_seed(11))
    of, ou = vae.decode(z.cuda())
    assert _rel(of[0].cpu().numpy(), g["decoded"].astype(np.float32)) < 3e-2
    assert np.abs(ou.cpu().numpy().astype(int) - g["image"].astype(int)).mean() < 2.0
    vae.close()


PROD_DIT_SLICE = dict(dim=3072, heads=24, n_double=1, n_single=1, joint_dim=7680, guidance=False)


def test_production_width_dit_slice_vs_golden(flux2mi, ctx):
    """Klein-4B widths (d 3072, 24 heads, J 7680; one double + one single block): the kernels at
    their production K/N extents against the committed oracle output.  rel-L2 <= 2e-2."""
    from flux2mi import api, synthetic as syn
    from flux2mi.api import DiT
    g = np.load("tests/golden/prod_dit_slice.npz")
    dit = DiT(ctx, PROD_DIT_SLICE).load(syn.make_dit_weights(PROD_DIT_SLICE, seed=2024)).finalize()
    inp = syn.make_inputs(PROD_DIT_SLICE, 128, 128, s_txt=32, noise_seed=5, txt_seed=6)
    lat = ctx.pack_patchified_to_sequence(inp["noise"].cuda())
    for weights, tol in (("bf16", 2e-2),):
        out = dit.forward(lat[0], inp["txt"].cuda(), 0.4, 0.0, torch.from_numpy(api.image_position_ids(128, 128)).cuda(),
                          torch.from_numpy(api.text_position_ids(32)).cuda())
        assert _rel(out.cpu().numpy(), g["forward_out"]) < tol
    dit.close()


def test_production_width_vae_vs_golden(flux2mi, ctx):
    """Standard VAE decoder widths [128,256,512,512] on an 8x8 latent -> 64x64 image."""
    from flux2mi import synthetic as syn
    from flux2mi.api import VAE
    g = np.load("tests/golden/prod_vae.npz")
    vcfg = syn.VAE_PRESETS["standard"]
    vae = VAE(ctx, vcfg).load(syn.make_vae_weights(vcfg, seed=77)).finalize()
    z = torch.randn((1, 32, 8, 8), generator=torch.Generator().manual_seed(12))
    of, ou = vae.decode(z.cuda())
    assert _rel(of[0].cpu().numpy(), g["decoded"].astype(np.float32)) < 3e-2
    assert np.abs(ou.cpu().numpy().astype(int) - g["image"].astype(int)).mean() < 2.0
    vae.close()
