"""MLX-compatible random stream on the device (flux2mi_mlx_random_*) against the oracle (oracle/mlx_random.py):
integer words bit-exact, the float transforms to float32 rounding, and the pipeline's seed -> noise path."""
import numpy as np
import pytest
import torch

import flux2mi
from flux2mi import synthetic as syn
from oracle import mlx_random as M

pytestmark = pytest.mark.gpu


def _u32(t):
    return t.cpu().numpy().view(np.uint32)


@pytest.mark.parametrize("n", [1, 2, 3, 255, 256, 4097, 1 << 20, (1 << 20) + 1])
def test_bits_bit_exact(ctx, n):
    from flux2mi.api import MLXRandom
    r = MLXRandom(ctx, seed=0x1234_5678_9ABC_DEF0)
    ks = M.KeySequence(0x1234_5678_9ABC_DEF0)
    for _ in range(2):  # two keyless draws: the key state advances the same way
        got = _u32(r.bits((n,)))
        want = M.bits(ks.next_key(), n)
        assert np.array_equal(got, want)


def test_split_matches_oracle(ctx):
    from flux2mi.api import MLXRandom
    r = MLXRandom(ctx, seed=7)
    k = M.key(7)
    for _ in range(5):
        k, sub = M.split(k)
        got = r.next_key()
        assert (int(got[0]), int(got[1])) == sub
        assert (int(r.state[0]), int(r.state[1])) == k


def test_uniform_exact(ctx):
    from flux2mi.api import MLXRandom
    r = MLXRandom(ctx, seed=99)
    ks = M.KeySequence(99)
    got = r.uniform((100003,)).cpu().numpy()
    assert np.array_equal(got, ks.uniform((100003,)))  # divide, min, multiply, add: all correctly rounded
    for lo, hi in [(-3.0, 5.0), (0.1, 0.7), (-1.3, 2.9)]:   # ranges that are not powers of two: multiply, THEN add
        got = r.uniform((4096,), lo, hi).cpu().numpy()
        assert np.array_equal(got, ks.uniform((4096,), lo, hi))


@pytest.mark.parametrize("shape", [(1, 128, 64, 64), (1, 128, 16, 16), (1, 4096, 128), (1, 3, 33, 35)])
def test_normal_matches_oracle(ctx, shape):
    from flux2mi.api import MLXRandom
    r = MLXRandom(ctx, seed=42)
    ks = M.KeySequence(42)
    got = r.normal(shape).cpu().numpy()
    want = ks.normal(shape)
    # identical uniforms; log() of the device and of numpy may differ in the last place, the polynomial follows
    err = np.abs(got - want) / np.maximum(1.0, np.abs(want))
    assert err.max() < 2e-6, (err.max(), got.flatten()[err.argmax()], want.flatten()[err.argmax()])
    assert abs(got.mean()) < 0.02 and abs(got.std() - 1) < 0.02


def test_pipeline_seed_draws_the_mlx_stream(flux2mi, ctx):
    from flux2mi.pipeline import Flux2Pipeline
    cfg, vcfg = syn.DIT_PRESETS["tiny"], syn.VAE_PRESETS["tiny"]
    pipe = Flux2Pipeline(model=cfg, vae_variant=vcfg).load_models(syn.make_dit_weights(cfg), syn.make_vae_weights(vcfg))
    inp = syn.make_inputs(syn.DIT_PRESETS["tiny"], 128, 128, s_txt=16)
    noise = torch.from_numpy(M.KeySequence(2024).normal((1, 128, 8, 8)))
    a = pipe.generate_with_result("textToImage", height=128, width=128, steps=3, seed=2024, precomputed_embeddings=inp["txt"])
    b = pipe.generate_with_result("textToImage", height=128, width=128, steps=3, noise=noise, precomputed_embeddings=inp["txt"])
    c = pipe.generate_with_result("textToImage", height=128, width=128, steps=3, seed=2024, precomputed_embeddings=inp["txt"])
    assert np.array_equal(np.asarray(a.image), np.asarray(c.image))          # same seed -> same image
    diff = np.abs(np.asarray(a.image).astype(np.int32) - np.asarray(b.image).astype(np.int32))
    assert diff.max() <= 2                                                     # device draw == oracle draw (to float32 rounding)
    d = pipe.generate_with_result("textToImage", height=128, width=128, steps=3, seed=2025, precomputed_embeddings=inp["txt"])
    assert not np.array_equal(np.asarray(a.image), np.asarray(d.image))
