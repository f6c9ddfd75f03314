"""CPU: pins the oracle (and, where the function exists there, the C-ABI host code) against
every fact the reference's own XCTest suite asserts for this path, plus the
known-answer sigma tables of SURVEY §8a(a4).  Each test cites the XCTest it mirrors
(paths relative to the reference's Tests/)."""
import math

import numpy as np
import pytest
import torch

from oracle import flux2_ref as R


@pytest.fixture(scope="module")
def api(flux2mi):
    from flux2mi import api as a
    return a


def both_schedulers(api):
    def oracle(steps, seq=None, strength=1.0):
        s, t0 = R.set_timesteps(steps, seq, strength)
        return np.array(s, dtype=np.float32), t0
    return [("oracle", oracle), ("c_abi", api.set_timesteps)]


def test_scheduler_timesteps(api):
    """Flux2CoreTests/Flux2CoreTests.swift:177-188 — 50 steps -> 51 sigmas, first > 0.9, last 0."""
    for name, f in both_schedulers(api):
        s, _ = f(50)
        assert len(s) == 51, name
        assert s[0] > 0.9 and abs(s[-1]) < 1e-3, name
        assert np.all(np.diff(s) < 0), name  # strictly decreasing


def test_scheduler_i2i_strength(api):
    """Flux2CoreTests.swift:348-361 — strength 0.5 of 50 steps -> 25 steps."""
    for name, f in both_schedulers(api):
        full, _ = f(50, None, 1.0)
        half, t0 = f(50, None, 0.5)
        assert len(half) - 1 == 25 and len(half) < len(full) and t0 == 25, name
        assert np.array_equal(half, full[25:]), name


def test_scheduler_custom_sigmas(api):
    """Flux2CoreTests.swift:336-346 — 4 custom sigmas -> 5 with terminal 0; already-0 tail untouched."""
    for f in (lambda x: np.array(R.set_custom_sigmas(x), dtype=np.float32), api.set_custom_sigmas):
        s = f([1.0, 0.65, 0.35, 0.1])
        assert len(s) == 5 and s[-1] == 0.0
        s2 = f([1.0, 0.5, 0.0])
        assert len(s2) == 3
    with pytest.raises(Exception):
        api.set_custom_sigmas([])


def test_scale_noise_identity_at_sigma_zero():
    """Flux2CoreTests.swift:405-415."""
    x, n = torch.randn(1, 8, 128), torch.randn(1, 8, 128)
    assert torch.equal(R.scale_noise(x, 0.0, n), x)
    assert torch.equal(R.scale_noise(x, 1.0, n), n)


def test_empirical_mu(api):
    """Flux2CoreTests.swift:1211-1233 — mu varies with sequence length and steps; > 0 for large images."""
    for f in (R.compute_empirical_mu, api.compute_empirical_mu):
        assert f(4096, 50) != f(1024, 50)
        assert f(5000, 50) > 0
        assert f(4096, 50) != f(4096, 20)
    # known answers, f32 arithmetic
    assert abs(float(R.compute_empirical_mu(4096, 28)) - 2.151443) < 2e-6
    assert abs(api.compute_empirical_mu(4096, 50) - 2.023351) < 2e-6
    assert abs(api.compute_empirical_mu(256, 4) - 1.965567) < 2e-6
    assert abs(api.compute_empirical_mu(8192, 28) - (0.00016927 * 8192 + 0.45666666)) < 2e-6  # seq > 4300 branch


def test_sigma_known_answers_and_golden(api):
    """SURVEY §8a(a4) known answers + committed tables; C ABI == oracle to 1 ulp."""
    gold = np.load("tests/golden/scheduler_sigmas.npz")
    s28, _ = api.set_timesteps(28, 4096)
    np.testing.assert_allclose(s28[:4], [1.0, 0.995710, 0.991132, 0.986234], atol=2e-6)
    assert abs(s28[-2] - 0.241515) < 2e-6 and s28[-1] == 0.0
    s4, _ = api.set_timesteps(4, 256)
    np.testing.assert_allclose(s4, [1.0, 0.955391, 0.877134, 0.704112, 0.0], atol=2e-6)
    s07, t0 = api.set_timesteps(50, 4096, 0.7)
    assert len(s07) == 36 and t0 == 15 and abs(s07[0] - 0.946376) < 2e-6
    for name, (steps, seq, strength) in {"n28_s4096": (28, 4096, 1.0), "n50_s4096": (50, 4096, 1.0),
                                          "n50_s4096_str07": (50, 4096, 0.7), "n4_s256": (4, 256, 1.0),
                                          "n4_s4096": (4, 4096, 1.0), "n28_s8192": (28, 8192, 1.0)}.items():
        o, t0 = R.set_timesteps(steps, seq, strength)
        c, t1 = api.set_timesteps(steps, seq, strength)
        assert t0 == t1 == int(gold[name + "_tstart"][0])
        np.testing.assert_array_equal(np.array(o, dtype=np.float32), gold[name])       # oracle is pinned exactly
        np.testing.assert_allclose(c, gold[name], rtol=0, atol=1.2e-7)                  # libm expf vs numpy: <= 1 ulp


def test_strength_clamp_edges(api):
    """strength is clamped to [0.01, 1] (FlowMatchEulerScheduler.swift:95-97)."""
    for name, f in both_schedulers(api):
        s, t0 = f(28, 4096, 0.0)      # clamps to 0.01 -> Int(0.28) = 0 steps kept -> only terminal sigma
        assert len(s) == 1 and t0 == 28, name
        s, t0 = f(28, 4096, 5.0)
        assert len(s) == 29 and t0 == 0, name


def test_pack_unpack_roundtrip():
    """Flux2CoreTests.swift:139-162, 802-819 (shapes) — values too."""
    p = torch.randn(1, 128, 64, 64)
    seq = R.pack_patchified_to_sequence(p)
    assert tuple(seq.shape) == (1, 4096, 128)
    assert torch.equal(R.unpack_sequence_to_patchified(seq, 1024, 1024), p)
    z = torch.randn(1, 32, 16, 24)
    pp = R.pack_latents_to_patchified(z)
    assert tuple(pp.shape) == (1, 128, 8, 12)
    assert torch.equal(R.unpatchify_latents(pp), z)
    # channel index = c*4 + py*2 + px (LatentUtils.swift:119-142)
    assert pp[0, 5 * 4 + 1 * 2 + 0, 3, 7] == z[0, 5, 2 * 3 + 1, 2 * 7 + 0]


def test_position_ids(api):
    """Flux2CoreTests.swift:164-173, 821-832; ImageToImageTrainingTests.swift:286-339 (T = 10)."""
    ids = R.image_position_ids(1024, 1024)
    assert tuple(ids.shape) == (4096, 4)
    assert ids[0].tolist() == [0, 0, 0, 0] and ids[65].tolist() == [0, 1, 1, 0] and ids[-1].tolist() == [0, 63, 63, 0]
    assert np.array_equal(api.image_position_ids(1024, 1024), ids.numpy())
    t = R.text_position_ids(512)
    assert t[511].tolist() == [0, 0, 0, 511] and np.array_equal(api.text_position_ids(512), t.numpy())
    r = R.reference_position_ids([2, 3], [4, 2])
    assert r.shape[0] == 14 and set(r[:8, 0].tolist()) == {10} and set(r[8:, 0].tolist()) == {20}
    assert np.array_equal(api.reference_position_ids([2, 3], [4, 2]), r.numpy())
    # non-square: 80 x 112 -> 5 x 7 tokens, row-major (y, x)
    assert np.array_equal(api.image_position_ids(80, 112), R.image_position_ids(80, 112).numpy())


def test_validate_and_chain_dimensions(api):
    """Flux2ChainsTests/Flux2ChainsTests.swift:13-44; LatentUtils.validateDimensions rounds UP to x16."""
    assert api.validate_dimensions(1024, 1024) == (1024, 1024) == R.validate_dimensions(1024, 1024)
    assert api.validate_dimensions(1000, 500) == (1008, 512) == R.validate_dimensions(1000, 500)
    for f in (R.resolve_chain_dimensions, api.resolve_chain_dimensions):
        assert f(512, 512) == (512, 512)
        h, w = f(1000, 600)
        assert w % 32 == 0 and h % 32 == 0 and w <= 1000 and h <= 600
        h, w = f(3840, 2160)
        assert w * h <= 1024 * 1024 and w % 32 == 0 and h % 32 == 0 and abs(3840 / 2160 - w / h) < 0.1
        h, w = f(1024, 1024, 512 * 512)
        assert w * h <= 512 * 512 and w % 32 == 0 and h % 32 == 0
        assert f(10, 10) == (32, 32)  # min 32


def test_mask_packing():
    """Flux2ChainsTests.swift:48-124 — [1, seq, 1], all-ones / all-zeros masks keep their sum."""
    ones = torch.ones(512, 512)
    m = R.pack_mask_for_latent_blending(ones, 512, 512)
    assert tuple(m.shape) == (1, 1024, 1) and float(m.sum()) == 1024.0
    assert float(R.pack_mask_for_latent_blending(torch.zeros(512, 512), 512, 512).sum()) == 0.0
    half = torch.zeros(512, 512)
    half[:, 256:] = 1.0
    assert float(R.pack_mask_for_latent_blending(half, 512, 512).sum()) == 512.0


def test_kv_extract_mask_pattern():
    """Mirrors the reference's mask-pattern test for the KV-extraction pass (Flux2Attention.swift:407-425):
    additive mask over [txt | ref | out], -inf exactly for (reference query, output key) pairs."""
    m = R._kv_mask(3, 2, 4)
    assert m.shape == (9, 9)
    blocked = torch.isinf(m) & (m < 0)
    exp = torch.zeros((9, 9), dtype=torch.bool)
    exp[3:5, 5:] = True
    assert torch.equal(blocked, exp)
    assert (m[~exp] == 0).all()


def test_kv_cached_pass_reproduces_extraction_pass():
    """Structural identity of the KV path (oracle): a cached pass fed the extraction pass's own inputs
    returns the extraction pass's output exactly (same keys and values in the same softmax)."""
    import flux2mi
    syn = flux2mi.synthetic
    cfg = syn.DIT_PRESETS["tiny"]
    P = syn.make_dit_weights(cfg, seed=5, std=0.05)
    inp = syn.make_inputs(cfg, 64, 64, s_txt=16)
    lat = R.pack_patchified_to_sequence(inp["noise"])
    ref = torch.randn((1, 24, 128), generator=torch.Generator().manual_seed(1))
    ids = (R.image_position_ids(64, 64), R.text_position_ids(16), R.reference_position_ids([4], [6]))
    t = torch.tensor([0.6])
    o0, cache = R.dit_forward_kv(P, cfg, lat, inp["txt"].float(), t, None, ids[0], ids[1], ref_lat=ref, ref_ids=ids[2])
    o1, _ = R.dit_forward_kv(P, cfg, lat, inp["txt"].float(), t, None, ids[0], ids[1], cache=cache)
    assert len(cache) == cfg["n_double"] + cfg["n_single"] and cache[0][0].shape == (1, cfg["heads"], 24, 128)
    assert torch.allclose(o0, o1, atol=1e-5)


def test_rope_table_structure():
    """Flux2RoPE.swift:129-169 — [S,128], pairs repeated, text rows only move the L axis."""
    ids = torch.cat([R.text_position_ids(4), R.image_position_ids(32, 32)], dim=0)
    cos, sin = R.rope_tables(ids)
    assert tuple(cos.shape) == (8, 128)
    assert torch.equal(cos[:, 0::2], cos[:, 1::2]) and torch.equal(sin[:, 0::2], sin[:, 1::2])
    assert torch.all(cos[:4, :96] == 1.0) and torch.all(sin[:4, :96] == 0.0)   # T,H,W axes are 0 for text
    assert abs(float(cos[3, 96]) - math.cos(3.0)) < 1e-6                       # L axis, omega_0 = 1
    assert abs(float(sin[5, 64]) - math.sin(1.0)) < 1e-6                       # image token (0,0,1,0): W axis


def test_sdpa_explicit_matches_torch_fused():
    q, k, v = torch.randn(1, 2, 40, 128), torch.randn(1, 2, 40, 128), torch.randn(1, 2, 40, 128)
    ref = torch.nn.functional.scaled_dot_product_attention(q, k, v)
    assert torch.allclose(R.sdpa(q, k, v), ref, atol=1e-5)


def test_truncating_u8_cast():
    """Flux2Pipeline.swift:2438-2444 — clip then TRUNCATE (not round)."""
    x = torch.tensor([[[[-1.5, -1.0, 0.0, 0.999 / 127.5 - 1.0 + 1.0, 1.0, 2.0]]]]).expand(1, 3, 1, 6).clone()
    x[0, :, 0, 3] = (200.9 / 127.5) - 1.0
    out = R.postprocess_vae_output(x)
    assert out[0, :, 0].tolist() == [0, 0, 127, 200, 255, 255]


# ---- the fp8 emulation's block quantiser (the MI355X path's own scheme: test infrastructure for the -m gpu parity tests) ----
def test_fp8_block_exponents_exact_at_the_boundaries():
    """dev.h fp8_block_exp / oracle fp8_block_exponents: the largest e with amax 2^e <= 448, from the float's fields (no log2 rounding)."""
    amax = torch.tensor([448.0, 448.0000305, 447.99997, 224.0, 0.875, 0.8750001, 3.5, 3.5000002, 1e-30, 0.0, 3e38, 5e-20])
    e = R.fp8_block_exponents(amax).tolist()
    assert e[:8] == [0, -1, 0, 1, 9, 8, 7, 6]
    assert e[8] == 60 and e[9] == 60 and e[10] == -60 and e[11] == 60        # clamped to [-60, 60]; an all-zero block takes 60
    for a, ee in zip(amax[:8].tolist(), e[:8]):
        assert a * 2.0 ** ee <= 448.0 < a * 2.0 ** (ee + 1)


def test_fake_quant_e4m3_blocks_properties():
    """16 rows x 128 columns per block: blocks are independent, a block's values keep e4m3's relative precision (2^-4) whatever its
    magnitude, an all-zero block stays zero, a partial last row block is padded with zeros, `mag` only lowers the exponent."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn((1, 37, 256), generator=g)
    x[:, :16, :128] *= 2.0 ** 20
    x[:, 16:32, 128:] *= 2.0 ** -20
    x[:, 32:, :128] = 0
    q = R.fake_quant_e4m3_blocks(x)
    assert q.shape == x.shape and torch.all(q[:, 32:, :128] == 0)
    nz = x != 0
    # above the subnormals of its block (2^-6 of the block's scaled range) a value is rounded to 3 mantissa bits
    big = nz & (x.abs() > x.abs().amax() * 0)   # every non-zero value: check the bound block by block
    for rs, cs in ((slice(0, 16), slice(0, 128)), (slice(16, 32), slice(128, 256)), (slice(0, 16), slice(128, 256)), (slice(32, 37), slice(128, 256))):
        xb, qb = x[:, rs, cs], q[:, rs, cs]
        top = xb.abs().amax()
        ok = xb.abs() >= top * 2.0 ** -14
        assert float(((qb - xb).abs()[ok] / xb.abs()[ok]).max()) <= 2.0 ** -4 + 1e-6
    # independence: changing one block leaves the others bit-identical
    y = x.clone()
    y[:, :16, :128] *= 3.0
    qy = R.fake_quant_e4m3_blocks(y)
    assert torch.equal(qy[:, 16:], q[:, 16:]) and torch.equal(qy[:, :16, 128:], q[:, :16, 128:])
    # a bound instead of the maximum: same values unless they fall into the subnormals
    qm = R.fake_quant_e4m3_blocks(x, mag=x * 1.9)
    assert float((qm - q).abs().max() / q.abs().max()) < 1e-6 or torch.allclose(qm, q, rtol=2.0 ** -3, atol=0)


def test_fp8_emulation_uses_blocks_only_in_plain_forwards_with_aligned_text():
    import flux2mi
    syn = flux2mi.synthetic
    cfg = syn.DIT_PRESETS["tiny"]
    P = syn.make_dit_weights(cfg, seed=3, std=0.05)
    for s_txt, expect in ((64, True), (16, False)):
        inp = syn.make_inputs(cfg, 64, 64, s_txt=s_txt)
        lat = R.pack_patchified_to_sequence(inp["noise"])
        with R.fp8_block_gemms():
            R.dit_forward(P, cfg, lat, inp["txt"].float(), torch.tensor([0.5]), None, R.image_position_ids(64, 64), R.text_position_ids(s_txt))
            assert R._FP8_ACT_BLOCKS is expect
    R.dit_forward(P, cfg, lat, inp["txt"].float(), torch.tensor([0.5]), None, R.image_position_ids(64, 64), R.text_position_ids(s_txt))
    assert R._FP8_ACT_BLOCKS is False
