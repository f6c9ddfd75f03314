"""-m gpu: kernel-level parity of the HIP ops (through the C ABI) against the oracle /
a plain torch fp32 reference of the same op on identical bf16-rounded inputs.

Tolerances (stated): bf16-output ops rel-L2 <= 1e-2 (GEMM / attention),
f32 elementwise ops <= 1e-6 (same arithmetic, f32)."""
import math

import numpy as np
import pytest
import torch

from oracle import flux2_ref as R
from tests.util import rel_l2

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (80, 256, 128), (300, 384, 192), (768, 3072, 3072),
                                   (513, 1000, 256), (64, 64, 64), (4096, 128, 3072), (256, 32, 32)])
def test_gemm_bf16(ctx, M, N, K):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    a = torch.randn((M, K), generator=g).to(torch.bfloat16)
    w = (torch.randn((N, K), generator=g) * 0.05).to(torch.bfloat16)
    out = ctx.op_gemm_bf16(a.cuda(), w.cuda())
    ref = a.float() @ w.float().t()
    assert rel_l2(out, ref) < 2e-3  # f32 accumulate of exact bf16 products; only summation order differs


def test_gemm_identity_asymmetric(ctx):
    """A = I with an asymmetric W catches a transposed C write (cdna guide §3)."""
    n = 128
    a = torch.eye(n).to(torch.bfloat16)
    w = (torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 251 - 125).to(torch.bfloat16)
    out = ctx.op_gemm_bf16(a.cuda(), w.cuda()).cpu()
    assert torch.equal(out, w.float().t())


@pytest.mark.parametrize("S,heads", [(64, 1), (80, 2), (200, 2), (768, 4), (1153, 3)])
def test_attention(ctx, S, heads):
    g = torch.Generator().manual_seed(S + heads)
    d = heads * 128
    qkv = torch.randn((S, 3 * d), generator=g).to(torch.bfloat16)
    out = ctx.op_attention(qkv.cuda(), heads)
    q, k, v = qkv.float().split(d, dim=1)
    hv = lambda x: x.reshape(1, S, heads, 128).permute(0, 2, 1, 3)
    ref = R.sdpa(hv(q), hv(k), hv(v)).permute(0, 2, 1, 3).reshape(S, d)
    assert rel_l2(out, ref) < 1e-2


def test_attention_rescale_branch(ctx):
    """Force the online-softmax rescale: one key per later tile dominates (guide rule 26)."""
    S, heads = 512, 1
    g = torch.Generator().manual_seed(3)
    qkv = torch.randn((S, 3 * 128), generator=g) * 0.5
    qkv[:, 0] = 4.0            # q[:,0]
    for t, key in enumerate([70, 200, 330, 460]):
        qkv[key, 128] = 4.0 * (t + 1)   # k[key,0] grows tile after tile -> running max jumps each tile
    qkv = qkv.to(torch.bfloat16)
    out = ctx.op_attention(qkv.cuda(), heads)
    q, k, v = qkv.float().split(128, dim=1)
    ref = R.sdpa(q[None, None], k[None, None], v[None, None])[0, 0]
    assert rel_l2(out, ref) < 1e-2


def test_sampler_elementwise(ctx):
    g = torch.Generator().manual_seed(0)
    S = 300
    x = torch.randn((1, S, 128), generator=g)
    v = torch.randn((1, S, 128), generator=g)
    img = torch.randn((1, S, 128), generator=g)
    noise = torch.randn((1, S, 128), generator=g)
    mask = torch.rand((1, S, 1), generator=g)
    xd = x.cuda().clone()
    ctx.sched_step(xd, v.cuda(), -0.0371)
    assert rel_l2(xd, R.scheduler_step(x, v, 0.9, 0.9 - 0.0371)) < 1e-6
    out = ctx.scale_noise(img.cuda(), 0.946, noise.cuda())
    assert rel_l2(out, R.scale_noise(img, 0.946, noise)) < 1e-6
    xd = x.cuda().clone()
    ctx.repaint_blend(xd, img.cuda(), noise.cuda(), mask.cuda(), 0.3)
    assert rel_l2(xd, R.repaint_blend(x, img, noise, mask, 0.3)) < 1e-6
    xd = x.cuda().clone()
    ctx.repaint_blend(xd, img.cuda(), noise.cuda(), mask.cuda(), 0.0)   # last step: sigma_next == 0
    assert rel_l2(xd, R.repaint_blend(x, img, noise, mask, 0.0)) < 1e-6
    vc = v.cuda().clone()
    ctx.cfg_combine(vc, x.cuda(), 3.5)
    assert rel_l2(vc, R.cfg_combine(v, x, 3.5)) < 1e-6


def test_latent_layout_ops(ctx):
    g = torch.Generator().manual_seed(1)
    h16, w16 = 6, 10  # non-square
    p = torch.randn((1, 128, h16, w16), generator=g)
    seq = ctx.pack_patchified_to_sequence(p.cuda())
    assert torch.equal(seq.cpu(), R.pack_patchified_to_sequence(p))
    back = ctx.unpack_sequence_to_patchified(seq, h16, w16)
    assert torch.equal(back.cpu(), p)
    mean = torch.randn(128, generator=g) * 0.5
    var = torch.rand(128, generator=g) + 0.5
    z = ctx.latents_to_vae_input(seq, mean.cuda(), var.cuda(), h16, w16)
    zr = R.latents_to_vae_input(R.pack_patchified_to_sequence(p), h16 * 16, w16 * 16, mean, var)
    assert rel_l2(z, zr) < 1e-6
    packed = ctx.vae_latents_to_packed(z, mean.cuda(), var.cuda())
    assert rel_l2(packed, seq) < 1e-5


def test_gemm_256_tile_kernel_forced_on_ragged_shapes():
    """The 256x256 LDS-DMA kernel (gemm256.hip) on ragged / small shapes: forced through
    FLUX2MI_GEMM=256 in a child process (the selector is read once per process)."""
    import os
    import subprocess
    import sys
    code = r'''
import sys, torch
sys.path.insert(0, %r)
import flux2mi
from flux2mi.api import Context
ctx = Context(0)
worst = 0.0
for (M, N, K) in [(256, 256, 64), (300, 520, 128), (80, 256, 192), (1000, 777, 448), (513, 1024, 3072), (2560, 2560, 256)]:
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn((M, K), generator=g).to(torch.bfloat16)
    w = (torch.randn((N, K), generator=g) * 0.05).to(torch.bfloat16)
    out = ctx.op_gemm_bf16(a.cuda(), w.cuda()).cpu()
    ref = a.float() @ w.float().t()
    rel = float((out - ref).norm() / ref.norm())
    worst = max(worst, rel)
    assert rel < 2e-3, (M, N, K, rel)
# asymmetric identity check (transposed C write would show)
n = 256
a = torch.eye(n).to(torch.bfloat16)
w = (torch.arange(n * n, dtype=torch.float32).reshape(n, n) %% 251 - 125).to(torch.bfloat16)
assert torch.equal(ctx.op_gemm_bf16(a.cuda(), w.cuda()).cpu(), w.float().t())
print("ok", worst)
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FLUX2MI_GEMM="256")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


def test_gemm_large_auto_dispatch(ctx):
    """Large problem -> automatically takes the 256x256 kernel; exact-integer data, exact result."""
    M, N, K = 2560, 2816, 512
    g = torch.Generator().manual_seed(1)
    a = torch.randint(-3, 4, (M, K), generator=g).to(torch.bfloat16)
    w = torch.randint(-3, 4, (N, K), generator=g).to(torch.bfloat16)
    out = ctx.op_gemm_bf16(a.cuda(), w.cuda()).cpu()
    assert torch.equal(out, a.float() @ w.float().t())


# ---- fp8 (e4m3) GEMM path: weight_format 1 ---------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (512, 768, 256), (300, 200, 384), (1000, 1544, 1024), (64, 4096, 3072)])
def test_gemm_fp8_vs_emulation(ctx, M, N, K):
    """op_gemm_fp8 quantises A and W row-wise to e4m3 and multiplies on the fp8 MFMA; the oracle's
    fake_quant_e4m3_rows does the same arithmetic in f32.  Tolerance 1e-3 rel-L2 (f32 accumulation
    order; quantised operands are identical up to rare rounding ties), and the result must be
    quantisation-noise close (< 6e-2) to the unquantised product."""
    from oracle import flux2_ref as R
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    a = torch.randn((M, K), generator=g).to(torch.bfloat16)
    w = (torch.randn((N, K), generator=g) * 0.05).to(torch.bfloat16)
    a[0, :] = 0          # an all-zero row must not produce NaN/Inf
    a[1, 5] = 300.0      # outlier: the row scale saturates nothing
    out = ctx.op_gemm_fp8(a.cuda(), w.cuda()).cpu()
    emu = R.fake_quant_e4m3_rows(a.float()) @ R.fake_quant_e4m3_rows(w.float()).t()
    exact = a.float() @ w.float().t()
    assert torch.isfinite(out).all()
    assert torch.equal(out[0], torch.zeros(N))
    assert rel_l2(out, emu) < 1e-3
    assert rel_l2(out, exact) < 6e-2


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (512, 768, 512), (300, 200, 384), (1000, 1544, 1024), (4608, 512, 3072), (1360, 512, 1024),
                                   (2240, 256, 256)])
def test_gemm_fp8_block_scaled_activations(ctx, M, N, K):
    """op_gemm_fp8_blocks: A in 16-row x 128-column e4m3 blocks with power-of-two scales that ride in the block-scaled MFMA's scale
    operand (GemmProblem::a_bscale; byte i % 4 of two VGPRs refreshed per K-tile from scalar loads), W per row — against the oracle's
    fake_quant_e4m3_blocks.  Row blocks of very different magnitude (every MFMA row block and K-tile takes ITS byte: a wrong byte
    select, K-tile or row block shows as a power-of-two error), tile heights 5-8 x 32 rows, unaligned first row blocks (M = 1360:
    tiles of 160 rows start at row block 10, 20, ...), a partial last block, an all-zero block."""
    from oracle import flux2_ref as R
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    a = torch.randn((M, K), generator=g)
    nrb, nkb = (M + 15) // 16, K // 128
    mag = torch.exp2(torch.randint(-12, 10, (nrb, nkb), generator=g).float())          # per-block magnitudes 2^-12 .. 2^9
    a = a * mag.repeat_interleave(16, 0)[:M].repeat_interleave(128, 1)
    if M > 48:
        a[32:48, :128] = 0                                                              # an all-zero block
    a = a.to(torch.bfloat16)
    w = (torch.randn((N, K), generator=g) * 0.05).to(torch.bfloat16)
    out = ctx.op_gemm_fp8_blocks(a.cuda(), w.cuda()).cpu()
    emu = R.fake_quant_e4m3_blocks(a.float()) @ R.fake_quant_e4m3_rows(w.float()).t()
    exact = a.float() @ w.float().t()
    assert torch.isfinite(out).all()
    # per 16-row block: the magnitudes differ by up to 2^21 between blocks
    for rb in range(0, M, 16 * max(1, nrb // 24)):
        sl = slice(rb, min(rb + 16, M))
        assert rel_l2(out[sl], emu[sl]) < 1e-3, (rb, rel_l2(out[sl], emu[sl]))
    assert rel_l2(out, emu) < 1e-3
    assert rel_l2(out, exact) < 6e-2
    assert torch.equal(out, ctx.op_gemm_fp8_blocks(a.cuda(), w.cuda()).cpu())


def test_gemm_fp8_rejects_bad_k(ctx):
    from flux2mi import Flux2Error
    a = torch.zeros((256, 192), dtype=torch.bfloat16, device="cuda")
    w = torch.zeros((256, 192), dtype=torch.bfloat16, device="cuda")
    with pytest.raises(Flux2Error):
        ctx.op_gemm_fp8(a, w)


# ---- production-shaped launches of the 256x256 kernel, exact data ------------------------------------
@pytest.mark.parametrize("M,N,K", [
    (4608, 3072, 3072),    # 216 tiles (< 256 CUs)
    (4608, 9216, 1024),    # 648 tiles
    (4500, 3000, 2048),    # ragged edges
    (4608, 3072, 64),      # a single K-tile
    (2304, 27648, 512),    # wide N
    (1536, 8192, 512),     # tile height 192 (256 workgroups instead of 192 of 256x256)
    (4608, 4096, 1024),    # Klein-9B width: tile height 160 (464 workgroups, 2 rounds of 5/8 tiles)
    (4500, 4096, 512),     # the same with a ragged last row block
    (4608, 6144, 512),     # Dev width: tile height 224
])
def test_gemm_production_shapes_exact(ctx, M, N, K):
    """Integer-valued operands: every partial sum is exact in f32, so the result must equal the
    reference bit for bit."""
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randint(-3, 4, (M, K), generator=g).to(torch.bfloat16)
    w = torch.randint(-3, 4, (N, K), generator=g).to(torch.bfloat16)
    out = ctx.op_gemm_bf16(a.cuda(), w.cuda()).cpu()
    assert torch.equal(out, a.float() @ w.float().t())


def test_gemm_deterministic(ctx):
    """Two runs on random data are bit-identical (no atomics, no timing-dependent summation order)."""
    g = torch.Generator().manual_seed(11)
    a = torch.randn((4608, 3072), generator=g).to(torch.bfloat16).cuda()
    w = (torch.randn((3072, 3072), generator=g) * 0.05).to(torch.bfloat16).cuda()
    o1 = ctx.op_gemm_bf16(a, w).cpu()
    o2 = ctx.op_gemm_bf16(a, w).cpu()
    assert torch.equal(o1, o2)
    assert rel_l2(o1, a.float().cpu() @ w.float().cpu().t()) < 2e-3


def test_gemm_race_screen(ctx):
    """The 256x256 kernel keeps LDS-DMA loads in flight across barriers behind ONE counted vmcnt per
    K-tile; a mis-placed wait shows up as rare wrong tiles that depend on timing.  Screen: the
    production linear1 / linear2 shapes on exact-integer data, 25 launches each, every result
    compared on the device with the f32 reference."""
    for (M, N, K) in [(4608, 27648, 3072), (4608, 3072, 12288)]:
        g = torch.Generator(device="cuda").manual_seed(M + K)
        a = torch.randint(-2, 3, (M, K), generator=g, device="cuda").to(torch.bfloat16)
        w = torch.randint(-2, 3, (N, K), generator=g, device="cuda").to(torch.bfloat16)
        ref = a.float() @ w.float().t()
        for it in range(25):
            out = ctx.op_gemm_bf16(a, w)
            assert torch.equal(out, ref), (M, N, K, it)


# ---- gate + residual epilogue of the block-output GEMMs ------------------------------------------------
@pytest.mark.parametrize("M,N,K", [
    (4608, 3072, 12288),   # single.linear2: 216 tiles, 192 K-tiles
    (4608, 3072, 9216),    # double.mlp_out
    (4500, 3000, 6144),    # ragged edges
    (4608, 3072, 3072),    # attention projection
    (2304, 3072, 12288),   # 512x512 image
    (4608, 4096, 4096),    # Klein-9B width (tile height 160)
    (100, 512, 256),       # every tile partial in M: the transposed-domain epilogue's predicated form only
    (4645, 3072, 3072),    # whole tiles (residual requested a batch ahead) and a last row of tiles with 37 rows
    (4608, 4096, 4096),    # Klein-9B's N = d launches: 288 tiles of 256 rows = the tail split (one round of 256 tiles + a K split of the last 512 rows)
    (768, 3072, 12288),    # linear2 at 256^2: 36 tiles, split-K into 7 ranges on the 256 x 256 kernel
])
def test_gemm_gate_resid_exact(ctx, M, N, K):
    """x += gate * (A W^T) on integer-valued data (every partial sum exact in f32): bit-exact against
    the reference and identical on repeated launches."""
    g = torch.Generator(device="cuda").manual_seed(M + K)
    a = torch.randint(-2, 3, (M, K), generator=g, device="cuda").to(torch.bfloat16)
    w = torch.randint(-2, 3, (N, K), generator=g, device="cuda").to(torch.bfloat16)
    gate = torch.randint(-2, 3, (N,), generator=g, device="cuda").float()
    x0 = torch.randint(-8, 9, (M, N), generator=g, device="cuda").float()
    ref = x0 + gate[None, :] * (a.float() @ w.float().t())
    for it in range(6):
        x = ctx.op_gemm_gate_resid(a, w, gate, x0.clone())
        assert torch.equal(x, ref), (M, N, K, it)


def test_gemm_gate_resid_beyond_32bit_offsets(ctx):
    """X of 4 GiB or more: the transposed-domain epilogue addresses X by 32-bit byte offsets, so the launcher must hand such a
    problem to the generic (64-bit offset) epilogue (csrc/gemm256s.hip, `grt` predicate) — rows behind the 4 GiB line included.
    Integer-valued data: exact."""
    M, N, K = 33024, 32768, 64                       # M * N * 4 = 4.03 GiB
    g = torch.Generator(device="cuda").manual_seed(77)
    a = torch.randint(-2, 3, (M, K), generator=g, device="cuda").to(torch.bfloat16)
    w = torch.randint(-2, 3, (N, K), generator=g, device="cuda").to(torch.bfloat16)
    gate = torch.randint(-2, 3, (N,), generator=g, device="cuda").float()
    x = torch.full((M, N), 3.0, device="cuda")
    x = ctx.op_gemm_gate_resid(a, w, gate, x)
    wf = w.float().t()
    for lo, hi in ((0, 300), (16000, 16300), (32700, M)):     # the last band lies behind byte offset 2^32
        ref = 3.0 + gate[None, :] * (a[lo:hi].float() @ wf)
        assert torch.equal(x[lo:hi], ref), (lo, hi)
    del x
    torch.cuda.empty_cache()


def test_gemm_gate_resid_random_deterministic(ctx):
    g = torch.Generator(device="cuda").manual_seed(5)
    a = torch.randn((4608, 12288), generator=g, device="cuda").to(torch.bfloat16)
    w = (torch.randn((3072, 12288), generator=g, device="cuda") * 0.02).to(torch.bfloat16)
    gate = torch.randn((3072,), generator=g, device="cuda")
    x0 = torch.randn((4608, 3072), generator=g, device="cuda")
    outs = [ctx.op_gemm_gate_resid(a, w, gate, x0.clone()) for _ in range(4)]
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    ref = x0 + gate[None, :] * (a.float() @ w.float().t())
    assert rel_l2(outs[0], ref) < 2e-3


# ---- per-op parity of the HBM-bound kernels (LN/GN/modulate <= 4e-3) --------------------
def _big_ids(S, seed=0):
    """Position ids with every axis near its maximum: T up to 30 (three references), H / W up to 127 (2048^2), L up to 511."""
    g = torch.Generator().manual_seed(seed)
    ids = torch.stack([torch.randint(0, 31, (S,), generator=g), torch.randint(0, 128, (S,), generator=g),
                       torch.randint(0, 128, (S,), generator=g), torch.randint(0, 512, (S,), generator=g)], dim=1).to(torch.int32)
    ids[0] = torch.tensor([30, 127, 127, 511])
    return ids


@pytest.mark.parametrize("dim,rows,split", [(3072, 349, 77), (4096, 1030, 512), (6144, 517, 0), (256, 54, 19)])
def test_op_layernorm_modulate(flux2mi, ctx, dim, rows, split):
    """LayerNorm (eps 1e-6, no affine) + per-stream (1 + scale) x + shift at Klein-4B / Klein-9B / Dev widths, ragged rows,
    an offset-heavy residual stream (mean 3, spread 2 — the two-pass variance must not cancel); bf16 output; rel-L2 <= 4e-3."""
    g = torch.Generator().manual_seed(dim)
    x = torch.randn((rows, dim), generator=g) * 2.0 + 3.0
    sh = [torch.randn(dim, generator=g) * 0.3 for _ in range(2)]
    sc = [torch.randn(dim, generator=g) * 0.3 for _ in range(2)]
    y = ctx.op_layernorm_modulate(x.cuda(), split, sh[0].cuda(), sc[0].cuda(), sh[1].cuda(), sc[1].cuda()).float().cpu()
    n = R.layer_norm_noaffine(x.unsqueeze(0))[0]
    ref = torch.cat([R.apply_modulation(n[:split].unsqueeze(0), sh[0].unsqueeze(0), sc[0].unsqueeze(0))[0],
                     R.apply_modulation(n[split:].unsqueeze(0), sh[1].unsqueeze(0), sc[1].unsqueeze(0))[0]], dim=0)
    assert rel_l2(y, ref) < 4e-3


def _qk_ref(q_pre, k_pre, heads, qw, kw, ids, q_mul):
    cos, sin = R.rope_tables(ids)
    S = q_pre.shape[0]
    hd = lambda t: t.reshape(1, S, heads, 128).permute(0, 2, 1, 3)
    q = R.apply_rope(R.rms_norm(hd(q_pre), qw), cos, sin) * q_mul
    k = R.apply_rope(R.rms_norm(hd(k_pre), kw), cos, sin)
    un = lambda t: t.permute(0, 2, 1, 3).reshape(S, heads * 128)
    return un(q), un(k)


@pytest.mark.parametrize("heads,S,split", [(24, 349, 77), (32, 300, 0), (48, 130, 64), (2, 54, 19)])
def test_op_qk_norm_rope_standalone(flux2mi, ctx, heads, S, split):
    """The standalone RMSNorm + RoPE kernel at d 3072 / 4096 / 6144 with positions at the top of every axis; per-stream
    norm weights (double blocks: text rows < split); rel-L2 <= 4e-3 for q and k, v bit-untouched."""
    d = heads * 128
    g = torch.Generator().manual_seed(S)
    qkv = (torch.randn((S, 3 * d), generator=g) * 1.7).to(torch.bfloat16)
    w = [(1.0 + 0.2 * torch.randn(128, generator=g)).to(torch.bfloat16) for _ in range(4)]   # qw0, kw0, qw1, kw1
    ids = _big_ids(S)
    q_mul = 1.4426950408889634 / 128 ** 0.5
    out = ctx.op_qk_norm_rope(qkv.cuda(), heads, split, *[x.cuda() for x in w], ids.cuda(), q_mul).cpu()
    qf, kf = qkv[:, :d].float(), qkv[:, d:2 * d].float()
    q0, k0 = _qk_ref(qf, kf, heads, w[0], w[1], ids, q_mul)
    q1, k1 = _qk_ref(qf, kf, heads, w[2], w[3], ids, q_mul)
    rows = torch.arange(S).unsqueeze(1) < split
    assert rel_l2(out[:, :d], torch.where(rows, q0, q1)) < 4e-3
    assert rel_l2(out[:, d:2 * d], torch.where(rows, k0, k1)) < 4e-3
    assert torch.equal(out[:, 2 * d:], qkv[:, 2 * d:])


@pytest.mark.parametrize("dim,S", [(3072, 349), (4096, 300), (6144, 261)])
def test_op_qk_proj_norm_rope_fused_epilogue(flux2mi, ctx, dim, S):
    """The same math in the 256x256 GEMM's epilogue (bf16 models): x W^T -> RMSNorm -> RoPE with the rotation angles computed
    in-kernel from the ids as v_sin / v_cos(fract(pos theta^(-k/16) / 2 pi)).  Positions at the top of every axis bound the
    argument-reduction error: the largest angle is 511 rad = 81 revolutions, whose fract keeps 17 of f32's 24 bits.
    rel-L2 <= 4e-3 vs the oracle on the f32 product."""
    heads = dim // 128
    g = torch.Generator().manual_seed(dim + 1)
    x = torch.randn((S, dim), generator=g).to(torch.bfloat16)
    w = (torch.randn((2 * dim, dim), generator=g) * 0.03).to(torch.bfloat16)
    qw = (1.0 + 0.2 * torch.randn(128, generator=g)).to(torch.bfloat16)
    kw = (1.0 + 0.2 * torch.randn(128, generator=g)).to(torch.bfloat16)
    ids = _big_ids(S, seed=2)
    q_mul = 1.4426950408889634 / 128 ** 0.5
    out = ctx.op_qk_proj_norm_rope(x.cuda(), w.cuda(), qw.cuda(), kw.cuda(), ids.cuda(), q_mul).float().cpu()
    p = x.float() @ w.float().t()
    q, k = _qk_ref(p[:, :dim], p[:, dim:], heads, qw, kw, ids, q_mul)
    assert rel_l2(out[:, :dim], q) < 4e-3
    assert rel_l2(out[:, dim:], k) < 4e-3
    # and the two forms agree with each other to bf16 rounding of the intermediate (the standalone one reads a bf16 product)
    qkv = torch.cat([p.to(torch.bfloat16), torch.zeros((S, dim), dtype=torch.bfloat16)], dim=1)
    alone = ctx.op_qk_norm_rope(qkv.cuda(), heads, 0, qw.cuda(), kw.cuda(), qw.cuda(), kw.cuda(), ids.cuda(), q_mul).float().cpu()
    assert rel_l2(out, alone[:, :2 * dim]) < 6e-3


@pytest.mark.parametrize("C,hw,silu", [(128, 96, True), (256, 64, True), (512, 32, False), (96, 40, True), (384, 24, True)])
def test_op_groupnorm_silu(flux2mi, ctx, C, hw, silu):
    """GroupNorm(32 groups, eps 1e-6, f32 statistics) [+ SiLU] on NHWC bf16 at the VAE's channel counts (standard and small
    decoder) with a per-channel offset (mean != 0: E[x^2] - E[x]^2 in f32 would lose digits); rel-L2 <= 4e-3."""
    g = torch.Generator().manual_seed(C)
    x = (torch.randn((1, C, hw, hw), generator=g) * 1.5 + torch.randn((1, C, 1, 1), generator=g) * 2.0).to(torch.bfloat16)
    gamma, beta = 1.0 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    y = ctx.op_groupnorm_silu(x[0].permute(1, 2, 0).contiguous().cuda(), 32, gamma.cuda(), beta.cuda(), 1e-6, silu).float().cpu()
    ref = R.group_norm(x.float(), gamma, beta, 32, 1e-6)
    if silu:
        ref = torch.nn.functional.silu(ref)
    assert rel_l2(y.permute(2, 0, 1), ref[0]) < 4e-3


# ---- long sequences with a running maximum that keeps growing ------------------------------------------------------------
@pytest.mark.parametrize("S,heads", [(1500, 6), (4608, 3), (1100, 40), (2500, 9)])
def test_attention_growing_maxima(ctx, S, heads):
    """8-wave workgroups (S >= 1024) on grids that leave a partial last round (36 / 54 / 200 / 90 items on 256 CUs).  The keys
    grow in norm along the sequence, so the online softmax rescales in almost every tile and one dominant key sits in the
    middle of every head (guide rule 26: force the rare branch).  Full-tensor check against the oracle's softmax, rel-L2
    <= 1e-2.  (op_attention is the un-prescaled 8-wave form; the KV-split of the one-wave-per-SIMD kernel has its own test below.)"""
    g = torch.Generator().manual_seed(S * 3 + heads)
    d = heads * 128
    qkv = torch.randn((S, 3 * d), generator=g) * 0.6
    ramp = torch.linspace(0.3, 2.5, S).unsqueeze(1)
    qkv[:, d:2 * d] *= ramp                                 # later keys score higher: maxima jump piece after piece
    qkv[S // 2, d:2 * d] *= 3.0                             # and one dominant key in the middle of every head
    qkv = qkv.to(torch.bfloat16)
    out = ctx.op_attention(qkv.cuda(), heads)
    q, k, v = qkv.float().split(d, dim=1)
    hv = lambda x: x.reshape(1, S, heads, 128).permute(0, 2, 1, 3)
    ref = R.sdpa(hv(q), hv(k), hv(v)).permute(0, 2, 1, 3).reshape(S, d)
    assert torch.isfinite(out).all()
    assert rel_l2(out, ref) < 1e-2
    # bit-reproducible
    assert torch.equal(out, ctx.op_attention(qkv.cuda(), heads))


@pytest.mark.parametrize("S,heads", [(64, 1), (128, 2), (192, 3), (256, 2), (320, 1), (448, 2), (704, 3), (1216, 2), (200, 2), (1153, 1)])
def test_attention_prescaled_fixed_offset(ctx, S, heads):
    """The form the blocks run (flux2mi_op_attention_prescaled): q pre-multiplied by log2(e)/sqrt(128), unit-RMS q / k rows as the
    QK RMSNorm leaves them, softmax against a fixed zero offset: the one-wave-per-SIMD kernel (attention64b.hip) — 1, 2, 3 KV
    tiles (prologue / tail paths only), 4-7 (one trip of the three-tile loop + 0-3 tail tiles), 11, 19, and key counts that are not
    multiples of 64 (200, 1153: clamped K rows + masked row sums in the last tile) — against the oracle's softmax on the un-scaled
    q (rel-L2 <= 1e-2), bit-reproducible, and within bf16 rounding of the running-maximum path (8-wave kernel) on the same operands."""
    import math
    g = torch.Generator().manual_seed(S * 7 + heads)
    d = heads * 128
    x = torch.randn((S, 3 * d), generator=g)
    q = x[:, :d].view(S, heads, 128)
    k = x[:, d:2 * d].view(S, heads, 128)
    q = q / q.pow(2).mean(-1, keepdim=True).sqrt()
    k = k / k.pow(2).mean(-1, keepdim=True).sqrt()
    k[S // 2] *= 1.5                       # one dominant key per head (still inside the bound below)
    c = math.log2(math.e) / math.sqrt(128.0)
    qkv = torch.cat([(q * c).reshape(S, d), k.reshape(S, d), x[:, 2 * d:]], dim=1).to(torch.bfloat16)
    bound = 1.5 * 128.0 * c                # |q.k| <= 128 * 1.5 for these rows, in the log2 domain
    out = ctx.op_attention_prescaled(qkv.cuda(), heads, bound)
    qs, ks, vs = qkv.float().split(d, dim=1)
    hv = lambda t: t.reshape(1, S, heads, 128).permute(0, 2, 1, 3)
    ref = R.sdpa(hv(qs / c), hv(ks), hv(vs)).permute(0, 2, 1, 3).reshape(S, d)
    assert torch.isfinite(out).all()
    assert rel_l2(out, ref) < 1e-2
    assert torch.equal(out, ctx.op_attention_prescaled(qkv.cuda(), heads, bound))
    online = ctx.op_attention_prescaled(qkv.cuda(), heads, 0.0)    # no bound: running maximum on the same pre-scaled q
    assert rel_l2(out, online.float()) < 6e-3


def test_attention_prescaled_large_scores(ctx):
    """Scores near the upper end of the fixed-offset range (|s| up to ~70 in the log2 domain, bound 75 <= 80): p = exp2(s) spans
    2^-70 .. 2^70 across rows, the softmax is nearly one-hot — still the oracle's result, finite, on the one-wave-per-SIMD kernel."""
    import math
    S, heads = 704, 2
    g = torch.Generator().manual_seed(11)
    d = heads * 128
    x = torch.randn((S, 3 * d), generator=g)
    q = x[:, :d].view(S, heads, 128)
    k = x[:, d:2 * d].view(S, heads, 128)
    q = q / q.pow(2).mean(-1, keepdim=True).sqrt()
    k = k / k.pow(2).mean(-1, keepdim=True).sqrt()
    k[::7] = q[::7] * 4.4              # aligned keys: q.k / sqrt(128) * log2(e) = 128 * 4.4 * 0.1275 = 71.8
    k[3::7] = -q[3::7] * 4.4           # and anti-aligned ones
    c = math.log2(math.e) / math.sqrt(128.0)
    qkv = torch.cat([(q * c).reshape(S, d), k.reshape(S, d), x[:, 2 * d:]], dim=1).to(torch.bfloat16)
    out = ctx.op_attention_prescaled(qkv.cuda(), heads, 75.0)
    qs, ks, vs = qkv.float().split(d, dim=1)
    smax = float((qs.view(S, heads, 128).permute(1, 0, 2) @ ks.view(S, heads, 128).permute(1, 2, 0)).abs().max())
    assert 60.0 < smax <= 75.0, smax
    hv = lambda t: t.reshape(1, S, heads, 128).permute(0, 2, 1, 3)
    ref = R.sdpa(hv(qs / c), hv(ks), hv(vs)).permute(0, 2, 1, 3).reshape(S, d)
    assert torch.isfinite(out).all()
    assert rel_l2(out, ref) < 1e-2
@pytest.mark.gpu
@pytest.mark.parametrize("S,heads", [(64, 1), (200, 2), (448, 3), (1153, 2), (4608, 2)])
def test_attention_prescaled_data_offset(ctx, S, heads):
    """No usable score bound (bound 0 = unknown, or > 80): the same one-wave-per-SIMD kernel with a per-query offset m_q = the query's
    largest score in KV tile 0 (attention64b.hip, DOFF) — vs the oracle's softmax (rel-L2 <= 1e-2), bit-reproducible, and the same
    result whether the bound is missing or merely too large."""
    import math
    g = torch.Generator().manual_seed(S * 5 + heads)
    d = heads * 128
    x = torch.randn((S, 3 * d), generator=g)
    q = x[:, :d].view(S, heads, 128)
    k = x[:, d:2 * d].view(S, heads, 128)
    q = q / q.pow(2).mean(-1, keepdim=True).sqrt()
    k = k / k.pow(2).mean(-1, keepdim=True).sqrt()
    q = q * 4.0                            # |scores| up to 98 in the log2 domain: outside the fixed-offset range (80)
    k[S // 2] *= 1.5
    c = math.log2(math.e) / math.sqrt(128.0)
    qkv = torch.cat([(q * c).reshape(S, d), k.reshape(S, d), x[:, 2 * d:]], dim=1).to(torch.bfloat16)
    out = ctx.op_attention_prescaled(qkv.cuda(), heads, 0.0)
    qs, ks, vs = qkv.float().split(d, dim=1)
    hv = lambda t: t.reshape(1, S, heads, 128).permute(0, 2, 1, 3)
    ref = R.sdpa(hv(qs / c), hv(ks), hv(vs)).permute(0, 2, 1, 3).reshape(S, d)
    assert torch.isfinite(out).all()
    assert rel_l2(out, ref) < 1e-2
    assert torch.equal(out, ctx.op_attention_prescaled(qkv.cuda(), heads, 0.0))
    assert torch.equal(out, ctx.op_attention_prescaled(qkv.cuda(), heads, 4.0 * 1.5 * 128.0 * c))


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["late_peak", "low_first_tile"])
def test_attention_prescaled_data_offset_repeated_pass(ctx, case):
    """The safety net of the data-offset kernel: scores that exceed the KV-tile-0 estimate by > 127 in the log2 domain overflow the
    row sum, and the workgroup repeats its pass with the offsets of exactly those queries raised (twice or more here: +100 per overflow,
    then the exact log2 of the sum).  Every second query has no such key (its offset stays), ragged S, three heads."""
    import math
    S, heads = 713, 3
    g = torch.Generator().manual_seed(23)
    d = heads * 128
    x = torch.randn((S, 3 * d), generator=g)
    q = x[:, :d].view(S, heads, 128).clone()
    k = x[:, d:2 * d].view(S, heads, 128).clone()
    q = q / q.pow(2).mean(-1, keepdim=True).sqrt()
    k = k / k.pow(2).mean(-1, keepdim=True).sqrt()
    u = torch.randn((heads, 128), generator=g)
    u = u / u.pow(2).mean(-1, keepdim=True).sqrt()
    q[1::2] = q[1::2] - (q[1::2] * u).sum(-1, keepdim=True) / 128.0 * u   # the others are orthogonal to it
    q[::2] = q[::2] * 0.25 + u             # every second query carries the common direction u
    c = math.log2(math.e) / math.sqrt(128.0)
    if case == "late_peak":
        k[300] = u * 20.0                  # score ~ +330 (log2 domain) for those queries, ~0 for the others; tile 0 holds ordinary keys
        k[500] = -u * 20.0
    else:
        k[:64] = -u * 20.0 + k[:64] * 0.1  # tile 0 scores ~ -330 for those queries: the estimate sits far below the rest of the row
    qkv = torch.cat([(q * c).reshape(S, d), k.reshape(S, d), x[:, 2 * d:]], dim=1).to(torch.bfloat16)
    out = ctx.op_attention_prescaled(qkv.cuda(), heads, 0.0)
    qs, ks, vs = qkv.float().split(d, dim=1)
    sc = qs.view(S, heads, 128).permute(1, 0, 2) @ ks.view(S, heads, 128).permute(1, 2, 0)
    jump = (sc.max(dim=2).values - sc[:, :, :64].max(dim=2).values)
    assert float(jump.max()) > 250.0 and float(jump[:, 1::2].max()) < 90.0, (float(jump.max()), float(jump[:, 1::2].max()))
    hv = lambda t: t.reshape(1, S, heads, 128).permute(0, 2, 1, 3)
    ref = R.sdpa(hv(qs / c), hv(ks), hv(vs)).permute(0, 2, 1, 3).reshape(S, d)
    assert torch.isfinite(out).all()
    assert rel_l2(out, ref) < 1e-2
    assert torch.equal(out, ctx.op_attention_prescaled(qkv.cuda(), heads, 0.0))




@pytest.mark.gpu
def test_mfma_stream_probe(ctx):
    """flux2mi_debug_mfma_stream (bench.py's roofline.mfma_stream): the register-only MFMA stream reports a rate between a third of
    the data-sheet peak and the peak itself (bf16 2.5 PF, e4m3 5 PF) — it is a measurement of the card, not a constant."""
    bf16 = ctx.mfma_stream_tflops(fp8=False, seconds=0.2)
    fp8 = ctx.mfma_stream_tflops(fp8=True, seconds=0.2)
    assert 800.0 < bf16 < 2600.0, bf16
    assert 1600.0 < fp8 < 5200.0, fp8
    assert fp8 > 1.5 * bf16


# ---- split-K of the 128x128 kernel (few tiles, long K: the text encoders' M = 512 projections) ----------
@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [
    (512, 2560, 9728),   # Qwen3-4B down_proj: 80 tiles, 152 K-tiles -> 6 K ranges
    (512, 2560, 4096),   # o_proj: 6 ranges of 10-11 K-tiles
    (512, 6144, 2560),   # qkv_proj: 192 tiles -> 2 ranges
    (500, 2500, 4096),   # ragged rows and columns
    (128, 128, 1024),    # one tile, 16 K-tiles -> 2 ranges of 8
    (384, 640, 1088),    # 17 K-tiles: uneven ranges
])
def test_gemm_split_k_exact_all_epilogues(ctx, M, N, K):
    """launch_gemm cuts the K dimension of such launches into ranges (one workgroup per tile and range), parks the partial
    accumulators in a workspace and a second kernel adds them in range order and applies the epilogue (gemm.hip,
    gemm_splitk_reduce_kernel).  Integer-valued operands make every partial sum exact: f32, bf16-rounded and gate + residual
    outputs must equal the reference bit for bit, on repeated launches."""
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    a = torch.randint(-2, 3, (M, K), generator=g, device="cuda").to(torch.bfloat16)
    w = torch.randint(-2, 3, (N, K), generator=g, device="cuda").to(torch.bfloat16)
    ref = a.float() @ w.float().t()
    gate = torch.randint(-2, 3, (N,), generator=g, device="cuda").float()
    x0 = torch.randint(-8, 9, (M, N), generator=g, device="cuda").float()
    for _ in range(3):
        assert torch.equal(ctx.op_gemm_bf16(a, w), ref)
        out16 = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
        ctx.op_gemm_bf16_out_bf16(a, w, out16)
        ctx.sync()
        assert torch.equal(out16, ref.to(torch.bfloat16))
        assert torch.equal(ctx.op_gemm_gate_resid(a, w, gate, x0.clone()), x0 + gate[None, :] * ref)


@pytest.mark.gpu
def test_gemm_split_k_random_deterministic(ctx):
    g = torch.Generator(device="cuda").manual_seed(9)
    a = torch.randn((512, 9728), generator=g, device="cuda").to(torch.bfloat16)
    w = (torch.randn((2560, 9728), generator=g, device="cuda") * 0.02).to(torch.bfloat16)
    outs = [ctx.op_gemm_bf16(a, w) for _ in range(4)]
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    assert rel_l2(outs[0].cpu(), (a.float() @ w.float().t()).cpu()) < 2e-3


@pytest.mark.gpu
@pytest.mark.parametrize("doff", [False, True])
@pytest.mark.parametrize("S,q0,nq,nk", [(704, 100, 300, 400), (704, 64, 45, 109), (320, 19, 60, 79), (1217, 512, 700, 1212), (200, 0, 200, 1), (4608, 512, 3600, 4112)])
def test_attention_prescaled_key_prefix_with_live_values_behind(ctx, S, q0, nq, nk, doff):
    """The launch shape of the KV-extraction pass (dit.cpp attend(): reference queries against the [text | reference] key PREFIX of
    the shared V^T image, Flux2Attention.swift:407-425): nk % 64 != 0 with REAL, large values behind key nk - 1.  The missing keys of
    the partial last KV tile must contribute nothing — per query row against the oracle's softmax over the prefix (every row's
    relative error, not a whole-tensor norm), for both softmax-offset forms; rows outside the query range are not written."""
    import math
    heads = 2
    g = torch.Generator().manual_seed(S * 3 + nk)
    d = heads * 128
    x = torch.randn((S, 3 * d), generator=g)
    q = x[:, :d].view(S, heads, 128)
    k = x[:, d:2 * d].view(S, heads, 128)
    q = q / q.pow(2).mean(-1, keepdim=True).sqrt() * (4.0 if doff else 1.0)
    k = k / k.pow(2).mean(-1, keepdim=True).sqrt()
    v = x[:, 2 * d:].clone()
    v[nk:] = 40.0 + 10.0 * v[nk:]            # what a leak of the keys behind the prefix would add: far outside the prefix's value range
    c = math.log2(math.e) / math.sqrt(128.0)
    qkv = torch.cat([(q * c).reshape(S, d), k.reshape(S, d), v], dim=1).to(torch.bfloat16)
    bound = 0.0 if doff else 128.0 * c
    out = ctx.op_attention_prescaled_range(qkv.cuda(), heads, bound, q0, nq, nk).float().cpu()
    qs, ks, vs = qkv.float().split(d, dim=1)
    hv = lambda t: t.reshape(1, t.shape[0], heads, 128).permute(0, 2, 1, 3)
    ref = R.sdpa(hv(qs[q0:q0 + nq] / c), hv(ks[:nk]), hv(vs[:nk])).permute(0, 2, 1, 3).reshape(nq, d)
    got = out[q0:q0 + nq]
    assert torch.isfinite(out).all()
    row_err = (got - ref).norm(dim=1) / ref.norm(dim=1).clamp_min(1e-3)
    assert float(row_err.max()) < 3e-2, (float(row_err.max()), int(row_err.argmax()))
    assert rel_l2(got, ref) < 1e-2
    assert float(out[:q0].abs().max() if q0 else 0.0) == 0.0 and float(out[q0 + nq:].abs().max() if q0 + nq < S else 0.0) == 0.0
    assert torch.equal(out, ctx.op_attention_prescaled_range(qkv.cuda(), heads, bound, q0, nq, nk).float().cpu())


@pytest.mark.gpu
def test_attention_prescaled_random_shapes(ctx):
    """A seeded sweep of 24 sequence lengths (1 .. 1500, most of them not multiples of the 64-key tile or the 256-query block) x 1-4
    heads through both softmax forms of the one-wave-per-SIMD kernel (fixed zero offset with a valid bound, per-query data offset
    without one) against the oracle: prologue-only, tail-only and loop paths, clamped K rows and masked row sums at every remainder."""
    import math
    import random
    rng = random.Random(2024)
    c = math.log2(math.e) / math.sqrt(128.0)
    cases = [(1, 1), (63, 2), (65, 1), (129, 3), (255, 2), (257, 1)] + [(rng.randint(1, 1500), rng.randint(1, 4)) for _ in range(18)]
    for S, heads in cases:
        g = torch.Generator().manual_seed(S * 13 + heads)
        d = heads * 128
        x = torch.randn((S, 3 * d), generator=g)
        q = x[:, :d].view(S, heads, 128)
        k = x[:, d:2 * d].view(S, heads, 128)
        q = q / q.pow(2).mean(-1, keepdim=True).sqrt()
        k = k / k.pow(2).mean(-1, keepdim=True).sqrt()
        qkv = torch.cat([(q * c).reshape(S, d), k.reshape(S, d), x[:, 2 * d:]], dim=1).to(torch.bfloat16)
        qs, ks, vs = qkv.float().split(d, dim=1)
        hv = lambda t: t.reshape(1, S, heads, 128).permute(0, 2, 1, 3)
        ref = R.sdpa(hv(qs / c), hv(ks), hv(vs)).permute(0, 2, 1, 3).reshape(S, d)
        for bound in (128.0 * c, 0.0):
            out = ctx.op_attention_prescaled(qkv.cuda(), heads, bound)
            assert torch.isfinite(out).all(), (S, heads, bound)
            assert rel_l2(out, ref) < 1e-2, (S, heads, bound, rel_l2(out, ref))


@pytest.mark.gpu
def test_gemm_random_shapes_exact(ctx):
    """A seeded sweep of 20 shapes over the dispatch (128x128 kernel with and without split-K, gemm256s at every tile height, ragged
    M and N, 1 .. 64 K-tiles) on integer-valued operands: the f32 result must equal the reference bit for bit, and the bf16-stored
    form its rounding."""
    import random
    rng = random.Random(77)
    cases = [(1, 1, 64), (257, 255, 128), (4607, 3071, 192), (5000, 4097, 64), (33, 8192, 1024), (2048, 2048, 4096)]
    cases += [(rng.randint(1, 5000), rng.randint(1, 6500), 64 * rng.randint(1, 40)) for _ in range(14)]
    for M, N, K in cases:
        g = torch.Generator(device="cuda").manual_seed(M * 31 + N * 7 + K)
        a = torch.randint(-2, 3, (M, K), generator=g, device="cuda").to(torch.bfloat16)
        w = torch.randint(-2, 3, (N, K), generator=g, device="cuda").to(torch.bfloat16)
        ref = a.float() @ w.float().t()
        assert torch.equal(ctx.op_gemm_bf16(a, w), ref), (M, N, K)
        out16 = ctx.op_gemm_bf16_out_bf16(a, w)
        ctx.sync()
        assert torch.equal(out16, ref.to(torch.bfloat16)), (M, N, K)


@pytest.mark.gpu
@pytest.mark.parametrize("doff", [False, True])
@pytest.mark.parametrize("S,heads", [(64, 1), (200, 2), (448, 3), (1153, 2), (4608, 2)])
def test_attention_e4m3_qk(ctx, S, heads, doff):
    """The fp8 models' attention (attention64b.hip, QK8; flux2mi_op_attention_qk8): Q K^T on e4m3 copies q8 = q 2^7, k8 = k 2^4 through ONE
    block-scaled MFMA 16x16x128 per 16 x 16 score block, the powers of two undone by its scale operands; softmax, P and P V as in the bf16
    form.  Against the oracle's softmax on the DE-QUANTISED q8 / k8 (the kernel sees exactly those values: rel-L2 <= 1e-2 like the bf16
    kernel), 1 .. 72 KV tiles, ragged key counts (clamped e4m3 K rows + cleared P slots), both softmax-offset forms, bit-reproducible."""
    import math
    g = torch.Generator().manual_seed(S * 11 + heads)
    d = heads * 128
    x = torch.randn((S, 3 * d), generator=g)
    q = x[:, :d].view(S, heads, 128)
    k = x[:, d:2 * d].view(S, heads, 128)
    q = q / q.pow(2).mean(-1, keepdim=True).sqrt() * (3.0 if doff else 1.0)
    k = k / k.pow(2).mean(-1, keepdim=True).sqrt()
    k[S // 2] *= 1.5
    c = math.log2(math.e) / math.sqrt(128.0)
    qe, ke = (5 if doff else 7), 4
    q8 = ((q * c).reshape(S, d) * 2.0 ** qe).clamp(-448, 448).to(torch.float8_e4m3fn)
    k8 = (k.reshape(S, d) * 2.0 ** ke).clamp(-448, 448).to(torch.float8_e4m3fn)
    v = x[:, 2 * d:].to(torch.bfloat16)
    bound = 0.0 if doff else 1.5 * 128.0 * c
    out = ctx.op_attention_qk8(q8.cuda(), k8.cuda(), v.cuda(), heads, qe, ke, bound).float().cpu()
    hv = lambda t: t.reshape(1, S, heads, 128).permute(0, 2, 1, 3)
    ref = R.sdpa(hv(q8.float() / 2.0 ** qe / c), hv(k8.float() / 2.0 ** ke), hv(v.float())).permute(0, 2, 1, 3).reshape(S, d)
    assert torch.isfinite(out).all()
    assert rel_l2(out, ref) < 1e-2, rel_l2(out, ref)
    assert torch.equal(out, ctx.op_attention_qk8(q8.cuda(), k8.cuda(), v.cuda(), heads, qe, ke, bound).float().cpu())


@pytest.mark.gpu
@pytest.mark.parametrize("S,heads", [(448, 2), (1153, 3), (4608, 2)])
def test_attention_e4m3_qk_block_scaled_output(ctx, S, heads):
    """The output form proj / linear2 of an fp8 model consume (attention64b.hip, AttnBlocksOut): e4m3 bytes + one power-of-two scale per
    16 queries x one head's 128 columns.  De-quantised it is the bf16 output of the same call to e4m3 precision; every scale byte is the
    exact exponent of its block's maximum (checked against the oracle's fp8_block_exponents on the bf16 output: equal or, where the
    f32 maximum and its bf16 rounding straddle a power-of-two boundary, one apart); ragged last row block; bit-reproducible."""
    import math
    g = torch.Generator().manual_seed(S + heads)
    d = heads * 128
    x = torch.randn((S, 3 * d), generator=g)
    q = x[:, :d].view(S, heads, 128); k = x[:, d:2 * d].view(S, heads, 128)
    q = q / q.pow(2).mean(-1, keepdim=True).sqrt(); k = k / k.pow(2).mean(-1, keepdim=True).sqrt()
    c = math.log2(math.e) / math.sqrt(128.0)
    q8 = ((q * c).reshape(S, d) * 2.0 ** 7).clamp(-448, 448).to(torch.float8_e4m3fn)
    k8 = (k.reshape(S, d) * 2.0 ** 4).clamp(-448, 448).to(torch.float8_e4m3fn)
    vv = x[:, 2 * d:] * torch.exp2(torch.randint(-6, 7, (1, heads, 1), generator=g).float()).expand(S, heads, 128).reshape(S, d)   # heads of different magnitude
    v = vv.to(torch.bfloat16)
    bound = 1.5 * 128.0 * c
    ref = ctx.op_attention_qk8(q8.cuda(), k8.cuda(), v.cuda(), heads, 7, 4, bound).float().cpu()
    out8, bs = ctx.op_attention_qk8_blocks(q8.cuda(), k8.cuda(), v.cuda(), heads, 7, 4, bound)
    out8, bs = out8.cpu(), bs.cpu()
    scale = torch.exp2(bs.float() - 127.0)                                        # [heads, row blocks]
    deq = out8.view(torch.float8_e4m3fn).float().view(S, heads, 128) * scale.t().repeat_interleave(16, 0)[:S].unsqueeze(-1)
    deq = deq.reshape(S, d)
    assert torch.isfinite(deq).all()
    assert rel_l2(deq, ref) < 3e-2                                                # e4m3: 2^-4 relative per element, ~2.6e-2 rms
    blk = torch.nn.functional.pad(ref, (0, 0, 0, (-S) % 16)).view(-1, 16, heads, 128)
    e_ref = R.fp8_block_exponents(blk.abs().amax(dim=(1, 3))).t()                 # [heads, row blocks]
    diff = (127 - bs.int()) - e_ref.int()
    assert int(diff.abs().max()) <= 1 and float((diff != 0).float().mean()) < 0.05, (int(diff.abs().max()), float((diff != 0).float().mean()))
    # no element saturates and the largest of a block uses the top binade
    top = out8.view(torch.float8_e4m3fn).float().abs().view(S, heads, 128)
    assert float(top.max()) <= 448.0
    o2, b2 = ctx.op_attention_qk8_blocks(q8.cuda(), k8.cuda(), v.cuda(), heads, 7, 4, bound)
    assert torch.equal(o2.cpu(), out8) and torch.equal(b2.cpu(), bs)


@pytest.mark.gpu
def test_attention_e4m3_qk_negative_exponent(ctx):
    """Operands larger than e4m3's 448 are scaled DOWN (k8 = k 2^-3 for keys of magnitude ~600, e.g. a checkpoint whose key-norm weights
    are large): the scale operand carries 2^+3, the scores are the same as with any other split of the powers of two."""
    import math
    S, heads = 448, 2
    g = torch.Generator().manual_seed(77)
    d = heads * 128
    x = torch.randn((S, 3 * d), generator=g)
    q = x[:, :d].view(S, heads, 128)
    k = x[:, d:2 * d].view(S, heads, 128)
    q = q / q.pow(2).mean(-1, keepdim=True).sqrt() / 200.0
    k = k / k.pow(2).mean(-1, keepdim=True).sqrt() * 200.0          # |k_d| up to ~800
    c = math.log2(math.e) / math.sqrt(128.0)
    qe, ke = 14, -3
    q8 = ((q * c).reshape(S, d) * 2.0 ** qe).clamp(-448, 448).to(torch.float8_e4m3fn)
    k8 = (k.reshape(S, d) * 2.0 ** ke).clamp(-448, 448).to(torch.float8_e4m3fn)
    assert float((k.abs() * 2.0 ** ke).max()) < 448.0 and float(k.abs().max()) > 448.0
    v = x[:, 2 * d:].to(torch.bfloat16)
    out = ctx.op_attention_qk8(q8.cuda(), k8.cuda(), v.cuda(), heads, qe, ke, 0.0).float().cpu()
    hv = lambda t: t.reshape(1, S, heads, 128).permute(0, 2, 1, 3)
    ref = R.sdpa(hv(q8.float() / 2.0 ** qe / c), hv(k8.float() / 2.0 ** ke), hv(v.float())).permute(0, 2, 1, 3).reshape(S, d)
    assert torch.isfinite(out).all()
    assert rel_l2(out, ref) < 1e-2, rel_l2(out, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("S,heads,qk8,blocks", [(4608, 32, False, False), (4608, 48, False, False), (4600, 32, False, False), (8500, 8, False, False),
                                                (4608, 32, True, False), (4600, 32, True, True), (4608, 24, False, False), (4608, 24, True, False)])
def test_attention_kv_split_of_the_partial_round(ctx, S, heads, qk8, blocks):
    """attention64b.hip, AttnSplit: when the (head, 256-query block) items leave a partial last round of at most half the CUs — Klein-9B's
    32 heads x 18 = 576 items on 256 CUs (2.25 rounds paid as 3), Dev's 864 — every item of that round is cut into f equal key ranges, the
    pieces leave raw O / row sums and a combine kernel adds them in part order and stores through the attention's own epilogue (bf16, or the
    block-scaled e4m3 of the fp8 models).  With the fixed zero softmax offset the partial sums of disjoint key ranges just add: the result
    against the oracle like every other attention (<= 1e-2), against the oracle with the kernel's own bf16 P (device_roundings) to 2e-4
    (measured 1.1e-4, split or not; the e4m3 Q K^T form 8.7e-4, split or not: the block-scaled MFMA sums its 128 products to ~1e-4
    relative, not to an f32 ulp, which moves p = 2^s by that much at |s| ~ 10), bit-reproducible; ragged last tile inside the last piece (S 4600), f = 8 (S 8500 x 8 heads), and Klein-4B's 432 items (tail 176 > half
    the CUs: NOT split, same checks).  flux2mi_op_attention_split_info reports the plan."""
    import math
    f, n_cu = ctx.op_attention_split_info(S, heads)
    items = ((S + 255) // 256) * heads
    if n_cu == 256:
        assert f == {(4608, 32): 4, (4608, 48): 2, (4600, 32): 4, (8500, 8): 8, (4608, 24): 1}[(S, heads)], (f, items)
    g = torch.Generator().manual_seed(S * 7 + heads)
    d = heads * 128
    c = math.log2(math.e) / math.sqrt(128.0)
    x = torch.randn((S, 3 * d), generator=g)
    q = x[:, :d].view(S, heads, 128); k = x[:, d:2 * d].view(S, heads, 128)
    q = q / q.pow(2).mean(-1, keepdim=True).sqrt(); k = k / k.pow(2).mean(-1, keepdim=True).sqrt()
    v = (x[:, 2 * d:] * (1.0 + torch.arange(S).float().unsqueeze(1) / S)).to(torch.bfloat16)     # values that depend on the key index: a mis-placed piece shows
    hv = lambda t: t.reshape(1, S, heads, 128).permute(0, 2, 1, 3)
    bound = 1.5 * 128.0 * c
    if qk8:
        q8 = ((q * c).reshape(S, d) * 2.0 ** 7).clamp(-448, 448).to(torch.float8_e4m3fn)
        k8 = (k.reshape(S, d) * 2.0 ** 4).clamp(-448, 448).to(torch.float8_e4m3fn)
        qd, kd = hv(q8.float() / 2.0 ** 7 / c), hv(k8.float() / 2.0 ** 4)
        run = lambda: ctx.op_attention_qk8(q8.cuda(), k8.cuda(), v.cuda(), heads, 7, 4, bound).float().cpu()
    else:
        qkv = torch.cat([(q * c).reshape(S, d), k.reshape(S, d), v.float()], dim=1).to(torch.bfloat16)
        qs, ks, _ = qkv.float().split(d, dim=1)
        qd, kd = hv(qs / c), hv(ks)
        run = lambda: ctx.op_attention_prescaled(qkv.cuda(), heads, bound).float().cpu()
    out = run()
    with R.device_roundings():
        ref16 = R.sdpa(qd, kd, hv(v.float())).permute(0, 2, 1, 3).reshape(S, d)
    ref32 = torch.empty_like(ref16)
    for h in range(heads):       # per head: the S x S f32 intermediates of 32 ... 48 heads at once do not fit comfortably
        ref32[:, h * 128:(h + 1) * 128] = R.sdpa(qd[:, h:h + 1], kd[:, h:h + 1], hv(v.float())[:, h:h + 1])[0, 0]
    assert torch.isfinite(out).all()
    e32, e16 = rel_l2(out, ref32), rel_l2(out, ref16.to(torch.bfloat16).float())
    print(f"attention S {S} x {heads} heads ({items} items on {n_cu} CUs, {f} pieces per tail item{', e4m3 QK' if qk8 else ''}): vs f32 oracle {e32:.2e}, vs bf16-P oracle {e16:.2e}")
    assert e32 < 1e-2, e32
    assert e16 < (2e-3 if qk8 else 2e-4), e16
    assert torch.equal(out, run())
    if blocks:
        out8, bs = ctx.op_attention_qk8_blocks(q8.cuda(), k8.cuda(), v.cuda(), heads, 7, 4, bound)
        scale = torch.exp2(bs.cpu().float() - 127.0)
        deq = out8.cpu().view(torch.float8_e4m3fn).float().view(S, heads, 128) * scale.t().repeat_interleave(16, 0)[:S].unsqueeze(-1)
        assert rel_l2(deq.reshape(S, d), out) < 3e-2
        o2, b2 = ctx.op_attention_qk8_blocks(q8.cuda(), k8.cuda(), v.cuda(), heads, 7, 4, bound)
        assert torch.equal(o2.cpu(), out8.cpu()) and torch.equal(b2.cpu(), bs.cpu())
