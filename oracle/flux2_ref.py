"""CPU oracle for the Flux.2 denoise + VAE-decode hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product path (the package
`flux-2-swift-mlx_amd/`, `libflux2mi.so`) may import, call or link this file;
only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` use it, and only as the checker.

What it is: a torch-CPU float32, eager, functional restatement of the Swift
reference (VincentGourbin/flux-2-swift-mlx @ v2.4.0), one function per row of
SURVEY.md §8(a).  Every function cites the reference file:line it follows
(paths relative to the reference's `Sources/`).

PARITY STATUS: **parity unpinned** for the floating-point tensors (DiT forward,
SDPA, RoPE values, VAE decode, seeded noise).  The reference arithmetic lives in
the un-vendored dependency `ml-explore/mlx-swift` exact 0.31.6 (mlx core
0.31.1, `Package.swift:23`), the reference is Swift/Metal and cannot be built
or imported in this environment, and its own XCTest suite holds no numeric
golden vector for this path.  What IS pinned: the
structural/known-answer facts the reference's tests assert (scheduler counts
and end points, strength slicing, custom-sigma terminal append, position-ID
layout, pack/unpack shapes, dimension rounding, mask packing sums) — see
`tests/test_oracle_structural.py`, which cites each XCTest it mirrors.
Exception: the TEXT-ENCODER restatements at the end of this file (Qwen3 for Klein, Mistral for Dev)
ARE pinned against an independent implementation that runs here, HuggingFace transformers 5.15
(`Qwen3Model`, `MistralModel`): `tests/golden/{qwen3,mistral}_tiny.npz` made by
`tests/golden/make_golden_qwen3.py`, checked in `tests/test_textenc_oracle.py` (max abs diff 3e-6).

Numerics: the reference runs fp16 weights x fp32 activations.  The
oracle runs everything in fp32 on the weights it is handed (tests hand it
bf16-rounded values so the GPU sees identical weights).
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
f32 = np.float32

# ---------------------------------------------------------------------------
# Model presets (Flux2Core/Configuration/Flux2Config.swift:258-329,
# Flux2Core/Configuration/VAEConfig.swift:44-81)
# ---------------------------------------------------------------------------

DIT_PRESETS = {
    #            d     H   nd  ns   J     guidance
    "klein-4b": dict(dim=3072, heads=24, n_double=5, n_single=20, joint_dim=7680, guidance=False),
    "klein-9b": dict(dim=4096, heads=32, n_double=8, n_single=24, joint_dim=12288, guidance=False),
    "dev": dict(dim=6144, heads=48, n_double=8, n_single=48, joint_dim=15360, guidance=True),
    # tiny config for kernel-level parity (head_dim must stay 128)
    "tiny": dict(dim=256, heads=2, n_double=1, n_single=2, joint_dim=384, guidance=False),
    "tiny-g": dict(dim=256, heads=2, n_double=1, n_single=1, joint_dim=384, guidance=True),
}
HEAD_DIM = 128
ROPE_AXES = (32, 32, 32, 32)
ROPE_THETA = 2000.0
IN_CHANNELS = 128

VAE_PRESETS = {
    "standard": dict(channels=(128, 256, 512, 512), latent=32, groups=32, layers_per_block=2),
    "small": dict(channels=(96, 192, 384, 384), latent=32, groups=32, layers_per_block=2),
    "tiny": dict(channels=(32, 64, 64, 64), latent=32, groups=32, layers_per_block=2),
}


# ---------------------------------------------------------------------------
# a4  scheduler — Flux2Core/Scheduler/FlowMatchEulerScheduler.swift
# ---------------------------------------------------------------------------

def compute_empirical_mu(image_seq_len: int, num_steps: int) -> np.float32:
    """FlowMatchEulerScheduler.swift:9-28 (all arithmetic in Swift `Float` = f32)."""
    a1, b1 = f32(8.73809524e-05), f32(1.89833333)
    a2, b2 = f32(0.00016927), f32(0.45666666)
    n = f32(image_seq_len)
    if image_seq_len > 4300:
        return f32(a2 * n + b2)
    m_200 = f32(a2 * n + b2)
    m_10 = f32(a1 * n + b1)
    a = f32(f32(m_200 - m_10) / f32(190.0))
    b = f32(m_200 - f32(f32(200.0) * a))
    return f32(f32(a * f32(num_steps)) + b)


def time_shift(mu: np.float32, sigma: np.float32, t: np.float32) -> np.float32:
    """FlowMatchEulerScheduler.swift:123-128: exp(mu) / (exp(mu) + (1/t - 1)^sigma)."""
    exp_mu = f32(np.exp(f32(mu)))
    base = f32(f32(1.0) / f32(t) - f32(1.0))
    den = f32(exp_mu + f32(np.power(base, f32(sigma))))
    return f32(exp_mu / den)


def set_timesteps(num_inference_steps: int, image_seq_len: Optional[int] = None,
                  strength: float = 1.0) -> Tuple[List[np.float32], int]:
    """FlowMatchEulerScheduler.swift:65-115.  Returns (sigmas, tStart)."""
    seq = 4096 if image_seq_len is None else image_seq_len
    mu = compute_empirical_mu(seq, num_inference_steps)
    all_sigmas: List[np.float32] = []
    for i in range(num_inference_steps):
        s = f32(f32(1.0) - f32(i) / f32(num_inference_steps))
        all_sigmas.append(time_shift(mu, f32(1.0), s))
    all_sigmas.append(f32(0.0))
    clamped = max(f32(0.01), min(f32(1.0), f32(strength)))
    init_idx = num_inference_steps - int(f32(f32(num_inference_steps) * f32(clamped)))
    t_start = max(0, init_idx)
    return all_sigmas[t_start:], t_start


def set_custom_sigmas(custom: Sequence[float]) -> List[np.float32]:
    """FlowMatchEulerScheduler.swift:236-260 (terminal 0 appended unless already 0)."""
    if len(custom) == 0:
        raise ValueError("empty custom sigmas are ignored by the reference")
    out = [f32(x) for x in custom]
    if out[-1] != f32(0.0):
        out.append(f32(0.0))
    return out


def scheduler_step(sample: Tensor, model_output: Tensor, sigma: float, sigma_next: float) -> Tensor:
    """a5 — FlowMatchEulerScheduler.swift:136-156: x + (sigma_next - sigma) * v."""
    dt = f32(f32(sigma_next) - f32(sigma))
    return sample + float(dt) * model_output


def scale_noise(sample: Tensor, sigma: float, noise: Tensor) -> Tensor:
    """a6 — FlowMatchEulerScheduler.swift:195-204: (1 - sigma)*sample + sigma*noise."""
    s = float(f32(sigma))
    return (1.0 - s) * sample + s * noise


def cfg_combine(v_cond: Tensor, v_uncond: Tensor, guidance: float) -> Tensor:
    """a7 — Flux2Core/Pipeline/Flux2Pipeline.swift:1955-1973."""
    return v_uncond + float(f32(guidance)) * (v_cond - v_uncond)


def repaint_blend(latents: Tensor, image_latents: Tensor, blend_noise: Tensor, mask: Tensor,
                  sigma_next: float) -> Tensor:
    """a8 — Flux2Chains/Flux2MaskedInpaintingChain.swift:399-403.
    mask is [1,S,1] (1 = regenerate, 0 = keep source)."""
    s = float(f32(sigma_next))
    original_noised = (1.0 - s) * image_latents + s * blend_noise
    return (1.0 - mask) * original_noised + mask * latents


# ---------------------------------------------------------------------------
# a1-a3, a21, a25  latent utils — Flux2Core/Pipeline/LatentUtils.swift
# ---------------------------------------------------------------------------

def validate_dimensions(height: int, width: int, patch: int = 2) -> Tuple[int, int]:
    """LatentUtils.swift:382-394 — round UP to a multiple of 16."""
    f = 8 * patch
    return ((height + f - 1) // f) * f, ((width + f - 1) // f) * f


def resolve_chain_dimensions(width: int, height: int, max_pixels: int = 1024 * 1024) -> Tuple[int, int]:
    """Flux2Core/Pipeline/Flux2Pipeline+ChainHelpers.swift:112-129 — scale down to
    max_pixels when exceeded (Double sqrt, Int truncation), then floor to a
    multiple of 32, min 32.  Returns (height, width) like the reference."""
    w, h = width, height
    if w * h > max_pixels:
        scale = math.sqrt(float(max_pixels) / float(w * h))
        w, h = int(float(w) * scale), int(float(h) * scale)
    w = max(32, (w // 32) * 32)
    h = max(32, (h // 32) * 32)
    return h, w


def pack_patchified_to_sequence(p: Tensor) -> Tensor:
    """LatentUtils.swift:76-86: [B,C,H,W] -> [B,H*W,C]."""
    B, C, H, W = p.shape
    return p.permute(0, 2, 3, 1).reshape(B, H * W, C)


def unpack_sequence_to_patchified(seq: Tensor, height: int, width: int) -> Tensor:
    """LatentUtils.swift:95-110: [B,S,C] -> [B,C,H/16,W/16]."""
    B, _, C = seq.shape
    return seq.reshape(B, height // 16, width // 16, C).permute(0, 3, 1, 2)


def unpatchify_latents(p: Tensor, latent_channels: int = 32, patch: int = 2) -> Tensor:
    """LatentUtils.swift:119-142: [B,C*p*p,H,W] -> [B,C,H*p,W*p], channel = c*4+py*2+px."""
    B, _, H, W = p.shape
    x = p.reshape(B, latent_channels, patch, patch, H, W).permute(0, 1, 4, 2, 5, 3)
    return x.reshape(B, latent_channels, H * patch, W * patch)


def pack_latents_to_patchified(z: Tensor, patch: int = 2) -> Tensor:
    """LatentUtils.swift:186-212: [B,32,H,W] -> [B,128,H/2,W/2]."""
    B, C, H, W = z.shape
    x = z.reshape(B, C, H // patch, patch, W // patch, patch).permute(0, 2, 4, 1, 3, 5)
    x = x.reshape(B, H // patch, W // patch, C * patch * patch)
    return x.permute(0, 3, 1, 2)


def image_position_ids(height: int, width: int, patch: int = 2) -> Tensor:
    """LatentUtils.swift:256-286: int32 [S,4] = (0, y, x, 0), row-major."""
    nh, nw = height // 8 // patch, width // 8 // patch
    ys = torch.arange(nh, dtype=torch.int32).view(nh, 1).expand(nh, nw).reshape(-1)
    xs = torch.arange(nw, dtype=torch.int32).view(1, nw).expand(nh, nw).reshape(-1)
    z = torch.zeros(nh * nw, dtype=torch.int32)
    return torch.stack([z, ys, xs, z], dim=1)


def text_position_ids(length: int) -> Tensor:
    """LatentUtils.swift:291-298: (0,0,0,l)."""
    z = torch.zeros(length, dtype=torch.int32)
    return torch.stack([z, z, z, torch.arange(length, dtype=torch.int32)], dim=1)


def reference_position_ids(latent_heights: Sequence[int], latent_widths: Sequence[int],
                           scale: int = 10) -> Tensor:
    """LatentUtils.swift:324-346: reference image i gets T = scale + scale*i."""
    rows = []
    for i, (h, w) in enumerate(zip(latent_heights, latent_widths)):
        t = scale + scale * i
        for y in range(h):
            for x in range(w):
                rows.append((t, y, x, 0))
    return torch.tensor(rows, dtype=torch.int32).reshape(-1, 4)


def normalize_latents_bn(x: Tensor, mean: Tensor, var: Tensor, eps: float = 1e-4) -> Tensor:
    """LatentUtils.swift:460-473 (NCHW, 128 patchified channels, eps 1e-4)."""
    C = mean.shape[0]
    return (x - mean.view(1, C, 1, 1)) / torch.sqrt(var.view(1, C, 1, 1) + eps)


def denormalize_latents_bn(x: Tensor, mean: Tensor, var: Tensor, eps: float = 1e-4) -> Tensor:
    """a21 — LatentUtils.swift:483-496."""
    C = mean.shape[0]
    return x * torch.sqrt(var.view(1, C, 1, 1) + eps) + mean.view(1, C, 1, 1)


def pack_mask_for_latent_blending(mask_hw: Tensor, height: int, width: int) -> Tensor:
    """Flux2Pipeline+ChainHelpers.swift:148-236: a [H,W] mask in [0,1] (1 = regenerate) ->
    [1, (H/16)*(W/16), 1], rendered to the latent grid in 8 bits.  The reference rasterises with
    CoreGraphics (closed filter); the filter used instead is restated and pinned in
    oracle/imaging_ref.py (`pack_mask_for_latent_blending`), which this delegates to."""
    from . import imaging_ref
    m8 = torch.round(mask_hw.float().clamp(0, 1) * 255.0).to(torch.uint8).numpy()
    return torch.from_numpy(imaging_ref.pack_mask_for_latent_blending(m8, height, width))


# ---------------------------------------------------------------------------
# a18 RoPE tables — Flux2Core/Transformer/Flux2RoPE.swift:129-169
# ---------------------------------------------------------------------------

def rope_tables(ids: Tensor, axes: Sequence[int] = ROPE_AXES, theta: float = ROPE_THETA) -> Tuple[Tensor, Tensor]:
    """cos/sin [S, sum(axes)] f32; per axis cos(pos * theta^(-2j/dim)) repeated twice
    (interleaved), axes concatenated.  Computed from the IDs (the reference's
    cache-by-length quirk, Flux2RoPE.swift:65-70, is deliberately not reproduced)."""
    cos_parts, sin_parts = [], []
    for a, dim in enumerate(axes):
        pos = ids[:, a].to(torch.float32).unsqueeze(1)
        freq_seq = torch.arange(0, dim, 2, dtype=torch.float32)
        inv_freq = 1.0 / torch.pow(torch.tensor(theta, dtype=torch.float32), freq_seq / float(dim))
        freqs = pos * inv_freq.unsqueeze(0)
        cos_parts.append(torch.cos(freqs).repeat_interleave(2, dim=1))
        sin_parts.append(torch.sin(freqs).repeat_interleave(2, dim=1))
    return torch.cat(cos_parts, dim=1), torch.cat(sin_parts, dim=1)


def rotate_half(x: Tensor) -> Tensor:
    """Flux2Attention.swift:442-461: pairs (2i,2i+1) -> (-x1, x0)."""
    xr = x.reshape(*x.shape[:-1], x.shape[-1] // 2, 2)
    return torch.stack([-xr[..., 1], xr[..., 0]], dim=-1).reshape(x.shape)


def apply_rope(x: Tensor, cos: Tensor, sin: Tensor) -> Tensor:
    """Flux2Attention.swift:212-229: x*cos + rotate_half(x)*sin, x [B,H,S,D], cos/sin [S,D]."""
    return x * cos[None, None] + rotate_half(x) * sin[None, None]


# ---------------------------------------------------------------------------
# a11 timestep embedding — Flux2Core/Transformer/Flux2Embeddings.swift:9-142
# ---------------------------------------------------------------------------

def sinusoid_embedding(t: Tensor, num_channels: int = 256) -> Tensor:
    """Timesteps(flipSinToCos: true, downscaleFreqShift: 0): [cos | sin]."""
    half = num_channels // 2
    if _DEV_ROUND:
        # the HIP path evaluates the embedding in f64 and rounds once (elementwise.hip, sinusoid_kernel): at arguments up to 1000 rad
        # an f32 evaluation carries ~4e-5 of argument error that a timestep MLP amplifies; the store-point emulation follows the device
        e64 = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float64) / float(half))
        a64 = t.to(torch.float32).double().unsqueeze(-1) * e64.unsqueeze(0)
        return torch.cat([torch.cos(a64), torch.sin(a64)], dim=-1).float()
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32) / float(half)
    emb = torch.exp(exponent)
    arg = t.to(torch.float32).unsqueeze(-1) * emb.unsqueeze(0)
    return torch.cat([torch.cos(arg), torch.sin(arg)], dim=-1)


def linear(x: Tensor, w: Tensor, b: Optional[Tensor] = None) -> Tensor:
    return F.linear(x, w.float(), None if b is None else b.float())


# ---- fp8 emulation of the product's weight_format 1 (include/flux2mi.h) ---------------------
# NOT a restatement of the reference: its quantised variants are MLX affine group-wise int8/int4,
# weight-only (Configuration/QuantizationConfig.swift:40-73, Flux2Pipeline.swift:573).  The
# MI355X path uses OCP e4m3 for weights (scale per output channel) AND activations (scale per
# token) of the block GEMMs; this emulates exactly that so the fp8 parity test has a checker.
_FP8_BLOCK_GEMMS = False
# plain forwards of an fp8 model whose text length is a multiple of 64 (dit.cpp, `bsa`): the activations of proj / mlp_out / linear2 are
# quantised in 16-row x 128-column blocks with power-of-two scales, straight from the producing epilogue's f32 values (set by dit_forward)
_FP8_ACT_BLOCKS = False
E4M3_MAX = 448.0


# ---- emulation of the MI355X path's bf16 STORE POINTS — NOT a restatement of the reference, which keeps f32
# activations.  Off: the oracle is the f32 restatement every tolerance is stated against.  On: every tensor the HIP path
# keeps in bf16 between kernels is rounded to bf16 at the same place — the packed latents in front of img_in, `xn` (LayerNorm +
# modulate output), q (with the score scale log2(e)/sqrt(128) folded in, as the epilogue stores it), k, v, the softmax numerators P
# in front of P V (denominator = the sum of the ROUNDED numerators: the kernel sums them on the matrix pipe), the attention output
# and the SwiGLU hidden (`comb`; in fp8 mode these two are quantised to e4m3 blocks straight from f32 instead).  What is left between
# the two sides is f32 summation order and the hardware's exp2 / sin / cos / rsqrt — small against ONE e4m3 step, which is what
# makes the fp8 model-level comparison sharp (an e4m3 rounding flips wherever the two sides differ by more than that).
_DEV_ROUND = False


def _r16(x: Tensor) -> Tensor:
    return x.to(torch.bfloat16).float() if _DEV_ROUND else x


class device_roundings:
    """Context manager: round to bf16 where the HIP path stores bf16 (see above)."""

    def __enter__(self):
        global _DEV_ROUND
        self._old, _DEV_ROUND = _DEV_ROUND, True

    def __exit__(self, *exc):
        global _DEV_ROUND
        _DEV_ROUND = self._old


class fp8_block_gemms:
    """Context manager: block GEMMs (qkv/proj/mlp/linear1/linear2) run on fake-quantised operands.  `device_roundings=True` adds the
    bf16 store points of the HIP path on top (class device_roundings)."""

    def __init__(self, device_roundings: bool = False):
        self._dev = device_roundings

    def __enter__(self):
        global _FP8_BLOCK_GEMMS, _DEV_ROUND
        self._old, _FP8_BLOCK_GEMMS = _FP8_BLOCK_GEMMS, True
        self._old_dev, _DEV_ROUND = _DEV_ROUND, _DEV_ROUND or self._dev

    def __exit__(self, *exc):
        global _FP8_BLOCK_GEMMS, _DEV_ROUND
        _FP8_BLOCK_GEMMS = self._old
        _DEV_ROUND = self._old_dev


def fake_quant_e4m3_rows(x: Tensor) -> Tensor:
    """Per-row (last dim) dynamic quantisation to OCP e4m3: scale = amax/448, round-to-nearest-even,
    saturating; returns the de-quantised f32 values."""
    amax = x.float().abs().amax(dim=-1, keepdim=True)
    scale = amax.clamp_min(1e-12) * (1.0 / E4M3_MAX)      # same f32 operation order as norm.hip
    inv = 1.0 / scale
    q = (x.float() * inv).clamp(-E4M3_MAX, E4M3_MAX).to(torch.float8_e4m3fn).float()
    return q * scale


def fp8_block_exponents(amax: Tensor) -> Tensor:
    """dev.h fp8_block_exp: the largest integer e with amax 2^e <= 448 (= 0.875 x 2^9), clamped to [-60, 60] (amax = 0: 60), from the
    float's mantissa / exponent fields — exact, so that the device and this restatement agree on every block."""
    m, x = torch.frexp(amax.float())
    e = torch.where(m <= 0.875, 9 - x, 8 - x).clamp(-60, 60)
    return torch.where(amax > 0, e, torch.full_like(e, 60))


def fake_quant_e4m3_blocks(x: Tensor, mag: Optional[Tensor] = None) -> Tensor:
    """Block quantisation of an activation [..., S, K] to OCP e4m3 (the MI355X path's GemmProblem::a_bscale — NOT a restatement of the
    reference): blocks of 16 consecutive rows x 128 consecutive columns share one power-of-two scale 2^-e (fp8_block_exponents of the
    block's absolute maximum, or of the maximum of `mag` >= |x| where the producer bounds it instead: the SwiGLU epilogue takes
    |gate x value|); rows beyond S count as zeros.  Returns the de-quantised f32 values."""
    xf = x.float()
    S, K = xf.shape[-2], xf.shape[-1]
    Sp = (S + 15) // 16 * 16
    blk = lambda t: F.pad(t, (0, 0, 0, Sp - S)).reshape(*xf.shape[:-2], Sp // 16, 16, K // 128, 128)
    xp = blk(xf)
    amax = (xp if mag is None else blk(mag.float())).abs().amax(dim=(-3, -1), keepdim=True)
    mult = torch.exp2(fp8_block_exponents(amax).float())
    q = (xp * mult).clamp(-E4M3_MAX, E4M3_MAX).to(torch.float8_e4m3fn).float() / mult
    return q.reshape(*xf.shape[:-2], Sp, K)[..., :S, :]


def fp8_qk(q: Tensor, k: Tensor, wq_max: float, wk_max: float) -> Tuple[Tensor, Tensor]:
    """fp8 mode, plain forwards only (the MI355X path's e4m3 Q K^T, attention64b.hip QK8 / dit.cpp finalize — NOT a restatement of the
    reference): q carries the score scale log2(e)/sqrt(128); q 2^qe and k 2^ke are rounded to e4m3 with the largest powers of two that
    keep the RMSNorm-bounded elements (|x_d| <= sqrt(128) max|w|) inside 448; returns the de-quantised q (score scale removed again), k."""
    if not _FP8_BLOCK_GEMMS:
        if _DEV_ROUND:      # the bf16 path stores q with the score scale folded in (gemm256s.hip, QK epilogue), and k
            c = math.log2(math.e) / math.sqrt(128.0)
            return _r16(q * c) / c, _r16(k)
        return q, k
    c = math.log2(math.e) / math.sqrt(128.0)
    r128 = math.sqrt(128.0)

    def e8(bound):
        return max(-30, min(30, int(math.floor(math.log2(448.0 / bound))))) if bound > 0 else 0
    qe, ke = e8(r128 * wq_max * c), e8(r128 * wk_max)

    def rq(x, e):
        m = float(2.0 ** e)
        return (x.float() * m).clamp(-E4M3_MAX, E4M3_MAX).to(torch.float8_e4m3fn).float() / m
    return rq(q * c, qe) / c, rq(k, ke)


def block_linear(x: Tensor, w: Tensor, blocks: bool = False, mag: Optional[Tensor] = None) -> Tensor:
    """A block GEMM.  In fp8 mode the activation is first rounded to bf16 (what the producing
    kernel stores) and both operands are fake-quantised row-wise; `blocks` (proj / mlp_out / linear2 of a plain forward, see
    _FP8_ACT_BLOCKS): the activation in 16 x 128 blocks instead, from the f32 values."""
    if not _FP8_BLOCK_GEMMS:
        return linear(x, w)
    if blocks and _FP8_ACT_BLOCKS:
        return F.linear(fake_quant_e4m3_blocks(x, mag), fake_quant_e4m3_rows(w.float()))
    xq = fake_quant_e4m3_rows(x.to(torch.bfloat16).float())
    wq = fake_quant_e4m3_rows(w.float())
    return F.linear(xq, wq)


def time_guidance_embed(P: Dict[str, Tensor], t: Tensor, g: Optional[Tensor]) -> Tensor:
    """Flux2Embeddings.swift:124-142; t/g already multiplied by 1000 by the caller
    (Flux2Transformer.swift:145-146)."""
    temb = linear(F.silu(linear(sinusoid_embedding(t), P["time_in.in_layer.weight"])),
                  P["time_in.out_layer.weight"])
    if g is not None and "guidance_in.in_layer.weight" in P:
        gemb = linear(F.silu(linear(sinusoid_embedding(g), P["guidance_in.in_layer.weight"])),
                      P["guidance_in.out_layer.weight"])
        temb = temb + gemb
    return temb


# ---------------------------------------------------------------------------
# a12-a17 blocks
# ---------------------------------------------------------------------------

def layer_norm_noaffine(x: Tensor, eps: float = 1e-6) -> Tensor:
    """a13 — LayerNorm(eps 1e-6, affine false), Flux2TransformerBlock.swift:56-61."""
    return F.layer_norm(x, (x.shape[-1],), None, None, eps)


def rms_norm(x: Tensor, w: Tensor, eps: float = 1e-6) -> Tensor:
    """Flux2Attention.swift:11-26 (MLXFast.rmsNorm over the last dim)."""
    return x * torch.rsqrt(x.pow(2).mean(dim=-1, keepdim=True) + eps) * w.float()


def modulation(temb: Tensor, w: Tensor, num_sets: int) -> List[Tuple[Tensor, Tensor, Tensor]]:
    """Flux2Modulation.swift:49-75 — linear(silu(temb)) chunked (shift, scale, gate) per set."""
    dim = temb.shape[-1]
    allp = linear(F.silu(temb), w)
    out = []
    for i in range(num_sets):
        s0 = i * dim * 3
        out.append((allp[:, s0:s0 + dim], allp[:, s0 + dim:s0 + 2 * dim], allp[:, s0 + 2 * dim:s0 + 3 * dim]))
    return out


def apply_modulation(x: Tensor, shift: Tensor, scale: Tensor) -> Tensor:
    """Flux2Modulation.swift:96-112."""
    return x * (1 + scale.unsqueeze(1)) + shift.unsqueeze(1)


def sdpa(q: Tensor, k: Tensor, v: Tensor, mask: Optional[Tensor] = None) -> Tensor:
    """softmax(QK^T/sqrt(D) [+mask]) V, explicit form (MLXFast.scaledDotProductAttention,
    Flux2Attention.swift:168-174)."""
    scale = 1.0 / math.sqrt(q.shape[-1])
    if _DEV_ROUND and mask is None:
        # attention64b.hip: scores in the log2 domain, P = exp2(score - offset) rounded to bf16 for the P V MFMA, row sums of the ROUNDED
        # P.  The kernel's offset is 0 (fixed-offset form); an integer offset is the same bf16 rounding (a power-of-two factor) without
        # the overflow of f32 exp2 on the host.  Per head, to bound the memory of the S x S intermediates.
        out = torch.empty_like(q)
        v16 = _r16(v)
        for h in range(q.shape[1]):
            s2 = torch.matmul(q[:, h] * (scale * math.log2(math.e)), k[:, h].transpose(-1, -2))
            pb = _r16(torch.exp2(s2 - torch.floor(s2.amax(dim=-1, keepdim=True))))
            out[:, h] = torch.matmul(pb, v16[:, h]) / pb.sum(dim=-1, keepdim=True)
        return out
    s = torch.matmul(q, k.transpose(-1, -2)) * scale
    if mask is not None:
        s = s + mask
    return torch.matmul(torch.softmax(s, dim=-1), v)


def _store16(x: Tensor) -> Tensor:
    """attention output / SwiGLU hidden (`comb`): bf16 in the bf16 path; in fp8 mode the consuming block_linear quantises them itself
    (e4m3 blocks straight from f32, or rows from the bf16 copy)."""
    return x if _FP8_BLOCK_GEMMS else _r16(x)


def _heads(x: Tensor, H: int) -> Tensor:
    B, S, _ = x.shape
    return x.reshape(B, S, H, HEAD_DIM).permute(0, 2, 1, 3)


def _unheads(x: Tensor) -> Tensor:
    B, H, S, D = x.shape
    return x.permute(0, 2, 1, 3).reshape(B, S, H * D)


def swiglu_ff(x: Tensor, w_in: Tensor, w_out: Tensor) -> Tensor:
    """a15 — Flux2FeedForward.swift:42-109: gate = first half, value = second half."""
    p = block_linear(x, w_in)
    gate, value = p.chunk(2, dim=-1)
    return block_linear(_store16(F.silu(gate) * value), w_out, blocks=True, mag=gate * value)


def double_block(P: Dict[str, Tensor], i: int, img: Tensor, txt: Tensor, H: int,
                 cos: Tensor, sin: Tensor, img_mod, txt_mod) -> Tuple[Tensor, Tensor]:
    """a14-a16 — Flux2TransformerBlock.swift:80-168 + Flux2Attention.swift:103-193."""
    pre = f"double_blocks.{i}."
    S_txt = txt.shape[1]
    img_n = _r16(apply_modulation(layer_norm_noaffine(img), img_mod[0][0], img_mod[0][1]))
    txt_n = _r16(apply_modulation(layer_norm_noaffine(txt), txt_mod[0][0], txt_mod[0][1]))

    def qkv(x, name):
        p = block_linear(x, P[pre + name + ".qkv.weight"])
        q, k, v = p.chunk(3, dim=-1)              # WeightLoader.swift:99-119: q | k | v
        q, k, v = _heads(q, H), _heads(k, H), _heads(v, H)
        q = rms_norm(q, P[pre + name + ".norm.query_norm.scale"])
        k = rms_norm(k, P[pre + name + ".norm.key_norm.scale"])
        return q, k, v

    q_i, k_i, v_i = qkv(img_n, "img_attn")
    q_t, k_t, v_t = qkv(txt_n, "txt_attn")
    q_i, k_i = apply_rope(q_i, cos[S_txt:], sin[S_txt:]), apply_rope(k_i, cos[S_txt:], sin[S_txt:])
    q_t, k_t = apply_rope(q_t, cos[:S_txt], sin[:S_txt]), apply_rope(k_t, cos[:S_txt], sin[:S_txt])
    q = torch.cat([q_t, q_i], dim=2)               # [txt ; img]  (Flux2Attention.swift:161-163)
    k = torch.cat([k_t, k_i], dim=2)
    v = torch.cat([v_t, v_i], dim=2)
    wmax = lambda s_: max(float(P[pre + "img_attn.norm." + s_].float().abs().max()), float(P[pre + "txt_attn.norm." + s_].float().abs().max()))
    q, k = fp8_qk(q, k, wmax("query_norm.scale"), wmax("key_norm.scale"))
    o = _store16(_unheads(sdpa(q, k, v)))
    txt_attn = block_linear(o[:, :S_txt], P[pre + "txt_attn.proj.weight"], blocks=True)
    img_attn = block_linear(o[:, S_txt:], P[pre + "img_attn.proj.weight"], blocks=True)
    img = img + img_attn * img_mod[0][2].unsqueeze(1)
    txt = txt + txt_attn * txt_mod[0][2].unsqueeze(1)

    img_n2 = _r16(apply_modulation(layer_norm_noaffine(img), img_mod[1][0], img_mod[1][1]))
    txt_n2 = _r16(apply_modulation(layer_norm_noaffine(txt), txt_mod[1][0], txt_mod[1][1]))
    img = img + swiglu_ff(img_n2, P[pre + "img_mlp.0.weight"], P[pre + "img_mlp.2.weight"]) * img_mod[1][2].unsqueeze(1)
    txt = txt + swiglu_ff(txt_n2, P[pre + "txt_mlp.0.weight"], P[pre + "txt_mlp.2.weight"]) * txt_mod[1][2].unsqueeze(1)
    return txt, img


def single_block(P: Dict[str, Tensor], i: int, hs: Tensor, H: int, cos: Tensor, sin: Tensor, mod) -> Tensor:
    """a17 — Flux2SingleBlock.swift:59-98 + Flux2ParallelAttention.swift:72-123."""
    pre = f"single_blocks.{i}."
    d = hs.shape[-1]
    x = _r16(apply_modulation(layer_norm_noaffine(hs), mod[0][0], mod[0][1]))
    p = block_linear(x, P[pre + "linear1.weight"])      # [q | k | v | gate | up]
    q, k, v = p[..., :d], p[..., d:2 * d], p[..., 2 * d:3 * d]
    mlp_hidden = (p.shape[-1] - 3 * d) // 2
    gate, up = p[..., 3 * d:3 * d + mlp_hidden], p[..., 3 * d + mlp_hidden:]
    q, k, v = _heads(q, H), _heads(k, H), _heads(v, H)
    q = apply_rope(rms_norm(q, P[pre + "norm.query_norm.scale"]), cos, sin)
    k = apply_rope(rms_norm(k, P[pre + "norm.key_norm.scale"]), cos, sin)
    q, k = fp8_qk(q, k, float(P[pre + "norm.query_norm.scale"].float().abs().max()), float(P[pre + "norm.key_norm.scale"].float().abs().max()))
    attn = _unheads(sdpa(q, k, v))
    out = block_linear(_store16(torch.cat([attn, F.silu(gate) * up], dim=-1)), P[pre + "linear2.weight"], blocks=True, mag=torch.cat([attn, gate * up], dim=-1))
    return hs + out * mod[0][2].unsqueeze(1)


def dit_forward(P: Dict[str, Tensor], cfg: dict, lat: Tensor, txt_emb: Tensor, t: Tensor,
                g: Optional[Tensor], img_ids: Tensor, txt_ids: Tensor,
                taps: Optional[dict] = None) -> Tensor:
    """a9 — Flux2Transformer2DModel.callAsFunction, Flux2Transformer.swift:123-327.
    lat [B,S_img,128], txt_emb [B,S_txt,J], t [B] in [0,1], g [B] or None.
    `taps`, if given, receives intermediate tensors (test aid)."""
    H = cfg["heads"]
    global _FP8_ACT_BLOCKS
    _FP8_ACT_BLOCKS = _FP8_BLOCK_GEMMS and txt_emb.shape[1] % 64 == 0 and cfg["dim"] % 128 == 0
    img = linear(_r16(lat.float()), P["img_in.weight"])
    txt = linear(_r16(txt_emb.float()), P["txt_in.weight"])
    temb = time_guidance_embed(P, t.float() * 1000.0, None if g is None else g.float() * 1000.0)
    cos, sin = rope_tables(torch.cat([txt_ids, img_ids], dim=0))
    img_mod = modulation(temb, P["double_stream_modulation_img.lin.weight"], 2)
    txt_mod = modulation(temb, P["double_stream_modulation_txt.lin.weight"], 2)
    if taps is not None:
        taps.update(temb=temb, cos=cos, sin=sin, img_in=img, txt_in=txt)
    for i in range(cfg["n_double"]):
        txt, img = double_block(P, i, img, txt, H, cos, sin, img_mod, txt_mod)
        if taps is not None:
            taps[f"double{i}.img"], taps[f"double{i}.txt"] = img, txt
    S_txt = txt.shape[1]
    hs = torch.cat([txt, img], dim=1)
    smod = modulation(temb, P["single_stream_modulation.lin.weight"], 1)
    for i in range(cfg["n_single"]):
        hs = single_block(P, i, hs, H, cos, sin, smod)
        if taps is not None:
            taps[f"single{i}"] = hs
    img = hs[:, S_txt:, :]
    # AdaLayerNormContinuous (Flux2Modulation.swift:125-156).  The BFL checkpoint
    # stores [shift | scale]; the reference swaps the halves at load
    # (WeightLoader.swift:182-202) and then reads (scale, shift).  Net effect on
    # the BFL-ordered weight used here: shift = first half, scale = second half.
    d = img.shape[-1]
    p = linear(F.silu(temb), P["final_layer.adaLN_modulation.1.weight"])
    shift, scale = p[:, :d], p[:, d:]
    img = _r16(apply_modulation(layer_norm_noaffine(img), shift, scale))
    return linear(img, P["final_layer.linear.weight"])


# ---------------------------------------------------------------------------
# a22-a23 VAE decoder — Flux2Core/VAE/{AutoencoderKL,VAEDecoder,ResnetBlock}.swift
# Weights in diffusers naming, conv weights OIHW as in the checkpoint file
# (the reference transposes to OHWI for MLX's NHWC conv: WeightLoader.swift:486-560;
# the arithmetic is identical).
# ---------------------------------------------------------------------------

# ---- KV-cached reference conditioning (klein-9b-kv) --------------------------------------------
def _kv_mask(s_txt: int, s_ref: int, s_out: int) -> Tensor:
    """Flux2Attention.swift:407-425: additive mask over [txt | ref | out]; reference queries do not
    see output keys, everything else is allowed."""
    S = s_txt + s_ref + s_out
    m = torch.zeros((S, S))
    m[s_txt:s_txt + s_ref, s_txt + s_ref:] = float("-inf")
    return m


def dit_forward_kv(P: Dict[str, Tensor], cfg: dict, lat: Tensor, txt_emb: Tensor, t: Tensor, g: Optional[Tensor],
                   img_ids: Tensor, txt_ids: Tensor, ref_lat: Optional[Tensor] = None, ref_ids: Optional[Tensor] = None,
                   cache: Optional[list] = None) -> Tuple[Tensor, list]:
    """Flux2Transformer.swift:346-446 (ref_lat given: extraction, returns the cache) and :459-546
    (cache given: cached pass).  cache[layer] = (K post-RoPE, V) of the reference tokens, [1,H,S_ref,128];
    layers = double blocks then single blocks."""
    extract = ref_lat is not None
    assert extract != (cache is not None)
    global _FP8_ACT_BLOCKS
    _FP8_ACT_BLOCKS = False     # the KV passes keep the row-wise activations (dit.cpp: bsa needs a plain forward)
    H, d = cfg["heads"], cfg["dim"]
    S_txt, S_out = txt_emb.shape[1], lat.shape[1]
    img = linear(_r16(lat.float()), P["img_in.weight"])
    txt = linear(_r16(txt_emb.float()), P["txt_in.weight"])
    if extract:
        S_ref = ref_lat.shape[1]
        img = torch.cat([linear(ref_lat.float(), P["img_in.weight"]), img], dim=1)     # [ref | out]
        ids = torch.cat([txt_ids, ref_ids, img_ids], dim=0)
        mask = _kv_mask(S_txt, S_ref, S_out)
        new_cache = []
    else:
        S_ref = cache[0][0].shape[2]
        ids = torch.cat([txt_ids, img_ids], dim=0)
        mask = None
    temb = time_guidance_embed(P, t * 1000.0, None if g is None else g * 1000.0)
    cos, sin = rope_tables(ids)
    img_mod = modulation(temb, P["double_stream_modulation_img.lin.weight"], 2)
    txt_mod = modulation(temb, P["double_stream_modulation_txt.lin.weight"], 2)
    sgl_mod = modulation(temb, P["single_stream_modulation.lin.weight"], 1)

    def attend(q, k, v, layer):
        # q,k,v [1,H,S,128] over [txt | (ref) | out]
        if extract:
            new_cache.append((k[:, :, S_txt:S_txt + S_ref].clone(), v[:, :, S_txt:S_txt + S_ref].clone()))
            return sdpa(q, k, v, mask)
        ck, cv = cache[layer]
        k = torch.cat([k[:, :, :S_txt], ck, k[:, :, S_txt:]], dim=2)
        v = torch.cat([v[:, :, :S_txt], cv, v[:, :, S_txt:]], dim=2)
        return sdpa(q, k, v)

    for i in range(cfg["n_double"]):
        pre = f"double_blocks.{i}."
        img_n = apply_modulation(layer_norm_noaffine(img), img_mod[0][0], img_mod[0][1])
        txt_n = apply_modulation(layer_norm_noaffine(txt), txt_mod[0][0], txt_mod[0][1])

        def qkv(x, name, c, s_):
            p = block_linear(x, P[pre + name + ".qkv.weight"])
            q, k, v = p.chunk(3, dim=-1)
            q, k, v = _heads(q, H), _heads(k, H), _heads(v, H)
            q = apply_rope(rms_norm(q, P[pre + name + ".norm.query_norm.scale"]), c, s_)
            k = apply_rope(rms_norm(k, P[pre + name + ".norm.key_norm.scale"]), c, s_)
            return q, k, v

        q_i, k_i, v_i = qkv(img_n, "img_attn", cos[S_txt:], sin[S_txt:])
        q_t, k_t, v_t = qkv(txt_n, "txt_attn", cos[:S_txt], sin[:S_txt])
        o = _unheads(attend(torch.cat([q_t, q_i], 2), torch.cat([k_t, k_i], 2), torch.cat([v_t, v_i], 2), i))
        txt = txt + block_linear(o[:, :S_txt], P[pre + "txt_attn.proj.weight"]) * txt_mod[0][2].unsqueeze(1)
        img = img + block_linear(o[:, S_txt:], P[pre + "img_attn.proj.weight"]) * img_mod[0][2].unsqueeze(1)
        img_n2 = apply_modulation(layer_norm_noaffine(img), img_mod[1][0], img_mod[1][1])
        txt_n2 = apply_modulation(layer_norm_noaffine(txt), txt_mod[1][0], txt_mod[1][1])
        img = img + swiglu_ff(img_n2, P[pre + "img_mlp.0.weight"], P[pre + "img_mlp.2.weight"]) * img_mod[1][2].unsqueeze(1)
        txt = txt + swiglu_ff(txt_n2, P[pre + "txt_mlp.0.weight"], P[pre + "txt_mlp.2.weight"]) * txt_mod[1][2].unsqueeze(1)

    hs = torch.cat([txt, img], dim=1)
    for i in range(cfg["n_single"]):
        pre = f"single_blocks.{i}."
        x = apply_modulation(layer_norm_noaffine(hs), sgl_mod[0][0], sgl_mod[0][1])
        p = block_linear(x, P[pre + "linear1.weight"])
        q, k, v = p[..., :d], p[..., d:2 * d], p[..., 2 * d:3 * d]
        hid = (p.shape[-1] - 3 * d) // 2
        gate, up = p[..., 3 * d:3 * d + hid], p[..., 3 * d + hid:]
        q, k, v = _heads(q, H), _heads(k, H), _heads(v, H)
        q = apply_rope(rms_norm(q, P[pre + "norm.query_norm.scale"]), cos, sin)
        k = apply_rope(rms_norm(k, P[pre + "norm.key_norm.scale"]), cos, sin)
        attn = _unheads(attend(q, k, v, cfg["n_double"] + i))
        hs = hs + block_linear(torch.cat([attn, F.silu(gate) * up], dim=-1), P[pre + "linear2.weight"]) * sgl_mod[0][2].unsqueeze(1)

    out_hs = hs[:, hs.shape[1] - S_out:]
    pm = linear(F.silu(temb), P["final_layer.adaLN_modulation.1.weight"])
    shift, scale = pm.chunk(2, dim=-1)      # BFL order [shift | scale], as in dit_forward
    y = apply_modulation(layer_norm_noaffine(out_hs), shift, scale)
    return linear(y, P["final_layer.linear.weight"]), (new_cache if extract else cache)


def group_norm(x: Tensor, w: Tensor, b: Tensor, groups: int = 32, eps: float = 1e-6) -> Tensor:
    """ResnetBlock.swift:9-55 (f32 statistics over (H,W,C/G))."""
    return F.group_norm(x, groups, w.float(), b.float(), eps)


def conv2d(x: Tensor, P: Dict[str, Tensor], name: str, padding: int) -> Tensor:
    return F.conv2d(x, P[name + ".weight"].float(), P[name + ".bias"].float(), padding=padding)


def resnet_block(x: Tensor, P: Dict[str, Tensor], pre: str, groups: int) -> Tensor:
    """ResnetBlock.swift:122-187."""
    h = F.silu(group_norm(x, P[pre + "norm1.weight"], P[pre + "norm1.bias"], groups))
    h = conv2d(h, P, pre + "conv1", 1)
    h = F.silu(group_norm(h, P[pre + "norm2.weight"], P[pre + "norm2.bias"], groups))
    h = conv2d(h, P, pre + "conv2", 1)
    if pre + "conv_shortcut.weight" in P:
        x = conv2d(x, P, pre + "conv_shortcut", 0)
    return h + x


def vae_attention(x: Tensor, P: Dict[str, Tensor], pre: str, groups: int) -> Tensor:
    """ResnetBlock.swift:257-313: single head, scale 1/sqrt(C), Linear with bias."""
    B, C, Hh, Ww = x.shape
    h = group_norm(x, P[pre + "group_norm.weight"], P[pre + "group_norm.bias"], groups)
    h = h.permute(0, 2, 3, 1).reshape(B, Hh * Ww, C)
    q = linear(h, P[pre + "to_q.weight"], P[pre + "to_q.bias"])
    k = linear(h, P[pre + "to_k.weight"], P[pre + "to_k.bias"])
    v = linear(h, P[pre + "to_v.weight"], P[pre + "to_v.bias"])
    a = torch.softmax(torch.matmul(q, k.transpose(1, 2)) * (1.0 / math.sqrt(C)), dim=-1)
    o = linear(torch.matmul(a, v), P[pre + "to_out.0.weight"], P[pre + "to_out.0.bias"])
    return o.reshape(B, Hh, Ww, C).permute(0, 3, 1, 2) + x


def upsample2x(x: Tensor, P: Dict[str, Tensor], pre: str) -> Tensor:
    """ResnetBlock.swift:217-254: nearest 2x then conv3x3 pad 1."""
    x = x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)
    return conv2d(x, P, pre + "conv", 1)


def vae_decode(P: Dict[str, Tensor], vcfg: dict, z: Tensor, taps: Optional[dict] = None) -> Tensor:
    """a22 — AutoencoderKL.swift:129-143 + VAEDecoder.swift:91-121.
    z [B,32,h,w] -> [B,3,8h,8w]; no scaling factor; post_quant_conv applied."""
    groups = vcfg["groups"]
    n_up = len(vcfg["channels"])
    x = conv2d(z.float(), P, "post_quant_conv", 0)
    x = conv2d(x, P, "decoder.conv_in", 1)
    x = resnet_block(x, P, "decoder.mid_block.resnets.0.", groups)
    x = vae_attention(x, P, "decoder.mid_block.attentions.0.", groups)
    x = resnet_block(x, P, "decoder.mid_block.resnets.1.", groups)
    if taps is not None:
        taps["mid"] = x
    for i in range(n_up):
        for j in range(vcfg["layers_per_block"] + 1):
            x = resnet_block(x, P, f"decoder.up_blocks.{i}.resnets.{j}.", groups)
        if i < n_up - 1:
            x = upsample2x(x, P, f"decoder.up_blocks.{i}.upsamplers.0.")
        if taps is not None:
            taps[f"up{i}"] = x
    x = F.silu(group_norm(x, P["decoder.conv_norm_out.weight"], P["decoder.conv_norm_out.bias"], groups))
    return conv2d(x, P, "decoder.conv_out", 1)


def downsample2x(x: Tensor, P: Dict[str, Tensor], pre: str) -> Tensor:
    """ResnetBlock.swift:189-214: zero-pad bottom/right by one, conv3x3 stride 2, no other padding."""
    x = F.pad(x, (0, 1, 0, 1))
    return F.conv2d(x, P[pre + "conv.weight"].float(), P[pre + "conv.bias"].float(), stride=2)


def vae_encode(P: Dict[str, Tensor], vcfg: dict, x: Tensor) -> Tensor:
    """a23 — AutoencoderKL.swift:90-123 with samplePosterior = false + VAEEncoder.swift:88-118.
    x [B,3,H,W] in [-1,1] -> posterior mean [B,32,H/8,W/8]; quant_conv applied; no scaling factor."""
    groups = vcfg["groups"]
    n = len(vcfg["channels"])
    h = conv2d(x.float(), P, "encoder.conv_in", 1)
    for i in range(n):
        for j in range(vcfg["layers_per_block"]):
            h = resnet_block(h, P, f"encoder.down_blocks.{i}.resnets.{j}.", groups)
        if i < n - 1:
            h = downsample2x(h, P, f"encoder.down_blocks.{i}.downsamplers.0.")
    h = resnet_block(h, P, "encoder.mid_block.resnets.0.", groups)
    h = vae_attention(h, P, "encoder.mid_block.attentions.0.", groups)
    h = resnet_block(h, P, "encoder.mid_block.resnets.1.", groups)
    h = F.silu(group_norm(h, P["encoder.conv_norm_out.weight"], P["encoder.conv_norm_out.bias"], groups))
    h = conv2d(h, P, "encoder.conv_out", 1)
    h = conv2d(h, P, "quant_conv", 0)
    return h[:, :vcfg["latent"]]


def preprocess_image_for_vae(image_u8_hwc: Tensor) -> Tensor:
    """Flux2Pipeline.swift:2408-2418: u8 [H,W,3] -> f32 [1,3,H,W] = x / 127.5 - 1 (resize not restated)."""
    return (image_u8_hwc.float() / 127.5 - 1.0).permute(2, 0, 1).unsqueeze(0)


def encode_image_to_packed_sequence(P: Dict[str, Tensor], vcfg: dict, image: Tensor) -> Tensor:
    """Flux2Pipeline+ChainHelpers.swift:75-103: encode (mean) -> patchify -> BN-normalise -> pack.
    image f32 [1,3,H,W] -> [1, (H/16)(W/16), 128]."""
    z = vae_encode(P, vcfg, image)
    pt = pack_latents_to_patchified(z)
    pt = normalize_latents_bn(pt, P["bn.running_mean"].float(), P["bn.running_var"].float())
    return pack_patchified_to_sequence(pt)


def postprocess_vae_output(x: Tensor) -> Tensor:
    """a24 — Flux2Pipeline.swift:2425-2468: clip((x+1)*127.5, 0, 255) -> TRUNCATING
    u8 cast -> [H,W,3]."""
    y = torch.clamp((x + 1.0) * 127.5, 0.0, 255.0)
    return y[0].permute(1, 2, 0).to(torch.uint8)   # torch float->uint8 truncates toward zero


def latents_to_vae_input(packed: Tensor, height: int, width: int, bn_mean: Tensor, bn_var: Tensor) -> Tensor:
    """Flux2Pipeline.swift:2059-2079: unpack -> BN-denormalise (128 ch) -> unpatchify."""
    p = unpack_sequence_to_patchified(packed, height, width)
    p = denormalize_latents_bn(p, bn_mean.float(), bn_var.float())
    return unpatchify_latents(p)


# ---------------------------------------------------------------------------
# Pipeline loop — Flux2Pipeline.swift:1442-2117 (T2I) / 1490-1808 (I2I)
# ---------------------------------------------------------------------------

StepHook = Callable[[dict, Tensor], Tensor]


def generate(P: Dict[str, Tensor], cfg: dict, VP: Optional[Dict[str, Tensor]], vcfg: Optional[dict],
             txt_emb: Tensor, noise_patchified: Tensor, height: int, width: int, steps: int,
             guidance: Optional[float] = None, init_latents: Optional[Tensor] = None,
             strength: float = 1.0, on_step: Optional[StepHook] = None,
             ref_latents: Optional[Tensor] = None, ref_ids: Optional[Tensor] = None,
             neg_emb: Optional[Tensor] = None, cfg_scale: float = 1.0,
             custom_sigmas: Optional[Sequence[float]] = None, decode: bool = True) -> dict:
    """Restates generateWithResult's hot section.  Explicit inputs instead of RNG /
    text encoder: `noise_patchified` [1,128,H/16,W/16] (the reference draws noise
    in this layout, Flux2Pipeline.swift:1448, then packs it, :1860), `txt_emb`
    [1,S_txt,J].  Returns sigmas, per-step latents and (if decode) the u8 image."""
    height, width = validate_dimensions(height, width)
    packed = pack_patchified_to_sequence(noise_patchified.float())
    S_img = packed.shape[1]
    txt_ids = text_position_ids(txt_emb.shape[1])
    img_ids = image_position_ids(height, width)
    strength = max(0.01, min(1.0, strength))                  # effectiveStrength, Flux2Pipeline.swift:1208
    if strength < 1.0 and init_latents is None:
        raise ValueError("strength < 1 requires initLatents (Flux2Pipeline.swift:1210-1213)")
    use_init = init_latents is not None and strength < 1.0    # useImg2ImgInit, :1209
    if custom_sigmas is not None:
        sigmas = set_custom_sigmas(custom_sigmas)
    else:
        sigmas, _ = set_timesteps(steps, S_img, strength if use_init else 1.0)
        if use_init and len(sigmas) < 2:
            raise ValueError("zero denoising steps (Flux2Pipeline.swift:1893-1896)")
        if use_init:
            if tuple(init_latents.shape) != tuple(packed.shape):
                raise ValueError("initLatents shape mismatch (Flux2Pipeline.swift:1898-1901)")
            packed = scale_noise(init_latents.float(), sigmas[0], packed)
    all_img_ids = img_ids if ref_ids is None else torch.cat([img_ids, ref_ids], dim=0)
    g = None if (guidance is None or not cfg["guidance"]) else torch.tensor([guidance], dtype=torch.float32)
    traj = []
    n = len(sigmas) - 1
    for k in range(n):
        sigma, sigma_next = sigmas[k], sigmas[k + 1]
        t = torch.tensor([float(sigma)], dtype=torch.float32)
        inp = packed if ref_latents is None else torch.cat([packed, ref_latents.float()], dim=1)
        v = dit_forward(P, cfg, inp, txt_emb, t, g, all_img_ids, txt_ids)
        if neg_emb is not None and cfg_scale > 1.0:
            v_u = dit_forward(P, cfg, inp, neg_emb, t, g, all_img_ids, text_position_ids(neg_emb.shape[1]))
            v = cfg_combine(v, v_u, cfg_scale)
        v = v[:, :S_img]
        packed = scheduler_step(packed, v, sigma, sigma_next)
        if on_step is not None:
            ctx = dict(stepIdx=k, totalSteps=n, sigma=float(sigma), sigmaNext=float(sigma_next),
                       height=height, width=width, isI2I=ref_latents is not None)
            packed = on_step(ctx, packed)
        traj.append(packed.clone())
    out = dict(sigmas=[float(s) for s in sigmas], latents=traj, final_latents=packed)
    if decode and VP is not None:
        z = latents_to_vae_input(packed, height, width, VP["bn.running_mean"], VP["bn.running_var"])
        dec = vae_decode(VP, vcfg, z)
        out["decoded"] = dec
        out["image"] = postprocess_vae_output(dec)
    return out


# ---------------------------------------------------------------------------
# Text encoder: Qwen3 hidden states -> Klein prompt embeddings
# FluxTextEncoders/Model/Qwen3/{Qwen3Model,Qwen3DecoderLayer,Qwen3Attention,Qwen3MLP}.swift,
# Embeddings/KleinEmbeddingExtractor.swift:50-131.
# PINNED: unlike the DiT/VAE restatements this one is checked against an independent implementation
# that runs here — HuggingFace transformers 5.15 `Qwen3Model` (the checkpoint's own reference code) —
# through tests/golden/qwen3_tiny.npz (made by tests/golden/make_golden_qwen3.py).
# ---------------------------------------------------------------------------
def _rms(x: Tensor, w: Tensor, eps: float) -> Tensor:
    return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps) * w.float()


def _rope_split_half(x: Tensor, theta: float) -> Tensor:
    """MLXFast.RoPE(traditional: false) (Qwen3Attention.swift:15-37): pairs (i, i + D/2), position = index.
    x [B, H, S, D]."""
    D, S = x.shape[-1], x.shape[-2]
    inv = theta ** (-torch.arange(0, D, 2, dtype=torch.float32) / D)
    ang = torch.arange(S, dtype=torch.float32)[:, None] * inv[None, :]
    cos, sin = torch.cos(ang), torch.sin(ang)
    a, b = x[..., : D // 2], x[..., D // 2:]
    return torch.cat([a * cos - b * sin, b * cos + a * sin], dim=-1)


def qwen3_hidden_states(P: Dict[str, Tensor], tcfg: dict, ids: Tensor, valid_len: int, layers: Sequence[int],
                        pad_left: int = 0) -> Dict[int, Tensor]:
    """Qwen3Model.forwardWithHiddenStates (Qwen3Model.swift:109-174).  ids int64 [S]; returns
    {layer index (1-based, "after layer i"): [1, S, hidden]}.  Causal mask + right-padding mask
    (:176-212, pad keys get -1e9), GQA by repetition (Qwen3Attention.swift:135-147), q_norm/k_norm
    before RoPE (:110-123), SwiGLU MLP (Qwen3MLP.swift).
    With tcfg["qk_norm"] == 0 and pad_left > 0 this is the Dev path (MistralModel.swift:95-195,
    MistralAttention.swift:395-470, EmbeddingExtractor.swift:202-278): no q/k norm, real tokens at
    [pad_left, pad_left + valid_len), same additive mask; the position-dependent query scale of
    MistralAttention.swift:422-431 is exactly 1 below position 16384."""
    H, Hkv, hd = tcfg["heads"], tcfg["kv_heads"], tcfg.get("head_dim", 128)
    eps, theta = tcfg.get("rms_eps", 1e-6), tcfg.get("rope_theta", 1.0e6)
    S = ids.shape[0]
    g = lambda k: P[k] if k in P else P["model." + k]
    x = g("embed_tokens.weight").float()[ids].unsqueeze(0)
    i_idx = torch.arange(S)
    mask = torch.where(i_idx[None, :] <= i_idx[:, None], 0.0, float("-inf"))
    mask = mask + torch.where((i_idx >= pad_left) & (i_idx < pad_left + valid_len), 0.0, -1e9)[None, :]
    out = {}
    for l in range(max(layers)):
        pre = f"layers.{l}."
        h = _rms(x, g(pre + "input_layernorm.weight"), eps)
        q = linear(h, g(pre + "self_attn.q_proj.weight")).reshape(1, S, H, hd)
        k = linear(h, g(pre + "self_attn.k_proj.weight")).reshape(1, S, Hkv, hd)
        v = linear(h, g(pre + "self_attn.v_proj.weight")).reshape(1, S, Hkv, hd)
        if tcfg.get("qk_norm", 1):
            q, k = _rms(q, g(pre + "self_attn.q_norm.weight"), eps), _rms(k, g(pre + "self_attn.k_norm.weight"), eps)
        q, k = q.permute(0, 2, 1, 3), k.permute(0, 2, 1, 3)
        v = v.permute(0, 2, 1, 3)
        q, k = _rope_split_half(q, theta), _rope_split_half(k, theta)
        rep = H // Hkv
        k, v = k.repeat_interleave(rep, dim=1), v.repeat_interleave(rep, dim=1)
        a = sdpa(q, k, v, mask).permute(0, 2, 1, 3).reshape(1, S, H * hd)
        x = x + linear(a, g(pre + "self_attn.o_proj.weight"))
        h = _rms(x, g(pre + "post_attention_layernorm.weight"), eps)
        x = x + linear(F.silu(linear(h, g(pre + "mlp.gate_proj.weight"))) * linear(h, g(pre + "mlp.up_proj.weight")),
                       g(pre + "mlp.down_proj.weight"))
        if l + 1 in layers:
            out[l + 1] = x
    return out


def klein_embeddings(P: Dict[str, Tensor], tcfg: dict, ids: Tensor, valid_len: int, pad_left: int = 0) -> Tensor:
    """KleinEmbeddingExtractor.extractKleinEmbeddings steps 8-10 (pad_left 0) / EmbeddingExtractor.
    extractFluxEmbeddings steps 7-9 (Dev: pad_left = S - valid_len): hidden states after the extract
    layers, concatenated on the feature axis -> [1, S, 3*hidden]."""
    ls = tuple(tcfg.get("extract_layers", (9, 18, 27)))
    hs = qwen3_hidden_states(P, tcfg, ids, valid_len, ls, pad_left)
    return torch.cat([hs[i] for i in ls], dim=-1)
