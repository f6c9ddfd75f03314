"""Synthetic, seeded Flux.2 weights and inputs (there are no checkpoints on the
target machines; SURVEY.md §8d fixes the recipe).

Weights follow the on-disk naming the reference ingests:
  * DiT: BFL-native keys (`Flux2Core/Loading/WeightLoader.swift:99-205`), all
    Linear weights `[out, in]`, no biases.
  * VAE: diffusers keys (`WeightLoader.swift:486-560`), conv weights OIHW,
    `bn.running_mean/var` of the 128 patchified channels.
All values are rounded to bf16 so that the CPU oracle and the HIP path consume
bit-identical parameters.
"""
from __future__ import annotations

import zlib
from collections.abc import Mapping
from typing import Dict, List, Tuple

import torch

HEAD_DIM = 128

DIT_PRESETS = {
    "klein-4b": dict(dim=3072, heads=24, n_double=5, n_single=20, joint_dim=7680, guidance=False),
    "klein-9b": dict(dim=4096, heads=32, n_double=8, n_single=24, joint_dim=12288, guidance=False),
    "dev": dict(dim=6144, heads=48, n_double=8, n_single=48, joint_dim=15360, guidance=True),
    "tiny": dict(dim=256, heads=2, n_double=1, n_single=2, joint_dim=384, guidance=False),
    "tiny-g": dict(dim=256, heads=2, n_double=1, n_single=1, joint_dim=384, guidance=True),
}

VAE_PRESETS = {
    "standard": dict(channels=(128, 256, 512, 512), latent=32, groups=32, layers_per_block=2),
    "small": dict(channels=(96, 192, 384, 384), latent=32, groups=32, layers_per_block=2),
    "tiny": dict(channels=(32, 64, 64, 64), latent=32, groups=32, layers_per_block=2),
}


def dit_param_shapes(cfg: dict) -> List[Tuple[str, Tuple[int, ...], str]]:
    """(name, shape, kind) for every DiT tensor; kind in {linear, norm}."""
    d, J = cfg["dim"], cfg["joint_dim"]
    out: List[Tuple[str, Tuple[int, ...], str]] = [
        ("img_in.weight", (d, 128), "linear"),
        ("txt_in.weight", (d, J), "linear"),
        ("time_in.in_layer.weight", (d, 256), "linear"),
        ("time_in.out_layer.weight", (d, d), "linear"),
    ]
    if cfg["guidance"]:
        out += [("guidance_in.in_layer.weight", (d, 256), "linear"),
                ("guidance_in.out_layer.weight", (d, d), "linear")]
    out += [
        ("double_stream_modulation_img.lin.weight", (6 * d, d), "linear"),
        ("double_stream_modulation_txt.lin.weight", (6 * d, d), "linear"),
        ("single_stream_modulation.lin.weight", (3 * d, d), "linear"),
    ]
    for i in range(cfg["n_double"]):
        p = f"double_blocks.{i}."
        for s in ("img", "txt"):
            out += [
                (p + f"{s}_attn.qkv.weight", (3 * d, d), "linear"),
                (p + f"{s}_attn.proj.weight", (d, d), "linear"),
                (p + f"{s}_attn.norm.query_norm.scale", (HEAD_DIM,), "norm"),
                (p + f"{s}_attn.norm.key_norm.scale", (HEAD_DIM,), "norm"),
                (p + f"{s}_mlp.0.weight", (6 * d, d), "linear"),
                (p + f"{s}_mlp.2.weight", (d, 3 * d), "linear"),
            ]
    for i in range(cfg["n_single"]):
        p = f"single_blocks.{i}."
        out += [
            (p + "linear1.weight", (9 * d, d), "linear"),
            (p + "linear2.weight", (d, 4 * d), "linear"),
            (p + "norm.query_norm.scale", (HEAD_DIM,), "norm"),
            (p + "norm.key_norm.scale", (HEAD_DIM,), "norm"),
        ]
    out += [
        ("final_layer.adaLN_modulation.1.weight", (2 * d, d), "linear"),
        ("final_layer.linear.weight", (128, d), "linear"),
    ]
    return out


def vae_decoder_param_shapes(vcfg: dict, encoder: bool = False) -> List[Tuple[str, Tuple[int, ...], str]]:
    """(name, shape, kind) for the decoder side of AutoencoderKLFlux2 (+ the encoder side when
    `encoder`); kind in {conv, bias, gn_w, gn_b, linear, bn_mean, bn_var}."""
    ch = list(vcfg["channels"])
    L = vcfg["latent"]
    out: List[Tuple[str, Tuple[int, ...], str]] = []

    def conv(name, co, ci, k):
        out.append((name + ".weight", (co, ci, k, k), "conv"))
        out.append((name + ".bias", (co,), "bias"))

    def gn(name, c):
        out.append((name + ".weight", (c,), "gn_w"))
        out.append((name + ".bias", (c,), "gn_b"))

    def resnet(p, ci, co):
        gn(p + "norm1", ci)
        conv(p + "conv1", co, ci, 3)
        gn(p + "norm2", co)
        conv(p + "conv2", co, co, 3)
        if ci != co:
            conv(p + "conv_shortcut", co, ci, 1)

    conv("post_quant_conv", L, L, 1)
    mid = ch[-1]
    conv("decoder.conv_in", mid, L, 3)
    resnet("decoder.mid_block.resnets.0.", mid, mid)
    a = "decoder.mid_block.attentions.0."
    gn(a + "group_norm", mid)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        out.append((a + n + ".weight", (mid, mid), "linear"))
        out.append((a + n + ".bias", (mid,), "bias"))
    resnet("decoder.mid_block.resnets.1.", mid, mid)
    prev = mid
    rev = list(reversed(ch))
    for i, co in enumerate(rev):
        for j in range(vcfg["layers_per_block"] + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}.", prev if j == 0 else co, co)
        prev = co
        if i < len(rev) - 1:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", co, co, 3)
    gn("decoder.conv_norm_out", ch[0])
    conv("decoder.conv_out", 3, ch[0], 3)
    out.append(("bn.running_mean", (4 * L,), "bn_mean"))
    out.append(("bn.running_var", (4 * L,), "bn_var"))
    if not encoder:
        return out
    # encoder side, appended AFTER the decoder list so that decoder values for a given seed do not
    # depend on it (VAE/VAEEncoder.swift:26-85; diffusers keys, WeightLoader.swift:486-560)
    conv("encoder.conv_in", ch[0], 3, 3)
    prev = ch[0]
    for i, co in enumerate(ch):
        for j in range(vcfg["layers_per_block"]):
            resnet(f"encoder.down_blocks.{i}.resnets.{j}.", prev if j == 0 else co, co)
        prev = co
        if i < len(ch) - 1:
            conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", co, co, 3)
    resnet("encoder.mid_block.resnets.0.", mid, mid)
    a = "encoder.mid_block.attentions.0."
    gn(a + "group_norm", mid)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        out.append((a + n + ".weight", (mid, mid), "linear"))
        out.append((a + n + ".bias", (mid,), "bias"))
    resnet("encoder.mid_block.resnets.1.", mid, mid)
    gn("encoder.conv_norm_out", mid)
    conv("encoder.conv_out", 2 * L, mid, 3)
    conv("quant_conv", 2 * L, 2 * L, 1)
    return out


def _gen(device, seed: int) -> torch.Generator:
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    return g


def make_dit_weights(cfg: dict, seed: int = 1234, device="cpu", std: float = 0.02) -> Dict[str, torch.Tensor]:
    """Linear ~ N(0, std) (sigma_w = 0.02); RMSNorm scales 1 + 0.1 N(0,1).
    bf16 tensors on `device`."""
    g = _gen(device, seed)
    P: Dict[str, torch.Tensor] = {}
    for name, shape, kind in dit_param_shapes(cfg):
        x = torch.randn(shape, generator=g, device=device, dtype=torch.float32)
        if kind == "linear":
            x = x * std
        else:
            x = 1.0 + 0.1 * x
        P[name] = x.to(torch.bfloat16)
    return P


class LazyDitWeights(Mapping):
    """The same recipe, one tensor at a time: `P[name]` draws the tensor from its OWN seed (seed, crc32(name)) on `device`
    and keeps nothing, so a 32 B-parameter Dev transformer (64 GB in bf16) is never resident outside the library — the
    loader (`DiT.load`, which iterates `.items()`) and the CPU oracle (`cpu()`: the same values moved to the host per access)
    see bit-identical parameters, each for the moment it uses them.  Values differ from `make_dit_weights` (one
    sequential generator); a given (seed, name, shape, device type) always yields the same tensor."""

    def __init__(self, cfg: dict, seed: int = 1234, device="cuda", std: float = 0.02, to_cpu: bool = False):
        self.cfg, self.seed, self.device, self.std, self.to_cpu = cfg, seed, device, std, to_cpu
        self.shapes = {name: (shape, kind) for name, shape, kind in dit_param_shapes(cfg)}

    def __len__(self):
        return len(self.shapes)

    def __iter__(self):
        return iter(self.shapes)

    def __contains__(self, name):
        return name in self.shapes

    def __getitem__(self, name: str) -> torch.Tensor:
        shape, kind = self.shapes[name]
        g = _gen(self.device, (self.seed * 1_000_003 + zlib.crc32(name.encode())) & 0x7FFFFFFFFFFFFFFF)
        x = torch.randn(shape, generator=g, device=self.device, dtype=torch.float32)
        x = (x * self.std if kind == "linear" else 1.0 + 0.1 * x).to(torch.bfloat16)
        return x.cpu() if self.to_cpu else x

    def cpu(self) -> "LazyDitWeights":
        """The oracle's view: the same device-generated values, copied to the host per access."""
        return LazyDitWeights(self.cfg, self.seed, self.device, self.std, to_cpu=True)


def make_vae_weights(vcfg: dict, seed: int = 4321, device="cpu", encoder: bool = True) -> Dict[str, torch.Tensor]:
    """conv ~ N(0, sqrt(2/fan_in)), linear ~ N(0, sqrt(1/in)), biases N(0, 0.05),
    GroupNorm weight 1+0.1N / bias 0.1N, BN mean N(0,0.5), var U(0.5,1.5)."""
    g = _gen(device, seed)
    P: Dict[str, torch.Tensor] = {}
    for name, shape, kind in vae_decoder_param_shapes(vcfg, encoder=encoder):
        if kind == "bn_var":
            x = 0.5 + torch.rand(shape, generator=g, device=device, dtype=torch.float32)
        else:
            x = torch.randn(shape, generator=g, device=device, dtype=torch.float32)
            if kind == "conv":
                fan_in = shape[1] * shape[2] * shape[3]
                x = x * (2.0 / fan_in) ** 0.5
            elif kind == "linear":
                x = x * (1.0 / shape[1]) ** 0.5
            elif kind == "bias":
                x = x * 0.05
            elif kind == "gn_w":
                x = 1.0 + 0.1 * x
            elif kind == "gn_b":
                x = 0.1 * x
            elif kind == "bn_mean":
                x = 0.5 * x
        P[name] = x.to(torch.bfloat16)
    return P


def make_inputs(cfg: dict, height: int, width: int, s_txt: int = 512, device="cpu",
                noise_seed: int = 42, txt_seed: int = 7) -> Dict[str, torch.Tensor]:
    """noise [1,128,H/16,W/16] f32 (seed 42) and text embeddings [1,S_txt,J] bf16
    (seed 7), N(0,1) — SURVEY §8d.  (Real Qwen3 hidden states have heavier tails.)"""
    gn = _gen(device, noise_seed)
    gt = _gen(device, txt_seed)
    noise = torch.randn((1, 128, height // 16, width // 16), generator=gn, device=device, dtype=torch.float32)
    txt = torch.randn((1, s_txt, cfg["joint_dim"]), generator=gt, device=device, dtype=torch.float32).to(torch.bfloat16)
    return dict(noise=noise, txt=txt)
