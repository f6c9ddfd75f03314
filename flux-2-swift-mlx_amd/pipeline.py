"""Host-side mirror of the reference's public surface for the hot path:
`Flux2Pipeline.generate*`, `Flux2StepHook`, `Flux2StepContext`,
`Flux2GenerationResult` (Flux2Core/Pipeline/Flux2Pipeline.swift:16-87, 855-1176) and
the RePaint chain (Flux2Chains/Flux2MaskedInpaintingChain.swift:300-431).

The loop itself runs in C++ (`flux2mi_generate`, csrc/pipeline.cpp); this file maps the
reference's argument names onto the C ABI so that tests read like the reference's.
What is an explicit input here and implicit in the reference:
  * text embeddings (`precomputed_embeddings`, bf16 [1,S_txt,J]) — the text encoders
    are outside the path;
  * noise — `noise` ([1,128,H/16,W/16] f32) may be passed explicitly; otherwise it is drawn as the
    reference draws it, `MLXRandom.seed(seed)` + `MLXRandom.normal` (MLX's Threefry stream restated in
    csrc/elementwise.hip; `self.random` carries the key state on for later draws such as the RePaint
    blend noise).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence

import os

import numpy as np
import torch

from . import _lib as L
from ._lib import Flux2Error, check, lib
from .api import Context, DiT, MLXRandom, VAE, resolve_chain_dimensions, validate_dimensions


@dataclass
class Flux2StepContext:
    """Flux2Pipeline.swift:42-57."""
    stepIdx: int
    totalSteps: int
    sigma: float
    sigmaNext: float
    height: int
    width: int
    isI2I: bool


# Flux2StepHook = (Flux2StepContext, latents [1,S,128] device tensor) -> latents
Flux2StepHook = Callable[[Flux2StepContext, torch.Tensor], torch.Tensor]


@dataclass
class Flux2GenerationResult:
    """Flux2Pipeline.swift:67-87 (image is u8 [H,W,3] instead of CGImage)."""
    image: Optional[np.ndarray]
    usedPrompt: str
    wasUpsampled: bool
    originalPrompt: str
    latents: Optional[torch.Tensor] = None
    effectiveSteps: int = 0
    msDenoise: float = 0.0
    msDecode: float = 0.0


class Flux2Pipeline:
    """`Flux2Pipeline(model:...)` restricted to the hot path: embeddings in, image out."""

    def __init__(self, model="klein-4b", vae_variant="small", device: int = 0, quantization: str = "bf16"):
        """`quantization` mirrors `Flux2Pipeline(model:quantization:)` (Flux2Pipeline.swift:286-308) for the
        transformer only: "bf16" or "fp8" (e4m3 weights + activations; stands in for qint8/int4)."""
        self.ctx = Context(device)
        # `model.supportsKVCache` (ModelRegistry: klein-9b-kv shares klein-9b's architecture)
        self.supports_kv_cache = isinstance(model, str) and model.endswith("-kv")
        if self.supports_kv_cache:
            model = model[:-3]
        self.quantization = quantization
        self.model = model
        self.vae_variant = vae_variant
        self.transformer: Optional[DiT] = None
        self.vae: Optional[VAE] = None
        self.text_encoder = None   # flux2mi.textenc.TextEncoder (Klein: Qwen3), optional
        self.random = MLXRandom(self.ctx)  # the reference's global MLXRandom state

    # loadModels(progressCallback:) — weights are handed over as tensors by key, or as .safetensors
    # paths (a str, or a list of shard paths) read by the native loader
    def load_models(self, dit_tensors=None, vae_tensors=None):
        def _is_paths(x):
            return isinstance(x, (str, os.PathLike)) or (isinstance(x, (list, tuple)) and x and isinstance(x[0], (str, os.PathLike)))

        if dit_tensors is not None:
            dit = DiT(self.ctx, self.model, weights=self.quantization)
            if _is_paths(dit_tensors):
                for path in ([dit_tensors] if isinstance(dit_tensors, (str, os.PathLike)) else dit_tensors):
                    dit.load_safetensors(path)
            else:
                dit.load(dit_tensors)
            self.transformer = dit.finalize()
        if vae_tensors is not None:
            vae = VAE(self.ctx, self.vae_variant)
            if _is_paths(vae_tensors):
                for path in ([vae_tensors] if isinstance(vae_tensors, (str, os.PathLike)) else vae_tensors):
                    vae.load_safetensors(path)
            else:
                vae.load(vae_tensors)
            self.vae = vae.finalize()
        return self

    def load_text_encoder(self, config, tensors, tokenizer_json: Optional[str] = None):
        """Klein text encoder (Qwen3-4B / Qwen3-8B, `KleinTextEncoder`): HF tensors by key + tokenizer.json."""
        from .textenc import TextEncoder
        self.text_encoder = TextEncoder(self.ctx, config).load(tensors).finalize()
        if tokenizer_json is not None:
            self.text_encoder.set_tokenizer(tokenizer_json)
        return self

    # ---- LoRA (Flux2Pipeline.swift:618-689, LoRA/LoRALoader.swift, WeightLoader.mergeLoRAWeights) ------
    def load_lora(self, path: str, scale: Optional[float] = None) -> int:
        """`loadLoRA(LoRAConfig(filePath:scale:))`: reads `<layer>.lora_A.weight` / `.lora_B.weight` pairs
        (prefixes `base_model.model.` / `transformer.` dropped), scale = (metadata lora_alpha / lora_rank,
        default 1) x `scale` (LoRAConfig.effectiveScale, default 1), and merges W += s B A into the resident
        bf16 weights.  Returns the number of merged layers; layers the transformer does not have are skipped
        like in the reference (:783-789)."""
        from safetensors import safe_open
        if self.transformer is None:
            raise Flux2Error(L.ERR_NOT_LOADED, "Transformer")
        merged = 0
        with safe_open(str(path), framework="pt") as f:
            md = f.metadata() or {}
            s = 1.0
            try:
                if float(md.get("lora_rank", 0)) > 0:
                    s = float(md["lora_alpha"]) / float(md["lora_rank"])
            except (KeyError, ValueError):
                s = 1.0
            s *= 1.0 if scale is None else float(scale)
            keys = set(f.keys())
            for k in sorted(keys):
                if not k.endswith(".lora_A.weight"):
                    continue
                base = k[:-len(".lora_A.weight")]
                kb = base + ".lora_B.weight"
                if kb not in keys:
                    continue
                layer = base
                for pre in ("base_model.model.", "transformer."):
                    if layer.startswith(pre):
                        layer = layer[len(pre):]
                try:
                    self.transformer.merge_lora(layer + ".weight", f.get_tensor(k), f.get_tensor(kb), s)
                    merged += 1
                except Flux2Error as e:
                    if e.code != L.ERR_INVALID_ARG:   # unknown layer: skip; anything else is an error
                        raise
        self._lora = getattr(self, "_lora", []) + [str(path)]
        return merged

    @property
    def hasLoRA(self) -> bool:
        return bool(getattr(self, "_lora", []))

    @property
    def isLoaded(self) -> bool:
        return self.transformer is not None and self.vae is not None

    @property
    def uses_guidance_embeds(self) -> bool:
        return bool(self.transformer.cfg.guidance_embeds) if self.transformer else False

    # ---- public generate* family -------------------------------------------------
    def generate_text_to_image(self, prompt: str = "", height: int = 1024, width: int = 1024, steps: int = 28,
                               guidance: float = 4.0, seed: Optional[int] = None, precomputed_embeddings=None,
                               noise=None, checkpoint_interval: Optional[int] = None, on_progress=None,
                               on_checkpoint=None, on_step: Optional[Flux2StepHook] = None) -> Flux2GenerationResult:
        return self.generate_with_result("textToImage", prompt=prompt, height=height, width=width, steps=steps,
                                         guidance=guidance, seed=seed, precomputed_embeddings=precomputed_embeddings,
                                         noise=noise, checkpoint_interval=checkpoint_interval, on_progress=on_progress,
                                         on_checkpoint=on_checkpoint, on_step=on_step)

    def generate_image_to_image(self, prompt: str = "", reference_latents=None, reference_ids=None, **kw):
        """`generateImageToImage(prompt:images:...)` with the references already encoded
        (VAE encode is a 'next' row): reference_latents f32 [1,S_ref,128], reference_ids i32 [S_ref,4]."""
        if reference_latents is None or reference_ids is None:
            raise Flux2Error(L.ERR_INVALID_ARG, "imageToImage requires 1-3 encoded reference images")
        return self.generate_with_result("imageToImage", prompt=prompt, reference_latents=reference_latents,
                                         reference_ids=reference_ids, **kw)

    def generate_with_result(self, mode: str = "textToImage", prompt: str = "", height: int = 1024,
                             width: int = 1024, steps: int = 28, guidance: float = 4.0, seed: Optional[int] = None,
                             precomputed_embeddings=None, negative_embeddings=None, cfg_scale: float = 1.0,
                             noise=None, checkpoint_interval: Optional[int] = None, init_latents=None,
                             strength: float = 1.0, on_progress=None, on_checkpoint=None,
                             on_step: Optional[Flux2StepHook] = None, reference_latents=None, reference_ids=None,
                             custom_sigmas: Optional[Sequence[float]] = None, skip_decode: bool = False,
                             return_latents: bool = False, kv_cache: Optional[bool] = None) -> Flux2GenerationResult:
        """`kv_cache`: reference tokens through the KV-cached path (the klein-9b-kv variant,
        Flux2Pipeline.swift:1555-1682); default = the pipeline's `supports_kv_cache`."""
        if self.transformer is None:
            raise Flux2Error(L.ERR_NOT_LOADED, "Transformer")
        if self.vae is None and not skip_decode:
            raise Flux2Error(L.ERR_NOT_LOADED, "VAE")
        if precomputed_embeddings is None:
            # "2. Text Encoding" phase (Flux2Pipeline.swift:1228-1300): Klein models encode the prompt with Qwen3
            if self.text_encoder is None:
                raise Flux2Error(L.ERR_NOT_LOADED, "TextEncoder (pass precomputed_embeddings, or load_text_encoder(...) for Klein models)")
            precomputed_embeddings = self.text_encoder.extract_klein_embeddings(prompt)
        dev = self.ctx.device
        H, W = validate_dimensions(height, width)
        s_img = (H // 16) * (W // 16)
        txt = precomputed_embeddings.to(device=dev, dtype=torch.bfloat16).reshape(-1, precomputed_embeddings.shape[-1]).contiguous()
        if noise is None:
            # Flux2Pipeline.swift:1203-1204 + LatentUtils.generatePatchifiedLatents (:32-41): seed, then one normal draw
            if seed is not None:
                self.random.seed(seed)
            noise = self.random.normal((1, 128, H // 16, W // 16))
        noise = noise.to(device=dev, dtype=torch.float32).contiguous()
        if tuple(noise.shape[-3:]) != (128, H // 16, W // 16):
            raise Flux2Error(L.ERR_SHAPE, f"noise shape {tuple(noise.shape)} != [1,128,{H // 16},{W // 16}]")

        p = L.GenerateParams()
        p.height, p.width, p.steps, p.guidance = height, width, steps, guidance
        p.txt_emb, p.s_txt = txt.data_ptr(), txt.shape[0]
        keep = [txt, noise]
        if negative_embeddings is not None:
            neg = negative_embeddings.to(device=dev, dtype=torch.bfloat16).reshape(-1, negative_embeddings.shape[-1]).contiguous()
            keep.append(neg)
            p.neg_emb, p.s_neg, p.cfg_scale = neg.data_ptr(), neg.shape[0], cfg_scale
        p.noise = noise.data_ptr()
        p.strength = strength
        if init_latents is not None:
            il = init_latents.to(device=dev, dtype=torch.float32).reshape(-1, 128).contiguous()
            if il.shape[0] != s_img:  # Flux2Pipeline.swift:1898-1901
                raise Flux2Error(L.ERR_SHAPE, f"initLatents shape {tuple(init_latents.shape)} does not match latents shape [1, {s_img}, 128]")
            keep.append(il)
            p.init_latents = il.data_ptr()
        if mode == "imageToImage" or reference_latents is not None:
            rl = reference_latents.to(device=dev, dtype=torch.float32).reshape(-1, 128).contiguous()
            ri = reference_ids.to(device=dev, dtype=torch.int32).contiguous()
            keep += [rl, ri]
            p.ref_latents, p.ref_ids, p.s_ref = rl.data_ptr(), ri.data_ptr(), rl.shape[0]
        if custom_sigmas is not None:
            cs = (C.c_float * len(custom_sigmas))(*[float(s) for s in custom_sigmas])
            keep.append(cs)
            p.custom_sigmas, p.n_custom_sigmas = cs, len(custom_sigmas)
        p.checkpoint_interval = checkpoint_interval or 0
        p.skip_decode = int(skip_decode)
        use_kv = self.supports_kv_cache if kv_cache is None else bool(kv_cache)
        p.use_kv_cache = int(use_kv and reference_latents is not None)

        stream = self.ctx.stream
        hook_error: List[BaseException] = []

        if on_step is not None:
            def _hook(cptr, lat_ptr, _stream, _user):
                try:
                    c = cptr.contents
                    ctx = Flux2StepContext(c.step_idx, c.total_steps, c.sigma, c.sigma_next, c.height, c.width, bool(c.is_i2i))
                    # zero-copy view of the library's latent buffer
                    lat = _tensor_from_ptr(lat_ptr, (1, s_img, 128), dev)
                    with torch.cuda.stream(stream):
                        out = on_step(ctx, lat)
                        if out is not None and out.data_ptr() != lat.data_ptr():
                            lat.copy_(out.reshape(1, s_img, 128))
                    return 0
                except BaseException as e:  # noqa: BLE001 - surfaced after the C call returns
                    hook_error.append(e)
                    return 1
            cb = L.STEP_HOOK(_hook)
            keep.append(cb)
            p.on_step = cb
        if on_progress is not None:
            pcb = L.PROGRESS_CB(lambda s, t, _u: on_progress(s, t))
            keep.append(pcb)
            p.on_progress = pcb
        if on_checkpoint is not None:
            def _ck(step, ptr, h, w, _u):
                arr = np.ctypeslib.as_array(ptr, shape=(h, w, 3)).copy()
                on_checkpoint(step, arr)
            ccb = L.CHECKPOINT_CB(_ck)
            keep.append(ccb)
            p.on_checkpoint = ccb

        image = None
        if not skip_decode:
            image = np.empty((H, W, 3), dtype=np.uint8)
            p.out_image = image.ctypes.data
        lat_out = None
        if return_latents or skip_decode:
            lat_out = torch.empty((1, s_img, 128), dtype=torch.float32, device=dev)
            p.out_latents = lat_out.data_ptr()

        self.ctx._enter()
        status = lib().flux2mi_generate(self.ctx.handle, self.transformer.handle,
                                        self.vae.handle if self.vae is not None else None, C.byref(p))
        self.ctx._exit()
        if hook_error:
            raise hook_error[0]
        check(status)
        return Flux2GenerationResult(image=image, usedPrompt=prompt, wasUpsampled=False, originalPrompt=prompt,
                                     latents=lat_out, effectiveSteps=p.effective_steps, msDenoise=p.ms_denoise,
                                     msDecode=p.ms_decode)

    # ---- chain helpers (Flux2Pipeline+ChainHelpers.swift) ------------------------
    resolveChainDimensions = staticmethod(resolve_chain_dimensions)

    @staticmethod
    def pack_mask_for_latent_blending(mask_hw: torch.Tensor, target_height: int, target_width: int) -> torch.Tensor:
        """[H,W] in [0,1], 1 = inpaint -> [1,(H/16)(W/16),1] (+ChainHelpers.swift:148-236).  The reference
        rasterises with CoreGraphics `.high` interpolation into 8 bits; this uses an exact area
        average rounded to 8 bits (host-side pre-processing, not a hot-path kernel)."""
        m = mask_hw.float().reshape(1, 1, *mask_hw.shape)
        m = torch.nn.functional.adaptive_avg_pool2d(m, (target_height // 16, target_width // 16))
        return (torch.round(m * 255.0) / 255.0).reshape(1, -1, 1)

    def encode_image_to_packed_sequence(self, image, target_height: Optional[int] = None,
                                        target_width: Optional[int] = None) -> torch.Tensor:
        """`encodeImageToPackedSequence(_:targetHeight:targetWidth:samplePosterior: false)`
        (+ChainHelpers.swift:75-103).  `image`: u8 [H,W,3] (numpy or tensor) or f32 [1,3,H,W] already in
        [-1,1].  The reference resizes with CoreGraphics first; here the image must already have the
        target size (resampling is host-side pre-processing outside the hot path)."""
        if self.vae is None:
            raise Flux2Error(L.ERR_NOT_LOADED, "VAE not loaded")
        dev = self.ctx.device
        x = torch.as_tensor(image)
        if x.dtype == torch.uint8:
            x = (x.to(dev).float() / 127.5 - 1.0).permute(2, 0, 1).unsqueeze(0).contiguous()   # Flux2Pipeline.swift:2408-2418
        else:
            x = x.to(dev, torch.float32).reshape(1, 3, x.shape[-2], x.shape[-1]).contiguous()
        H, W = x.shape[-2], x.shape[-1]
        if (target_height not in (None, H)) or (target_width not in (None, W)):
            raise Flux2Error(L.ERR_SHAPE, f"image is {H}x{W}, target {target_height}x{target_width}: resize it first")
        return self.vae.encode_to_packed(x)

    def clearAll(self):
        if self.transformer:
            self.transformer.close()
        if self.vae:
            self.vae.close()
        self.transformer = self.vae = None


def _tensor_from_ptr(ptr: int, shape, device) -> torch.Tensor:
    """Zero-copy f32 device tensor over library-owned memory (valid during the callback)."""
    n = int(np.prod(shape))

    class _Holder:
        pass

    h = _Holder()
    h.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(h, device=device).view(*shape)


class Flux2MaskedInpaintingChain:
    """RePaint inpainting chain (Flux2Chains/Flux2MaskedInpaintingChain.swift:300-431) on
    already-encoded inputs: image latents [1,S,128] (VAE-encoded, BN-normalised, packed),
    mask [1,S,1] (1 = inpaint), blend noise [1,S,128] drawn once — passed in, or drawn at `run` the way
    the reference draws it (:380-385: `MLXRandom.seed(seed)` then `MLXRandom.normal(imageLatents.shape)`,
    before the pipeline re-seeds for the initial noise)."""

    def __init__(self, pipeline: Flux2Pipeline, image_latents: torch.Tensor, mask_latents: torch.Tensor,
                 blend_noise: Optional[torch.Tensor] = None):
        self.pipeline = pipeline
        dev = pipeline.ctx.device
        self.image_latents = image_latents.to(dev, torch.float32).contiguous()
        self.mask_latents = mask_latents.to(dev, torch.float32).contiguous()
        self.blend_noise = None if blend_noise is None else blend_noise.to(dev, torch.float32).contiguous()

    def on_step(self, ctx: Flux2StepContext, latents: torch.Tensor) -> torch.Tensor:
        """:399-403 — orig = (1-s')*img + s'*noise ; x = (1-m)*orig + m*x, one fused kernel, in place."""
        c = self.pipeline.ctx
        check(lib().flux2mi_repaint_blend(c.handle, latents.data_ptr(), self.image_latents.data_ptr(),
                                          self.blend_noise.data_ptr(), self.mask_latents.data_ptr(),
                                          ctx.sigmaNext, latents.shape[-2], latents.shape[-1]))
        return latents

    def run(self, precomputed_embeddings, height: int, width: int, steps: int, guidance: float = 4.0,
            strength: float = 1.0, noise=None, seed=None, **kw) -> Flux2GenerationResult:
        if self.blend_noise is None:
            if seed is not None:
                self.pipeline.random.seed(seed)
            self.blend_noise = self.pipeline.random.normal(tuple(self.image_latents.shape))
        return self.pipeline.generate_with_result(
            "textToImage", height=height, width=width, steps=steps, guidance=guidance, seed=seed, noise=noise,
            precomputed_embeddings=precomputed_embeddings,
            init_latents=self.image_latents if strength < 1.0 else None, strength=strength,
            on_step=self.on_step, **kw)
