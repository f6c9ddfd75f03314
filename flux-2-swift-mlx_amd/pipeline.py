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


@dataclass
class SchedulerOverrides:
    """LoRA/LoRAConfig.swift:7-35."""
    customSigmas: Optional[List[float]] = None
    numSteps: Optional[int] = None
    guidance: Optional[float] = None

    @property
    def hasOverrides(self) -> bool:
        return self.customSigmas is not None or self.numSteps is not None or self.guidance is not None


@dataclass
class LoRAConfig:
    """LoRA/LoRAConfig.swift:38-94 (JSON field names as the Codable struct writes them)."""
    filePath: str
    scale: Optional[float] = 1.0
    activationKeyword: Optional[str] = None
    schedulerOverrides: Optional[SchedulerOverrides] = None

    @property
    def name(self) -> str:
        return os.path.splitext(os.path.basename(self.filePath))[0]

    @property
    def effectiveScale(self) -> float:
        return 1.0 if self.scale is None else float(self.scale)

    @classmethod
    def load(cls, path: str) -> "LoRAConfig":
        import json
        with open(path) as f:
            d = json.load(f)
        so = d.get("schedulerOverrides")
        return cls(filePath=d["filePath"], scale=d.get("scale"), activationKeyword=d.get("activationKeyword"),
                   schedulerOverrides=None if so is None else SchedulerOverrides(so.get("customSigmas"), so.get("numSteps"), so.get("guidance")))

    def save(self, path: str):
        import json
        d = {"filePath": self.filePath, "scale": self.scale, "activationKeyword": self.activationKeyword}
        if self.schedulerOverrides is not None:
            d["schedulerOverrides"] = {k: v for k, v in vars(self.schedulerOverrides).items() if v is not None}
        with open(path, "w") as f:
            json.dump({k: v for k, v in d.items() if v is not None}, f, indent=2, sort_keys=True)


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
    def load_lora(self, config, scale: Optional[float] = None) -> int:
        """`loadLoRA(LoRAConfig)`: `config` is a `LoRAConfig` or a file path (then `scale` = LoRAConfig.scale).
        Reads `<layer>.lora_A.weight` / `.lora_B.weight` pairs, strips the prefixes `base_model.model.`, `transformer.`,
        `diffusion_model.` one after the other (LoRALoader.swift:132-140), scale = (metadata lora_alpha / lora_rank,
        default 1) x LoRAConfig.effectiveScale, and merges W += s B A into the resident bf16 weights.  Scheduler
        overrides of the config become `activeSchedulerOverrides` (:625-637).  Returns the number of merged layers;
        layers the transformer does not have are skipped like in the reference (:783-789) — a file in which NO layer
        matches is an error instead of a LoRA that silently does nothing.  File parsing, prefix handling and the merge are the
        library's (flux2mi_dit_load_lora_safetensors, csrc/loader.cpp): the Swift shim calls the same entry point."""
        if not isinstance(config, LoRAConfig):
            config = LoRAConfig(filePath=str(config), scale=scale)
        if self.transformer is None:
            raise Flux2Error(L.ERR_NOT_LOADED, "Transformer")
        n_merged, n_unmatched = C.c_int(0), C.c_int(0)
        self.ctx._enter()
        check(lib().flux2mi_dit_load_lora_safetensors(self.transformer.handle, str(config.filePath).encode(), float(config.effectiveScale),
                                                      C.byref(n_merged), C.byref(n_unmatched)))
        self.ctx._exit()
        merged = n_merged.value
        self._lora = getattr(self, "_lora", []) + [config.name]
        if config.schedulerOverrides is not None and config.schedulerOverrides.hasOverrides:
            self._lora_scheduler_overrides = config.schedulerOverrides
        return merged

    loadLoRA = load_lora

    def unloadLoRA(self, name: str):
        """:666-672.  As in the reference, weights already merged into the transformer stay merged; the name is dropped
        and the scheduler overrides are cleared when no LoRA is left."""
        self._lora = [n for n in getattr(self, "_lora", []) if n != name]
        if not self._lora:
            self._lora_scheduler_overrides = None

    def unloadAllLoRAs(self):
        self._lora = []
        self._lora_scheduler_overrides = None

    @property
    def activeSchedulerOverrides(self) -> Optional["SchedulerOverrides"]:
        """:685-689."""
        return getattr(self, "_lora_scheduler_overrides", None)

    def exportPrequantizedTransformer(self, path: str, source_fingerprint: str = "", allowLoRABaked: bool = False) -> str:
        """`Flux2Pipeline.exportPrequantizedTransformer` (Flux2Pipeline.swift:759-840): only for a quantised
        transformer configuration, and not with a merged LoRA unless explicitly allowed."""
        if self.quantization == "bf16":
            raise Flux2Error(L.ERR_INVALID_ARG, "exportPrequantizedTransformer requires a quantized transformer configuration (current: bf16)")
        if self.transformer is None:
            raise Flux2Error(L.ERR_NOT_LOADED, "Transformer")
        if self.hasLoRA and not allowLoRABaked:
            raise Flux2Error(L.ERR_INVALID_ARG, "a LoRA is merged into the transformer; pass allowLoRABaked=True to export it baked in")
        return self.transformer.export_prequantized(path, source_fingerprint)

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
                             return_latents: bool = False, kv_cache: Optional[bool] = None,
                             image_device: Optional[torch.Tensor] = None) -> Flux2GenerationResult:
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
        if custom_sigmas is None and self.activeSchedulerOverrides is not None:
            # Turbo-style LoRAs bring their own schedule; strength / initLatents are then ignored (Flux2Pipeline.swift:1879-1884)
            custom_sigmas = self.activeSchedulerOverrides.customSigmas
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
        if not skip_decode and image_device is not None:
            # the decoded bytes stay in HBM (u8 [H,W,3] device tensor of the caller), no D2H copy here
            if image_device.dtype != torch.uint8 or not image_device.is_cuda or image_device.numel() != H * W * 3 or not image_device.is_contiguous():
                raise Flux2Error(L.ERR_SHAPE, f"image_device must be a contiguous u8 device tensor of {H}x{W}x3")
            p.out_image_device = image_device.data_ptr()
        elif not skip_decode:
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

    def pack_mask_for_latent_blending(self, mask, target_height: int, target_width: int,
                                      convention: int = L.MASK_GRAYSCALE_WHITE_INPAINT, box=None) -> torch.Tensor:
        """`Flux2Pipeline.packMaskForLatentBlending` (+ChainHelpers.swift:148-236): mask u8 [H,W] (grayscale,
        white = inpaint) or u8 [H,W,4] with `MASK_ALPHA_TRANSPARENT_INPAINT` -> f32 [1,(H/16)(W/16),1], 1 = inpaint,
        rendered to the latent grid in 8 bits.  A float [H,W] mask in [0,1] is quantised to 8 bits first.  The
        reference rasterises with CoreGraphics; see include/flux2mi.h for the filter used here."""
        m = torch.as_tensor(mask)
        if m.dtype != torch.uint8:
            m = torch.round(m.float().clamp(0, 1) * 255.0).to(torch.uint8)
        w = self.ctx.mask_inpaint_weights(m, target_height // 16, target_width // 16, convention, box)
        return w.reshape(1, -1, 1)

    def encode_image_to_packed_sequence(self, image, target_height: Optional[int] = None,
                                        target_width: Optional[int] = None) -> torch.Tensor:
        """`encodeImageToPackedSequence(_:targetHeight:targetWidth:samplePosterior: false)`
        (+ChainHelpers.swift:75-103).  `image`: u8 [H,W,3|4] (numpy or tensor), resampled to the target size on the
        device when it differs (the reference draws it into a target-sized CGContext), or f32 [1,3,H,W] already in
        [-1,1] at the target size."""
        if self.vae is None:
            raise Flux2Error(L.ERR_NOT_LOADED, "VAE not loaded")
        dev = self.ctx.device
        x = torch.as_tensor(image)
        if x.dtype == torch.uint8:
            H, W = x.shape[0], x.shape[1]
            th, tw = target_height or H, target_width or W
            if (th, tw) != (H, W):
                x = self.ctx.image_resample_u8(x, th, tw)
            x = self.ctx.preprocess_image_for_vae(x)   # Flux2Pipeline.swift:2359-2418
        else:
            x = x.to(dev, torch.float32).reshape(1, 3, x.shape[-2], x.shape[-1]).contiguous()
            H, W = x.shape[-2], x.shape[-1]
            if (target_height not in (None, H)) or (target_width not in (None, W)):
                raise Flux2Error(L.ERR_SHAPE, f"image is {H}x{W}, target {target_height}x{target_width}: pass the u8 image to have it resized")
        return self.vae.encode_to_packed(x)

    def encode_reference_images(self, images, max_reference_pixels: int = 1024 * 1024):
        """`encodeReferenceImages` (Flux2Pipeline.swift:2143-2260): each u8 [H,W,3] image is scaled down to at most
        `max_reference_pixels`, floored to multiples of 32, VAE-encoded (posterior mean), patchified, BN-normalised and
        packed; the sequences are concatenated and get position ids with T = 10, 20, ... per image.
        -> (latents f32 [1,S_ref,128], ids i32 [S_ref,4])."""
        import math
        from .api import reference_position_ids
        if not images:
            raise Flux2Error(L.ERR_INVALID_ARG, "No reference images provided")
        max_area = max(int(max_reference_pixels), 32 * 32)
        packed, lh, lw = [], [], []
        for img in images:
            img = self.ctx._u8(img)
            th, tw = img.shape[0], img.shape[1]
            if tw * th > max_area:
                scale = math.sqrt(max_area / float(tw * th))
                tw, th = int(tw * scale), int(th * scale)
            tw, th = max((tw // 32) * 32, 32), max((th // 32) * 32, 32)
            packed.append(self.encode_image_to_packed_sequence(img, th, tw).reshape(-1, 128))
            lh.append(th // 16)
            lw.append(tw // 16)
        ids = torch.from_numpy(reference_position_ids(lh, lw)).to(self.ctx.device)
        return torch.cat(packed, 0).unsqueeze(0), ids

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


class Flux2InpaintCompositing:
    """Crop-and-stitch + pixel composite helpers (Flux2Chains/Flux2InpaintCompositing.swift), on the device.
    Rects are (x, y, w, h) in image space, top-left origin."""

    def __init__(self, ctx: Context):
        self.ctx = ctx

    def maskBoundingBox(self, mask, convention: int, imageWidth: int, imageHeight: int, threshold: float = 8.0 / 255.0):
        return self.ctx.mask_bounding_box(mask, convention, imageWidth, imageHeight, threshold)

    @staticmethod
    def expandCropRegion(bbox, padding: int, imageWidth: int, imageHeight: int):
        return Context.expand_crop_region(bbox, padding, imageWidth, imageHeight)

    def renderInpaintWeights(self, mask, width: int, height: int, convention: int):
        return self.ctx.mask_inpaint_weights(mask, height, width, convention)

    def composite(self, original, generated, cropRect, maskCrop, convention: int):
        """:112-162 — None when the crop leaves the image, like the reference."""
        o = torch.as_tensor(original)
        cx, cy, cw, ch = [int(v) for v in cropRect]
        if not (cw > 0 and ch > 0 and cx >= 0 and cy >= 0 and cx + cw <= o.shape[1] and cy + ch <= o.shape[0]):
            return None
        g = torch.as_tensor(generated)
        if tuple(g.shape[:2]) != (ch, cw):
            g = self.ctx.image_resample_u8(g, ch, cw)
        weights = self.ctx.mask_inpaint_weights(maskCrop, ch, cw, convention)
        return self.ctx.composite_u8(o, g, (cx, cy, cw, ch), weights)


def _crop_integral(img: torch.Tensor, rect):
    """`CGImage.cropping(to:)`: the rect is made integral (origin floored, far edge ceiled) and clipped."""
    import math
    x0, y0 = int(math.floor(rect[0])), int(math.floor(rect[1]))
    x1, y1 = int(math.ceil(rect[0] + rect[2])), int(math.ceil(rect[1] + rect[3]))
    x0, y0, x1, y1 = max(0, x0), max(0, y0), min(img.shape[1], x1), min(img.shape[0], y1)
    return img[y0:y1, x0:x1]


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

    def on_step_unfused(self, ctx: Flux2StepContext, latents: torch.Tensor) -> torch.Tensor:
        """The reference's closure AS WRITTEN (:399-403), every operator one façade call (api.MLXArray = what the Swift
        shim's MLXArray is): bit-identical to `on_step`'s fused kernel."""
        from .api import MLXArray
        c = self.pipeline.ctx
        imageLatentsCaptured = MLXArray.wrap(c, self.image_latents)
        maskLatentsCaptured = MLXArray.wrap(c, self.mask_latents)
        blendNoise = MLXArray.wrap(c, self.blend_noise)
        lat = MLXArray.wrap(c, latents)
        sigmaNext = MLXArray(c, ctx.sigmaNext)
        originalNoised = (1 - sigmaNext) * imageLatentsCaptured + sigmaNext * blendNoise
        return ((1 - maskLatentsCaptured) * originalNoised + maskLatentsCaptured * lat).torch()

    def run(self, precomputed_embeddings, height: int, width: int, steps: int, guidance: float = 4.0,
            strength: float = 1.0, noise=None, seed=None, unfused_hook: bool = False, **kw) -> Flux2GenerationResult:
        if self.blend_noise is None:
            if seed is not None:
                self.pipeline.random.seed(seed)
            self.blend_noise = self.pipeline.random.normal(tuple(self.image_latents.shape))
        mode = "imageToImage" if kw.get("reference_latents") is not None else "textToImage"   # :405-412
        return self.pipeline.generate_with_result(
            mode, height=height, width=width, steps=steps, guidance=guidance, seed=seed, noise=noise,
            precomputed_embeddings=precomputed_embeddings,
            init_latents=self.image_latents if strength < 1.0 else None, strength=strength,
            on_step=self.on_step_unfused if unfused_hook else self.on_step, **kw)


    @classmethod
    def run_on_image(cls, pipeline: Flux2Pipeline, image, mask, precomputed_embeddings, steps: int, guidance: float = 4.0,
                     strength: float = 1.0, seed=None, maskConvention: int = L.MASK_GRAYSCALE_WHITE_INPAINT,
                     maskCropPadding: Optional[int] = None, maxPixels: int = 1024 * 1024, compositeOnOriginal: bool = False,
                     referenceImages=None, useImageAsReference: bool = False, **kw) -> Flux2GenerationResult:
        """The chain's `run()` from pixels (:300-431): optional crop-and-stitch around the mask, VAE-encode the (cropped)
        source, rasterise the mask to the latent grid, draw the blend noise, denoise with the RePaint hook, and paste the
        result back onto the untouched original with the soft mask as per-pixel alpha.
        image: u8 [H,W,3]; mask: u8 [h,w] (grayscale) or [h,w,4] (alpha convention)."""
        comp = Flux2InpaintCompositing(pipeline.ctx)
        image = pipeline.ctx._u8(image)
        mask_t = torch.as_tensor(mask)
        mask_t = mask_t.to(pipeline.ctx.device).contiguous()
        ih, iw = image.shape[0], image.shape[1]
        mh, mw = mask_t.shape[0], mask_t.shape[1]
        work_image, work_mask, crop_rect = image, mask_t, None
        if maskCropPadding is not None:
            bbox = comp.maskBoundingBox(mask_t, maskConvention, iw, ih)
            if bbox is not None:
                region = comp.expandCropRegion(bbox, maskCropPadding, iw, ih)
                mask_region = (region[0] * mw / iw, region[1] * mh / ih, region[2] * mw / iw, region[3] * mh / ih)
                ci, cm = _crop_integral(image, region), _crop_integral(mask_t, mask_region)
                if ci.numel() and cm.numel():
                    work_image, work_mask, crop_rect = ci.contiguous(), cm.contiguous(), region
        target_h, target_w = resolve_chain_dimensions(work_image.shape[1], work_image.shape[0], maxPixels)
        image_latents = pipeline.encode_image_to_packed_sequence(work_image, target_h, target_w)
        mask_latents = pipeline.pack_mask_for_latent_blending(work_mask, target_h, target_w, maskConvention)
        chain = cls(pipeline, image_latents, mask_latents)
        refs = list(referenceImages) if referenceImages else ([work_image] if useImageAsReference else [])   # :405-412
        if refs:
            kw["reference_latents"], kw["reference_ids"] = pipeline.encode_reference_images(refs)
        result = chain.run(precomputed_embeddings, target_h, target_w, steps, guidance=guidance, strength=strength, seed=seed, **kw)
        if (crop_rect is not None or maskCropPadding is not None or compositeOnOriginal) and result.image is not None:
            region = crop_rect if crop_rect is not None else (0, 0, iw, ih)
            out = comp.composite(image, torch.as_tensor(result.image), region, work_mask, maskConvention)
            if out is not None:
                result.image = out.cpu().numpy()
            elif crop_rect is not None:
                raise Flux2Error(L.ERR_SHAPE, "Inpainting crop composite failed — cannot assemble the full-resolution output")
        return result


class Flux2OutpaintingChain:
    """`Flux2OutpaintingChain` (Flux2Chains/Flux2OutpaintingChain.swift): extend the canvas by top/bottom/left/right
    pixels (rounded UP to multiples of 32), fill the strips with seeded grey Gaussian noise, build the smart mask (white
    strips, ramp inside the keep, black keep) and run the masked-inpainting chain on the canvas with the original image
    as I2I reference."""

    def __init__(self, pipeline: Flux2Pipeline, image, top: int = 0, bottom: int = 0, left: int = 0, right: int = 0,
                 steps: int = 4, guidance: float = 1.0, seed: Optional[int] = None, transitionPixels: int = 32,
                 maxPixels: int = 1024 * 1024):
        self.pipeline, self.image = pipeline, image
        self.top, self.bottom, self.left, self.right = top, bottom, left, right
        self.steps, self.guidance, self.seed = steps, guidance, seed
        self.transitionPixels, self.maxPixels = transitionPixels, maxPixels

    @staticmethod
    def roundUpToMultipleOf32(x: int) -> int:   # :324-326
        return 0 if x == 0 else ((x + 31) // 32) * 32

    def buildOutpaintCanvas(self, sourceImage, canvasWidth: int, canvasHeight: int, offsetX: int, offsetY: int, noiseSeed: int):
        return self.pipeline.ctx.outpaint_canvas(sourceImage, canvasHeight, canvasWidth, offsetX, offsetY, noiseSeed)

    def buildSmartMask(self, canvasWidth: int, canvasHeight: int, keepX: int, keepY: int, keepWidth: int, keepHeight: int,
                       transitionPixels: int):
        return self.pipeline.ctx.outpaint_smart_mask(canvasHeight, canvasWidth, keepX, keepY, keepWidth, keepHeight, transitionPixels)

    def run(self, precomputed_embeddings, **kw) -> Flux2GenerationResult:
        """:185-277."""
        if min(self.top, self.bottom, self.left, self.right) < 0:
            raise Flux2Error(L.ERR_INVALID_ARG, "Padding values must be non-negative")
        if self.top + self.bottom + self.left + self.right <= 0:
            raise Flux2Error(L.ERR_INVALID_ARG, "At least one side must have non-zero padding")
        t, b, l, r = (self.roundUpToMultipleOf32(v) for v in (self.top, self.bottom, self.left, self.right))
        image = self.pipeline.ctx._u8(self.image)
        ih, iw = image.shape[0], image.shape[1]
        canvas_w, canvas_h = iw + l + r, ih + t + b
        if canvas_w % 32 or canvas_h % 32:
            raise Flux2Error(L.ERR_INVALID_ARG, f"Canvas dimensions {canvas_w}x{canvas_h} are not multiples of 32 — the input "
                             "image's own dimensions need to be padded by the caller, or supply paddings that compensate.")
        canvas = self.buildOutpaintCanvas(image, canvas_w, canvas_h, l, t, self.seed or 0)
        mask = self.buildSmartMask(canvas_w, canvas_h, l, t, iw, ih, self.transitionPixels)
        return Flux2MaskedInpaintingChain.run_on_image(
            self.pipeline, canvas, mask, precomputed_embeddings, steps=self.steps, guidance=self.guidance, seed=self.seed,
            referenceImages=[image], maxPixels=max(self.maxPixels, canvas_w * canvas_h), **kw)
