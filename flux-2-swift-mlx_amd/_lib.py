"""ctypes binding of libflux2mi.so (include/flux2mi.h).

The library is the product; this module only declares prototypes.  It fails loudly
when the shared object is missing — there is no CPU or PyTorch fallback.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, os.environ.get("FLUX2MI_LIB", "libflux2mi.so"))   # FLUX2MI_LIB: timing-only ablation builds


class Flux2Error(RuntimeError):
    """Mirror of `Flux2Error` (Flux2Core/Flux2Core.swift:14-40); `.code` is the C status."""

    def __init__(self, code: int, message: str):
        super().__init__(f"[flux2mi status {code}] {message}")
        self.code = code
        self.message = message


MASK_GRAYSCALE_WHITE_INPAINT, MASK_ALPHA_TRANSPARENT_INPAINT = 0, 1  # Flux2MaskConvention
OK, ERR_INVALID_ARG, ERR_HIP, ERR_NOT_LOADED, ERR_SHAPE, ERR_CANCELLED, ERR_UNSUPPORTED, ERR_OOM, ERR_UNKNOWN_KEY = range(9)
F32, BF16, F16 = 0, 1, 2


class DitConfig(C.Structure):
    _fields_ = [("dim", C.c_int), ("heads", C.c_int), ("n_double", C.c_int), ("n_single", C.c_int),
                ("joint_dim", C.c_int), ("guidance_embeds", C.c_int), ("weight_format", C.c_int)]


class TextEncConfig(C.Structure):
    _fields_ = [("vocab", C.c_int), ("hidden", C.c_int), ("intermediate", C.c_int), ("n_layers", C.c_int),
                ("heads", C.c_int), ("kv_heads", C.c_int), ("head_dim", C.c_int), ("rope_theta", C.c_float),
                ("rms_eps", C.c_float), ("extract_layers", C.c_int * 3), ("qk_norm", C.c_int)]


class VaeConfig(C.Structure):
    _fields_ = [("channels", C.c_int * 4), ("latent_channels", C.c_int), ("groups", C.c_int),
                ("layers_per_block", C.c_int)]


class StepContext(C.Structure):
    _fields_ = [("step_idx", C.c_int), ("total_steps", C.c_int), ("sigma", C.c_float),
                ("sigma_next", C.c_float), ("height", C.c_int), ("width", C.c_int), ("is_i2i", C.c_int)]


STEP_HOOK = C.CFUNCTYPE(C.c_int, C.POINTER(StepContext), C.c_void_p, C.c_void_p, C.c_void_p)
PROGRESS_CB = C.CFUNCTYPE(None, C.c_int, C.c_int, C.c_void_p)
CHECKPOINT_CB = C.CFUNCTYPE(None, C.c_int, C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_void_p)


class GenerateParams(C.Structure):
    _fields_ = [
        ("height", C.c_int), ("width", C.c_int), ("steps", C.c_int), ("guidance", C.c_float),
        ("txt_emb", C.c_void_p), ("s_txt", C.c_int),
        ("neg_emb", C.c_void_p), ("s_neg", C.c_int), ("cfg_scale", C.c_float),
        ("noise", C.c_void_p), ("init_latents", C.c_void_p), ("strength", C.c_float),
        ("ref_latents", C.c_void_p), ("ref_ids", C.c_void_p), ("s_ref", C.c_int),
        ("custom_sigmas", C.POINTER(C.c_float)), ("n_custom_sigmas", C.c_int),
        ("checkpoint_interval", C.c_int),
        ("on_step", STEP_HOOK), ("on_step_user", C.c_void_p),
        ("on_progress", PROGRESS_CB), ("on_progress_user", C.c_void_p),
        ("on_checkpoint", CHECKPOINT_CB), ("on_checkpoint_user", C.c_void_p),
        ("skip_decode", C.c_int), ("use_kv_cache", C.c_int),
        ("out_image", C.c_void_p), ("out_image_device", C.c_void_p), ("out_latents", C.c_void_p),
        ("out_height", C.c_int), ("out_width", C.c_int), ("effective_steps", C.c_int),
        ("ms_denoise", C.c_float), ("ms_decode", C.c_float),
    ]


# name -> (restype, argtypes); every symbol include/flux2mi.h declares
_I, _F, _P, _I64 = C.c_int, C.c_float, C.c_void_p, C.c_int64
_PI = C.POINTER(C.c_int)
PROTOTYPES = {
    "flux2mi_version": (_I, []),
    "flux2mi_last_error": (C.c_char_p, []),
    "flux2mi_ctx_create": (_I, [_I, _P, C.POINTER(_P)]),
    "flux2mi_ctx_destroy": (_I, [_P]),
    "flux2mi_ctx_sync": (_I, [_P]),
    "flux2mi_ctx_memory_stats": (_I, [_P, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "flux2mi_ctx_reset_peak": (_I, [_P]),
    "flux2mi_ctx_trim": (_I, [_P, C.POINTER(C.c_size_t)]),
    "flux2mi_buffer_alloc": (_I, [_P, C.c_size_t, C.POINTER(_P)]),
    "flux2mi_buffer_free": (_I, [_P, _P]),
    "flux2mi_buffer_upload": (_I, [_P, _P, _P, C.c_size_t]),
    "flux2mi_buffer_download": (_I, [_P, _P, _P, C.c_size_t]),
    "flux2mi_dp_rccl_library": (C.c_char_p, []),
    "flux2mi_dp_unique_id": (_I, [_P]),
    "flux2mi_dp_init": (_I, [_P, _I, _I, _P]),
    "flux2mi_dp_finalize": (_I, [_P]),
    "flux2mi_dp_rank": (_I, [_P]),
    "flux2mi_dp_world": (_I, [_P]),
    "flux2mi_dp_broadcast_embeddings": (_I, [_P, _P, C.c_int64, _I]),
    "flux2mi_dp_all_gather_images": (_I, [_P, _P, _P, C.c_int64]),
    "flux2mi_dp_all_reduce_max": (_I, [_P, C.POINTER(C.c_double)]),
    "flux2mi_dp_collective_timing": (_I, [_P, _I]),
    "flux2mi_dp_collective_stats": (_I, [_P, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int64), _I]),
    "flux2mi_array_to_bf16": (_I, [_P, _P, _P]),
    "flux2mi_dit_load_lora_safetensors": (_I, [_P, C.c_char_p, C.c_float, C.POINTER(_I), C.POINTER(_I)]),
    "flux2mi_ctx_stream": (_P, [_P]),
    "flux2mi_compute_empirical_mu": (_F, [_I, _I]),
    "flux2mi_set_timesteps": (_I, [_I, _I, _F, C.POINTER(_F), _I, _PI, _PI]),
    "flux2mi_set_custom_sigmas": (_I, [C.POINTER(_F), _I, C.POINTER(_F), _I, _PI]),
    "flux2mi_validate_dimensions": (None, [_I, _I, _PI, _PI]),
    "flux2mi_resolve_chain_dimensions": (None, [_I, _I, _I, _PI, _PI]),
    "flux2mi_image_position_ids": (_I, [_I, _I, C.POINTER(C.c_int32), _I]),
    "flux2mi_text_position_ids": (_I, [_I, C.POINTER(C.c_int32), _I]),
    "flux2mi_reference_position_ids": (_I, [_PI, _PI, _I, _I, C.POINTER(C.c_int32), _I]),
    "flux2mi_sched_step": (_I, [_P, _P, _P, _F, _I64]),
    "flux2mi_scale_noise": (_I, [_P, _P, _P, _P, _F, _I64]),
    "flux2mi_cfg_combine": (_I, [_P, _P, _P, _F, _I64]),
    "flux2mi_repaint_blend": (_I, [_P, _P, _P, _P, _P, _F, _I, _I]),
    "flux2mi_dit_export_prequantized": (_I, [_P, C.c_char_p, C.c_char_p]),
    "flux2mi_dit_load_prequantized": (_I, [_P, C.c_char_p, C.c_char_p, _PI]),
    "flux2mi_image_resample_u8": (_I, [_P, _P, _I, _I, _I, C.POINTER(_F), _P, _I, _I]),
    "flux2mi_mask_inpaint_weights": (_I, [_P, _P, _I, _I, _I, _I, C.POINTER(_F), _P, _I, _I]),
    "flux2mi_mask_bounding_box": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _F, _PI, _PI]),
    "flux2mi_expand_crop_region": (None, [_PI, _I, _I, _I, _PI]),
    "flux2mi_composite_u8": (_I, [_P, _P, _I, _I, _I, _P, _I, _PI, _P]),
    "flux2mi_preprocess_image_for_vae": (_I, [_P, _P, _I, _I, _I, _P]),
    "flux2mi_outpaint_canvas": (_I, [_P, _P, _I, _I, _I, _P, _I, _I, _I, _I, C.c_uint64]),
    "flux2mi_outpaint_smart_mask": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I]),
    "flux2mi_mlx_random_key": (None, [C.c_uint64, C.POINTER(C.c_uint32)]),
    "flux2mi_mlx_random_split": (None, [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "flux2mi_mlx_random_bits": (_I, [_P, C.POINTER(C.c_uint32), _P, _I64]),
    "flux2mi_mlx_random_uniform": (_I, [_P, C.POINTER(C.c_uint32), _P, _I64, _F, _F]),
    "flux2mi_mlx_random_normal": (_I, [_P, C.POINTER(C.c_uint32), _P, _I64]),
    "flux2mi_array_alloc": (_I, [_P, C.POINTER(_I64), _I, C.POINTER(_P)]),
    "flux2mi_array_from_scalar": (_I, [_P, _F, C.POINTER(_P)]),
    "flux2mi_array_from_host": (_I, [_P, _P, C.POINTER(_I64), _I, C.POINTER(_P)]),
    "flux2mi_array_wrap": (_I, [_P, _P, C.POINTER(_I64), _I, C.POINTER(_P)]),
    "flux2mi_array_free": (_I, [_P]),
    "flux2mi_array_ndim": (_I, [_P]),
    "flux2mi_array_dim": (_I64, [_P, _I]),
    "flux2mi_array_data": (_P, [_P]),
    "flux2mi_array_add": (_I, [_P, _P, _P, C.POINTER(_P)]),
    "flux2mi_array_sub": (_I, [_P, _P, _P, C.POINTER(_P)]),
    "flux2mi_array_mul": (_I, [_P, _P, _P, C.POINTER(_P)]),
    "flux2mi_array_copy_to": (_I, [_P, _P, _P, _I64]),
    "flux2mi_array_to_host": (_I, [_P, _P, _P]),
    "flux2mi_array_random_normal": (_I, [_P, C.POINTER(C.c_uint32), C.POINTER(_I64), _I, C.POINTER(_P)]),
    "flux2mi_pack_patchified_to_sequence": (_I, [_P, _P, _P, _I, _I, _I]),
    "flux2mi_unpack_sequence_to_patchified": (_I, [_P, _P, _P, _I, _I, _I]),
    "flux2mi_latents_to_vae_input": (_I, [_P, _P, _P, _P, _P, _I, _I]),
    "flux2mi_vae_latents_to_packed": (_I, [_P, _P, _P, _P, _P, _I, _I]),
    "flux2mi_dit_preset": (_I, [C.c_char_p, C.POINTER(DitConfig)]),
    "flux2mi_dit_create": (_I, [_P, C.POINTER(DitConfig), C.POINTER(_P)]),
    "flux2mi_dit_destroy": (_I, [_P]),
    "flux2mi_dit_set_tensor": (_I, [_P, C.c_char_p, _I, C.POINTER(_I64), _I, _P]),
    "flux2mi_dit_finalize": (_I, [_P]),
    "flux2mi_dit_load_safetensors": (_I, [_P, C.c_char_p, _PI, _PI]),
    "flux2mi_dit_set_tensor_q8": (_I, [_P, C.c_char_p, _P, _P, _I64, _I64]),
    "flux2mi_dit_merge_lora": (_I, [_P, C.c_char_p, _I, _P, _P, _I, _F]),
    "flux2mi_dit_merge_lora_checked": (_I, [_P, C.c_char_p, _I, _P, _I64, _I64, _I, _P, _I64, _I64, _F]),
    "flux2mi_dit_slot_shape": (_I, [_P, C.c_char_p, C.POINTER(_I64), C.POINTER(_I64)]),
    "flux2mi_dit_param_bytes": (_I64, [_P]),
    "flux2mi_dit_forward": (_I, [_P, _P, _I, _P, _I, _F, _F, _P, _P, _P, _I]),
    "flux2mi_dit_forward_block": (_I, [_P, _I, _P, _P, _P, _I, _P, _I, _F, _F, _P, _P, _P]),
    "flux2mi_dit_last_modulation": (_I, [_P, _P, _I64]),
    "flux2mi_op_attention_split_info": (_I, [_P, _I, _I, _PI, _PI]),
    "flux2mi_dit_set_attention_softmax": (_I, [_P, _I]),
    "flux2mi_dit_attention_softmax_info": (_I, [_P, _PI, _PI]),
    "flux2mi_kv_cache_create": (_I, [_P, _I, _I, C.POINTER(_P)]),
    "flux2mi_kv_cache_destroy": (_I, [_P]),
    "flux2mi_kv_cache_entry": (_I, [_P, _I, _P, _P]),
    "flux2mi_dit_forward_kv_extract": (_I, [_P, _P, _P, _I, _P, _I, _P, _I, _F, _F, _P, _P, _P, _P]),
    "flux2mi_dit_forward_kv_cached": (_I, [_P, _P, _P, _I, _P, _I, _F, _F, _P, _P, _P]),
    "flux2mi_textenc_preset": (_I, [C.c_char_p, C.POINTER(TextEncConfig)]),
    "flux2mi_textenc_create": (_I, [_P, C.POINTER(TextEncConfig), C.POINTER(_P)]),
    "flux2mi_textenc_destroy": (_I, [_P]),
    "flux2mi_textenc_set_tensor": (_I, [_P, C.c_char_p, _I, C.POINTER(_I64), _I, _P]),
    "flux2mi_textenc_load_safetensors": (_I, [_P, C.c_char_p, _PI, _PI]),
    "flux2mi_textenc_finalize": (_I, [_P]),
    "flux2mi_textenc_param_bytes": (_I64, [_P]),
    "flux2mi_textenc_forward": (_I, [_P, _P, _I, _I, _I, _P]),
    "flux2mi_vae_preset": (_I, [C.c_char_p, C.POINTER(VaeConfig)]),
    "flux2mi_vae_create": (_I, [_P, C.POINTER(VaeConfig), C.POINTER(_P)]),
    "flux2mi_vae_destroy": (_I, [_P]),
    "flux2mi_vae_set_tensor": (_I, [_P, C.c_char_p, _I, C.POINTER(_I64), _I, _P]),
    "flux2mi_vae_finalize": (_I, [_P]),
    "flux2mi_vae_load_safetensors": (_I, [_P, C.c_char_p, _PI, _PI]),
    "flux2mi_vae_bn_mean": (_P, [_P]),
    "flux2mi_vae_bn_var": (_P, [_P]),
    "flux2mi_vae_decode": (_I, [_P, _P, _I, _I, _P, _P]),
    "flux2mi_vae_encode": (_I, [_P, _P, _I, _I, _P]),
    "flux2mi_vae_has_encoder": (_I, [_P]),
    "flux2mi_generate": (_I, [_P, _P, _P, C.POINTER(GenerateParams)]),
    "flux2mi_debug_mfma_stream": (_I, [_P, _I, C.c_double, C.POINTER(C.c_double)]),
    "flux2mi_prof_begin": (_I, [_P, _I]),
    "flux2mi_prof_end": (_I, [_P, _PI, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "flux2mi_op_gemm_bf16": (_I, [_P, _P, _P, _P, _I, _I, _I]),
    "flux2mi_op_gemm_bf16_out_bf16": (_I, [_P, _P, _P, _P, _I, _I, _I]),
    "flux2mi_op_gemm_fp8": (_I, [_P, _P, _P, _P, _I, _I, _I]),
    "flux2mi_op_gemm_fp8_blocks": (_I, [_P, _P, _P, _P, _I, _I, _I]),
    "flux2mi_op_attention_qk8_blocks": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, C.c_float]),
    "flux2mi_op_gemm_gate_resid": (_I, [_P, _P, _P, _P, _P, _I, _I, _I]),
    "flux2mi_op_vae_attention": (_I, [_P, _P, _P, _P, _P, _I, _I]),
    "flux2mi_op_layernorm_modulate": (_I, [_P, _P, _P, _I, _I, _I, _P, _P, _P, _P]),
    "flux2mi_op_qk_norm_rope": (_I, [_P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _F]),
    "flux2mi_op_qk_proj_norm_rope": (_I, [_P, _P, _P, _I, _I, _P, _P, _P, _F, _P]),
    "flux2mi_op_groupnorm_silu": (_I, [_P, _P, _P, C.c_longlong, _I, _I, _P, _P, _F, _I]),
    "flux2mi_op_attention": (_I, [_P, _P, _P, _I, _I]),
    "flux2mi_op_attention_prescaled": (_I, [_P, _P, _P, _I, _I, C.c_float]),
    "flux2mi_op_attention_qk8": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, C.c_float]),
    "flux2mi_op_attention_prescaled_range": (_I, [_P, _P, _P, _I, _I, C.c_float, _I, _I, _I]),
}

_lib = None


def lib() -> C.CDLL:
    """Load libflux2mi.so (once).  Raises if the HIP library was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP library first "
            f"(`python -c 'import __graft_entry__ as g; g.build()'` or `make -C {_HERE}/csrc`). "
            "There is no CPU fallback.")
    handle = C.CDLL(LIB_PATH)
    older_build = "FLUX2MI_LIB" in os.environ   # tools/ab_libs.sh: an earlier round's library for a same-box A/B may lack this round's entry points
    for name, (res, args) in PROTOTYPES.items():
        if older_build and not hasattr(handle, name):
            continue
        fn = getattr(handle, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = handle
    return _lib


def check(status: int) -> None:
    if status != OK:
        msg = lib().flux2mi_last_error()
        raise Flux2Error(status, msg.decode("utf-8", "replace") if msg else "")
