"""Thin object wrappers over the C ABI.  torch is used for device memory and stream
ordering only; every compute call goes through libflux2mi.so."""
from __future__ import annotations

import ctypes as C
import weakref
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib as L
from ._lib import Flux2Error, check, lib

_DT = {torch.float32: L.F32, torch.bfloat16: L.BF16, torch.float16: L.F16}


# --------------------------------------------------------------------------- host-side functions
def compute_empirical_mu(image_seq_len: int, num_steps: int) -> float:
    return float(lib().flux2mi_compute_empirical_mu(image_seq_len, num_steps))


def set_timesteps(num_inference_steps: int, image_seq_len: Optional[int] = None, strength: float = 1.0):
    """FlowMatchEulerScheduler.setTimesteps -> (sigmas float32 array, tStart)."""
    cap = num_inference_steps + 2
    buf = (C.c_float * cap)()
    n, t0 = C.c_int(0), C.c_int(0)
    check(lib().flux2mi_set_timesteps(num_inference_steps, image_seq_len or 0, strength, buf, cap, C.byref(n), C.byref(t0)))
    return np.array(buf[: n.value], dtype=np.float32), t0.value


def set_custom_sigmas(sigmas: Sequence[float]) -> np.ndarray:
    n_in = len(sigmas)
    src = (C.c_float * max(n_in, 1))(*[float(s) for s in sigmas])
    dst = (C.c_float * (n_in + 1))()
    n = C.c_int(0)
    check(lib().flux2mi_set_custom_sigmas(src, n_in, dst, n_in + 1, C.byref(n)))
    return np.array(dst[: n.value], dtype=np.float32)


def validate_dimensions(height: int, width: int) -> Tuple[int, int]:
    h, w = C.c_int(0), C.c_int(0)
    lib().flux2mi_validate_dimensions(height, width, C.byref(h), C.byref(w))
    return h.value, w.value


def resolve_chain_dimensions(width: int, height: int, max_pixels: int = 1024 * 1024) -> Tuple[int, int]:
    """-> (height, width), Flux2Pipeline.resolveChainDimensions."""
    h, w = C.c_int(0), C.c_int(0)
    lib().flux2mi_resolve_chain_dimensions(width, height, max_pixels, C.byref(h), C.byref(w))
    return h.value, w.value


def image_position_ids(height: int, width: int) -> np.ndarray:
    n = (height // 16) * (width // 16)
    out = np.zeros((max(n, 1), 4), dtype=np.int32)
    check(lib().flux2mi_image_position_ids(height, width, out.ctypes.data_as(C.POINTER(C.c_int32)), n))
    return out[:n]


def text_position_ids(length: int) -> np.ndarray:
    out = np.zeros((max(length, 1), 4), dtype=np.int32)
    check(lib().flux2mi_text_position_ids(length, out.ctypes.data_as(C.POINTER(C.c_int32)), length))
    return out[:length]


def reference_position_ids(latent_heights: Sequence[int], latent_widths: Sequence[int], scale: int = 10) -> np.ndarray:
    n = sum(h * w for h, w in zip(latent_heights, latent_widths))
    k = len(latent_heights)
    hs = (C.c_int * max(k, 1))(*latent_heights)
    ws = (C.c_int * max(k, 1))(*latent_widths)
    out = np.zeros((max(n, 1), 4), dtype=np.int32)
    check(lib().flux2mi_reference_position_ids(hs, ws, k, scale, out.ctypes.data_as(C.POINTER(C.c_int32)), n))
    return out[:n]


# --------------------------------------------------------------------------- device objects
def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _req(t: torch.Tensor, dtype, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise Flux2Error(L.ERR_INVALID_ARG, f"{name} must be a device tensor")
    if t.dtype != dtype:
        raise Flux2Error(L.ERR_INVALID_ARG, f"{name} must be {dtype}, got {t.dtype}")
    return t.contiguous()


class Context:
    """One GPU + one HIP stream + one generation in flight (flux2mi_ctx)."""

    def __init__(self, device: int = 0):
        if not torch.cuda.is_available():
            raise Flux2Error(L.ERR_HIP, "no HIP device visible: the hot path has no CPU fallback")
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        self.stream = torch.cuda.Stream(device=self.device)
        h = C.c_void_p()
        check(lib().flux2mi_ctx_create(device, C.c_void_p(self.stream.cuda_stream), C.byref(h)))
        self.handle = h
        self._children = []  # weakrefs of DiT / VAE objects: they must be destroyed before the context

    # stream ordering against torch's current stream (no host sync)
    def _enter(self):
        self.stream.wait_stream(torch.cuda.current_stream(self.device))

    def _exit(self):
        torch.cuda.current_stream(self.device).wait_stream(self.stream)

    def sync(self):
        check(lib().flux2mi_ctx_sync(self.handle))

    # ---- MLX.Memory (Flux2CLI/InpaintCommand.swift:253-271): what the library holds on this GPU
    def memory_stats(self) -> dict:
        """{"active", "peak", "cache"} in bytes: MLX.Memory.activeMemory / peakMemory / cacheMemory (cache = trimmable workspaces)."""
        a, p, w = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
        check(lib().flux2mi_ctx_memory_stats(self.handle, C.byref(a), C.byref(p), C.byref(w)))
        return {"active": int(a.value), "peak": int(p.value), "cache": int(w.value)}

    def reset_peak_memory(self):
        check(lib().flux2mi_ctx_reset_peak(self.handle))

    def clear_cache(self) -> int:
        """MLX.Memory.clearCache(): release the library's workspaces on this GPU; returns the bytes given back."""
        f = C.c_size_t(0)
        check(lib().flux2mi_ctx_trim(self.handle, C.byref(f)))
        return int(f.value)

    def close(self):
        if self.handle:
            for ref in list(self._children):
                child = ref()
                if child is not None:
                    child.close()
            self._children = []
            lib().flux2mi_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- elementwise ops (in place on torch tensors)
    def sched_step(self, x: torch.Tensor, v: torch.Tensor, dt: float):
        self._enter()
        check(lib().flux2mi_sched_step(self.handle, _ptr(x), _ptr(v), dt, x.numel()))
        self._exit()

    def scale_noise(self, sample: torch.Tensor, sigma: float, noise: torch.Tensor) -> torch.Tensor:
        out = torch.empty_like(sample)
        self._enter()
        check(lib().flux2mi_scale_noise(self.handle, _ptr(out), _ptr(sample), _ptr(noise), sigma, out.numel()))
        self._exit()
        return out

    def cfg_combine(self, v_cond: torch.Tensor, v_uncond: torch.Tensor, guidance: float):
        self._enter()
        check(lib().flux2mi_cfg_combine(self.handle, _ptr(v_cond), _ptr(v_uncond), guidance, v_cond.numel()))
        self._exit()

    def repaint_blend(self, x, image_latents, blend_noise, mask, sigma_next: float):
        seq, c = x.shape[-2], x.shape[-1]
        self._enter()
        check(lib().flux2mi_repaint_blend(self.handle, _ptr(x), _ptr(image_latents), _ptr(blend_noise), _ptr(mask),
                                          sigma_next, seq, c))
        self._exit()

    # ---- pixel-space pre/post-processing of the chains (u8 [H,W,C] device tensors; csrc/imaging.hip)
    def _u8(self, image) -> torch.Tensor:
        t = torch.as_tensor(image)
        if t.dtype != torch.uint8:
            raise Flux2Error(L.ERR_INVALID_ARG, f"expected a uint8 image, got {t.dtype}")
        if t.dim() == 2:
            t = t.unsqueeze(-1)
        return t.to(self.device).contiguous()

    @staticmethod
    def _box(box):
        return None if box is None else (C.c_float * 4)(*[float(v) for v in box])

    def image_resample_u8(self, image, out_h: int, out_w: int, box=None) -> torch.Tensor:
        """`CGContext.draw(image, in:)` of renderRGBA (Flux2InpaintCompositing.swift:166-182) with the library's
        documented filter; box = (x0, y0, x1, y1) source rectangle."""
        src = self._u8(image)
        h, w, c = src.shape
        out = torch.empty((out_h, out_w, c), dtype=torch.uint8, device=self.device)
        self._enter()
        check(lib().flux2mi_image_resample_u8(self.handle, _ptr(src), h, w, c, self._box(box), _ptr(out), out_h, out_w))
        self._exit()
        return out

    def mask_inpaint_weights(self, mask, out_h: int, out_w: int, convention: int = 0, box=None) -> torch.Tensor:
        """renderInpaintWeights (:196-227) -> f32 [out_h, out_w], 1 = inpaint."""
        src = self._u8(mask)
        h, w, c = src.shape
        out = torch.empty((out_h, out_w), dtype=torch.float32, device=self.device)
        self._enter()
        check(lib().flux2mi_mask_inpaint_weights(self.handle, _ptr(src), h, w, c, convention, self._box(box), _ptr(out), out_h, out_w))
        self._exit()
        return out

    def mask_bounding_box(self, mask, convention: int, image_width: int, image_height: int, threshold: float = 8.0 / 255.0):
        """maskBoundingBox (:23-65) -> (x, y, w, h) or None."""
        src = self._u8(mask)
        h, w, c = src.shape
        rect, found = (C.c_int * 4)(), C.c_int(0)
        self._enter()
        check(lib().flux2mi_mask_bounding_box(self.handle, _ptr(src), h, w, c, convention, image_width, image_height, threshold,
                                              rect, C.byref(found)))
        self._exit()
        return tuple(rect) if found.value else None

    @staticmethod
    def expand_crop_region(bbox, padding: int, image_width: int, image_height: int):
        """expandCropRegion (:69-108) -> (x, y, w, h)."""
        out = (C.c_int * 4)()
        lib().flux2mi_expand_crop_region((C.c_int * 4)(*[int(v) for v in bbox]), int(padding), image_width, image_height, out)
        return tuple(out)

    def composite_u8(self, base, generated, crop_rect, weights) -> torch.Tensor:
        """composite (:112-162); returns the composited copy of `base` (u8 [H,W,3|4])."""
        b = self._u8(base).clone()
        g = self._u8(generated)
        wt = weights.to(self.device, torch.float32).contiguous()
        cx, cy, cw, ch = [int(v) for v in crop_rect]
        if tuple(g.shape[:2]) != (ch, cw) or wt.numel() != cw * ch:
            raise Flux2Error(L.ERR_SHAPE, f"generated {tuple(g.shape)} / weights {tuple(wt.shape)} do not match the crop {cw}x{ch}")
        self._enter()
        check(lib().flux2mi_composite_u8(self.handle, _ptr(b), b.shape[0], b.shape[1], b.shape[2], _ptr(g), g.shape[2],
                                         (C.c_int * 4)(cx, cy, cw, ch), _ptr(wt)))
        self._exit()
        return b

    def preprocess_image_for_vae(self, image) -> torch.Tensor:
        """preprocessImageForVAE (Flux2Pipeline.swift:2359-2418): u8 [H,W,3|4] -> f32 [1,3,H,W] in [-1,1]."""
        src = self._u8(image)
        h, w, c = src.shape
        out = torch.empty((1, 3, h, w), dtype=torch.float32, device=self.device)
        self._enter()
        check(lib().flux2mi_preprocess_image_for_vae(self.handle, _ptr(src), h, w, c, _ptr(out)))
        self._exit()
        return out

    def outpaint_canvas(self, source, canvas_h: int, canvas_w: int, offset_x: int, offset_y: int, noise_seed: int) -> torch.Tensor:
        """Flux2OutpaintingChain.buildOutpaintCanvas (:331-395) -> u8 [canvas_h, canvas_w, 3]."""
        src = self._u8(source)
        out = torch.empty((canvas_h, canvas_w, 3), dtype=torch.uint8, device=self.device)
        self._enter()
        check(lib().flux2mi_outpaint_canvas(self.handle, _ptr(src), src.shape[0], src.shape[1], src.shape[2], _ptr(out), canvas_h, canvas_w,
                                            offset_x, offset_y, C.c_uint64(int(noise_seed) & 0xFFFFFFFFFFFFFFFF)))
        self._exit()
        return out

    def outpaint_smart_mask(self, canvas_h: int, canvas_w: int, keep_x: int, keep_y: int, keep_w: int, keep_h: int,
                            transition_pixels: int) -> torch.Tensor:
        """Flux2OutpaintingChain.buildSmartMask (:397-474) -> u8 [canvas_h, canvas_w]."""
        out = torch.empty((canvas_h, canvas_w), dtype=torch.uint8, device=self.device)
        self._enter()
        check(lib().flux2mi_outpaint_smart_mask(self.handle, _ptr(out), canvas_h, canvas_w, keep_x, keep_y, keep_w, keep_h, transition_pixels))
        self._exit()
        return out

    def pack_patchified_to_sequence(self, p: torch.Tensor) -> torch.Tensor:
        _, c, h, w = p.shape
        out = torch.empty((1, h * w, c), dtype=torch.float32, device=p.device)
        self._enter()
        check(lib().flux2mi_pack_patchified_to_sequence(self.handle, _ptr(p.contiguous()), _ptr(out), c, h, w))
        self._exit()
        return out

    def unpack_sequence_to_patchified(self, s: torch.Tensor, h: int, w: int) -> torch.Tensor:
        c = s.shape[-1]
        out = torch.empty((1, c, h, w), dtype=torch.float32, device=s.device)
        self._enter()
        check(lib().flux2mi_unpack_sequence_to_patchified(self.handle, _ptr(s.contiguous()), _ptr(out), c, h, w))
        self._exit()
        return out

    def latents_to_vae_input(self, packed, bn_mean, bn_var, h16: int, w16: int) -> torch.Tensor:
        z = torch.empty((1, 32, 2 * h16, 2 * w16), dtype=torch.float32, device=packed.device)
        self._enter()
        check(lib().flux2mi_latents_to_vae_input(self.handle, _ptr(packed.contiguous()), _ptr(bn_mean), _ptr(bn_var),
                                                 _ptr(z), h16, w16))
        self._exit()
        return z

    def vae_latents_to_packed(self, z, bn_mean, bn_var) -> torch.Tensor:
        h16, w16 = z.shape[-2] // 2, z.shape[-1] // 2
        out = torch.empty((1, h16 * w16, 128), dtype=torch.float32, device=z.device)
        self._enter()
        check(lib().flux2mi_vae_latents_to_packed(self.handle, _ptr(z.contiguous()), _ptr(bn_mean), _ptr(bn_var),
                                                  _ptr(out), h16, w16))
        self._exit()
        return out

    # ---- measurement aid
    def prof_begin(self, kernel_class: int):
        check(lib().flux2mi_prof_begin(self.handle, kernel_class))

    PROF_KERNELS = {
        0: "gemm256s_kernel (single_blocks.linear1: [S,d] x [9d,d]^T as two launches: [q|k] with fused RMSNorm+RoPE "
           "+ SwiGLU rows, and V^T = W_v x^T; fp8 weights: gemm256s_kernel<NI, FP8>)",
        1: "gemm256s_kernel (bf16 / fp8) / gemm_bf16_128x128_kernel (every DiT GEMM)",
        2: "attn64b_kernel (fixed / per-query data offset) / attn_fwd_kernel (joint attention)",
        3: "gemm256s_kernel<NI, false, CONV> / conv2d_g128_kernel / conv2d_kernel (VAE convolutions)",
        4: "gemm256s_kernel (single_blocks.linear2: [S,4d] x [d,4d]^T, one round of tiles, x += gate*y epilogue)",
        5: "layernorm_modulate_kernel (LayerNorm + AdaLN modulate; work = algorithmic bytes: f32 row in, bf16 / e4m3 row out)",
        6: "gn_apply_padded_kernel (GroupNorm apply + SiLU into the zero-haloed operand; work = algorithmic bytes)",
        7: "gemm256s_kernel (double blocks' attention proj and mlp_out: the one-round N = d GEMMs)",
        8: "vae_attention_kernel (the VAE mid block's single-head attention, head dim 512 / 384, streaming: no P x P scores in HBM; "
           "work = 4 P^2 C)",
    }

    def prof_kernel_name(self, kernel_class: int) -> str:
        return self.PROF_KERNELS.get(kernel_class, str(kernel_class))

    def mfma_stream_tflops(self, fp8: bool = False, seconds: float = 0.6) -> float:
        """The matrix-pipe rate this card sustains on a register-only MFMA stream at its power cap (diagnostic, synchronous)."""
        v = C.c_double(0.0)
        self._enter()
        check(lib().flux2mi_debug_mfma_stream(self.handle, 1 if fp8 else 0, float(seconds), C.byref(v)))
        self._exit()
        return v.value

    def prof_end(self):
        n, ms, fl = C.c_int(0), C.c_double(0.0), C.c_double(0.0)
        check(lib().flux2mi_prof_end(self.handle, C.byref(n), C.byref(ms), C.byref(fl)))
        return n.value, ms.value, fl.value

    # ---- standalone ops
    def op_gemm_bf16(self, a: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
        a, w = _req(a, torch.bfloat16, "A"), _req(w, torch.bfloat16, "W")
        m, k = a.shape
        n = w.shape[0]
        out = torch.empty((m, n), dtype=torch.float32, device=a.device)
        self._enter()
        check(lib().flux2mi_op_gemm_bf16(self.handle, _ptr(a), _ptr(w), _ptr(out), m, n, k))
        self._exit()
        return out

    def op_gemm_bf16_out_bf16(self, a: torch.Tensor, w: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        a, w = _req(a, torch.bfloat16, "A"), _req(w, torch.bfloat16, "W")
        m, k = a.shape
        n = w.shape[0]
        if out is None:
            out = torch.empty((m, n), dtype=torch.bfloat16, device=a.device)
        self._enter()
        check(lib().flux2mi_op_gemm_bf16_out_bf16(self.handle, _ptr(a), _ptr(w), _ptr(out), m, n, k))
        self._exit()
        return out

    def op_gemm_gate_resid(self, a: torch.Tensor, w: torch.Tensor, gate: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
        """x f32 [M,N] += gate[N] * (a @ w.T), in place; returns x."""
        a, w = _req(a, torch.bfloat16, "A"), _req(w, torch.bfloat16, "W")
        gate, x = _req(gate, torch.float32, "gate"), _req(x, torch.float32, "x")
        m, k = a.shape
        n = w.shape[0]
        self._enter()
        check(lib().flux2mi_op_gemm_gate_resid(self.handle, _ptr(a), _ptr(w), _ptr(gate), _ptr(x), m, n, k))
        self._exit()
        return x

    def op_vae_attention(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
        """softmax(q k^T / sqrt(C)) v for bf16 [P, C] (C = 512 / 384): the VAE mid block's streaming attention kernel."""
        q, k, v = _req(q, torch.bfloat16, "q"), _req(k, torch.bfloat16, "k"), _req(v, torch.bfloat16, "v")
        p, c = q.shape
        out = torch.empty_like(q)
        self._enter()
        check(lib().flux2mi_op_vae_attention(self.handle, _ptr(q), _ptr(k), _ptr(v), _ptr(out), p, c))
        self._exit()
        return out

    def op_gemm_fp8(self, a: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
        a, w = _req(a, torch.bfloat16, "A"), _req(w, torch.bfloat16, "W")
        m, k = a.shape
        n = w.shape[0]
        out = torch.empty((m, n), dtype=torch.float32, device=a.device)
        self._enter()
        check(lib().flux2mi_op_gemm_fp8(self.handle, _ptr(a), _ptr(w), _ptr(out), m, n, k))
        self._exit()
        return out

    def op_gemm_fp8_blocks(self, a: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
        """A quantised in 16-row x 128-column e4m3 blocks with power-of-two scales (the form the fp8 models' epilogues emit), W per row."""
        a, w = _req(a, torch.bfloat16, "A"), _req(w, torch.bfloat16, "W")
        m, k = a.shape
        n = w.shape[0]
        out = torch.empty((m, n), dtype=torch.float32, device=a.device)
        self._enter()
        check(lib().flux2mi_op_gemm_fp8_blocks(self.handle, _ptr(a), _ptr(w), _ptr(out), m, n, k))
        self._exit()
        return out

    def op_layernorm_modulate(self, x, split, shift0, scale0, shift1, scale1) -> torch.Tensor:
        x = _req(x, torch.float32, "x")
        vs = [_req(v, torch.float32, "modulation vector") for v in (shift0, scale0, shift1, scale1)]
        y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
        self._enter()
        check(lib().flux2mi_op_layernorm_modulate(self.handle, _ptr(x), _ptr(y), x.shape[0], x.shape[1], split, *[_ptr(v) for v in vs]))
        self._exit()
        return y

    def op_qk_norm_rope(self, qkv, heads, split, qw0, kw0, qw1, kw1, ids, q_mul=1.0) -> torch.Tensor:
        """Returns a copy of qkv with q | k normalised and rotated (v untouched)."""
        out = _req(qkv, torch.bfloat16, "qkv").clone()
        ws = [_req(w, torch.bfloat16, "norm weight") for w in (qw0, kw0, qw1, kw1)]
        ids = _req(ids, torch.int32, "ids")
        self._enter()
        check(lib().flux2mi_op_qk_norm_rope(self.handle, _ptr(out), out.shape[0], heads, split, *[_ptr(w) for w in ws], _ptr(ids), q_mul))
        self._exit()
        return out

    def op_qk_proj_norm_rope(self, x, w, qw, kw, ids, q_mul=1.0) -> torch.Tensor:
        x, w = _req(x, torch.bfloat16, "x"), _req(w, torch.bfloat16, "w")
        qw, kw, ids = _req(qw, torch.bfloat16, "qw"), _req(kw, torch.bfloat16, "kw"), _req(ids, torch.int32, "ids")
        s, d = x.shape
        out = torch.empty((s, 2 * d), dtype=torch.bfloat16, device=x.device)
        self._enter()
        check(lib().flux2mi_op_qk_proj_norm_rope(self.handle, _ptr(x), _ptr(w), s, d, _ptr(qw), _ptr(kw), _ptr(ids), q_mul, _ptr(out)))
        self._exit()
        return out

    def op_groupnorm_silu(self, x_nhwc, groups, gamma, beta, eps=1e-6, silu=True) -> torch.Tensor:
        x = _req(x_nhwc, torch.bfloat16, "x")
        gamma, beta = _req(gamma, torch.float32, "gamma"), _req(beta, torch.float32, "beta")
        c = x.shape[-1]
        y = torch.empty_like(x)
        self._enter()
        check(lib().flux2mi_op_groupnorm_silu(self.handle, _ptr(x), _ptr(y), x.numel() // c, c, groups, _ptr(gamma), _ptr(beta), eps, int(silu)))
        self._exit()
        return y

    def op_attention(self, qkv: torch.Tensor, heads: int) -> torch.Tensor:
        qkv = _req(qkv, torch.bfloat16, "qkv")
        s = qkv.shape[0]
        out = torch.empty((s, heads * 128), dtype=torch.bfloat16, device=qkv.device)
        self._enter()
        check(lib().flux2mi_op_attention(self.handle, _ptr(qkv), _ptr(out), s, heads))
        self._exit()
        return out

    def op_attention_prescaled(self, qkv: torch.Tensor, heads: int, score_bound_log2: float) -> torch.Tensor:
        """qkv bf16 [S, 3*heads*128] with q already multiplied by log2(e)/sqrt(128); the blocks' form of the attention."""
        qkv = _req(qkv, torch.bfloat16, "qkv")
        s = qkv.shape[0]
        out = torch.empty((s, heads * 128), dtype=torch.bfloat16, device=qkv.device)
        self._enter()
        check(lib().flux2mi_op_attention_prescaled(self.handle, _ptr(qkv), _ptr(out), s, heads, float(score_bound_log2)))
        self._exit()
        return out

    def op_attention_split_info(self, S: int, heads: int):
        """(KV pieces per item of the partial last round — 1 = whole items only —, compute units) of the blocks' attention on this GPU."""
        f, n = C.c_int(0), C.c_int(0)
        check(lib().flux2mi_op_attention_split_info(self.handle, S, heads, C.byref(f), C.byref(n)))
        return int(f.value), int(n.value)

    def op_attention_qk8(self, q8: torch.Tensor, k8: torch.Tensor, v: torch.Tensor, heads: int, q_exp: int, k_exp: int, score_bound_log2: float) -> torch.Tensor:
        """q8 / k8: e4m3 bytes (torch.float8_e4m3fn or uint8) [S, heads*128], q8 = q 2^q_exp with q pre-scaled by log2(e)/sqrt(128),
        k8 = k 2^k_exp; v bf16 [S, heads*128]: the fp8 models' attention."""
        s = v.shape[0]
        q8 = q8.view(torch.uint8).contiguous()
        k8 = k8.view(torch.uint8).contiguous()
        v = _req(v, torch.bfloat16, "v")
        out = torch.empty((s, heads * 128), dtype=torch.bfloat16, device=v.device)
        self._enter()
        check(lib().flux2mi_op_attention_qk8(self.handle, _ptr(q8), _ptr(k8), _ptr(v), _ptr(out), s, heads, q_exp, k_exp, float(score_bound_log2)))
        self._exit()
        return out

    def op_attention_qk8_blocks(self, q8, k8, v, heads: int, q_exp: int, k_exp: int, score_bound_log2: float):
        """op_attention_qk8 with the output as block-scaled e4m3: returns (bytes uint8 [S, heads*128], scale bytes uint8 [heads, ceil(S/16)]);
        value = e4m3(byte) * 2^(scale - 127), one scale per 16 queries x one head's 128 columns."""
        s = v.shape[0]
        q8 = q8.view(torch.uint8).contiguous()
        k8 = k8.view(torch.uint8).contiguous()
        v = _req(v, torch.bfloat16, "v")
        out8 = torch.empty((s, heads * 128), dtype=torch.uint8, device=v.device)
        nrb = (s + 15) // 16
        bs = torch.full((heads, nrb), 127, dtype=torch.uint8, device=v.device)
        self._enter()
        check(lib().flux2mi_op_attention_qk8_blocks(self.handle, _ptr(q8), _ptr(k8), _ptr(v), None, _ptr(out8), _ptr(bs), nrb, s, heads, q_exp, k_exp,
                                                    float(score_bound_log2)))
        self._exit()
        return out8, bs

    def op_attention_prescaled_range(self, qkv: torch.Tensor, heads: int, score_bound_log2: float, q0: int, nq: int, nk: int) -> torch.Tensor:
        """Queries [q0, q0+nq) against the key prefix [0, nk) of the full V^T image (the KV-extraction pass's launch shape);
        rows outside the query range come back zero."""
        qkv = _req(qkv, torch.bfloat16, "qkv")
        s = qkv.shape[0]
        out = torch.zeros((s, heads * 128), dtype=torch.bfloat16, device=qkv.device)
        self._enter()
        check(lib().flux2mi_op_attention_prescaled_range(self.handle, _ptr(qkv), _ptr(out), s, heads, float(score_bound_log2), q0, nq, nk))
        self._exit()
        return out


def _set_tensors(fn, handle, tensors: Dict[str, torch.Tensor]):
    for name, t in tensors.items():
        if t.dtype not in _DT:
            t = t.float()
        t = t.contiguous()
        if t.is_cuda:   # the library copies on its own stream (and synchronises it before returning): the producer must be done
            torch.cuda.current_stream(t.device).synchronize()
        shape = (C.c_int64 * max(t.dim(), 1))(*t.shape)
        check(fn(handle, name.encode(), _DT[t.dtype], shape, t.dim(), C.c_void_p(t.data_ptr())))


class MLXRandom:
    """The reference's global random state: `MLXRandom.seed(seed)` followed by keyless `MLXRandom.normal(shape)` /
    `uniform` draws (LatentUtils.swift:32-41, Flux2Pipeline.swift:1203-1204, 2313, 2403).  Each draw splits the
    state key on the host (two Threefry hashes) and generates on the device with the sub-key, so a given seed yields
    the stream MLX yields (include/flux2mi.h, "MLX-compatible random stream")."""

    def __init__(self, ctx: "Context", seed: int = 0):
        self.ctx = ctx
        self.state = (C.c_uint32 * 2)()
        self.seed(seed)

    def seed(self, seed: int):
        lib().flux2mi_mlx_random_key(C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), self.state)

    def next_key(self):
        first, second = (C.c_uint32 * 2)(), (C.c_uint32 * 2)()
        lib().flux2mi_mlx_random_split(self.state, first, second)
        self.state = first
        return second

    def _draw(self, shape, fn, dtype=torch.float32, *extra):
        out = torch.empty(tuple(shape), dtype=dtype, device=self.ctx.device)
        key = self.next_key()
        self.ctx._enter()
        check(fn(self.ctx.handle, key, _ptr(out), out.numel(), *extra))
        self.ctx._exit()
        return out

    def bits(self, shape) -> torch.Tensor:
        """uint32 words, returned as int32 bit patterns (torch has no uint32 arithmetic)."""
        return self._draw(shape, lib().flux2mi_mlx_random_bits, torch.int32)

    def uniform(self, shape, low: float = 0.0, high: float = 1.0) -> torch.Tensor:
        return self._draw(shape, lib().flux2mi_mlx_random_uniform, torch.float32, float(low), float(high))

    def normal(self, shape) -> torch.Tensor:
        return self._draw(shape, lib().flux2mi_mlx_random_normal)

    def normal_array(self, shape) -> "MLXArray":
        """`MLXRandom.normal(shape)` as the façade type (same key split, same values as `normal`)."""
        h = C.c_void_p()
        sh = (C.c_int64 * max(len(shape), 1))(*shape)
        key = self.next_key()
        self.ctx._enter()
        check(lib().flux2mi_array_random_normal(self.ctx.handle, key, sh, len(shape), C.byref(h)))
        self.ctx._exit()
        return MLXArray(self.ctx, _handle=h)


class MLXArray:
    """What the Swift shim's `MLXArray` is (include/flux2mi.h, "MLXArray façade"): a handle to a flux2mi_array, with
    exactly the MLX surface code outside Flux2Core uses on hook / chain tensors — `MLXArray(Float)`, `+ - *` with
    broadcasting (Python scalars promote like Swift literals do), `.shape`, `eval`.  Every operator is one C-ABI call."""

    def __init__(self, ctx: "Context", value=None, *, _handle=None, _keep=None):
        self.ctx = ctx
        self._keep = _keep          # memory a view depends on
        if _handle is not None:
            self.handle = _handle
            return
        h = C.c_void_p()
        if isinstance(value, (int, float)):
            check(lib().flux2mi_array_from_scalar(ctx.handle, float(value), C.byref(h)))
        else:
            t = torch.as_tensor(value, dtype=torch.float32).cpu().contiguous()
            shape = (C.c_int64 * max(t.dim(), 1))(*t.shape)
            check(lib().flux2mi_array_from_host(ctx.handle, t.data_ptr(), shape, t.dim(), C.byref(h)))
        self.handle = h

    @classmethod
    def wrap(cls, ctx: "Context", t: torch.Tensor) -> "MLXArray":
        """Zero-copy view of an f32 device tensor (the hook's latents, `encodeImageToPackedSequence` results)."""
        t = _req(t, torch.float32, "tensor")
        h = C.c_void_p()
        shape = (C.c_int64 * max(t.dim(), 1))(*t.shape)
        check(lib().flux2mi_array_wrap(ctx.handle, t.data_ptr(), shape, t.dim(), C.byref(h)))
        return cls(ctx, _handle=h, _keep=t)

    @property
    def shape(self):
        n = lib().flux2mi_array_ndim(self.handle)
        return [int(lib().flux2mi_array_dim(self.handle, i)) for i in range(n)]

    def _bin(self, fn, a, b):
        a = a if isinstance(a, MLXArray) else MLXArray(self.ctx, a)
        b = b if isinstance(b, MLXArray) else MLXArray(self.ctx, b)
        h = C.c_void_p()
        self.ctx._enter()
        check(fn(self.ctx.handle, a.handle, b.handle, C.byref(h)))
        self.ctx._exit()
        return MLXArray(self.ctx, _handle=h, _keep=(a, b))   # operands stay alive until the result is released

    def __add__(self, o): return self._bin(lib().flux2mi_array_add, self, o)
    def __radd__(self, o): return self._bin(lib().flux2mi_array_add, o, self)
    def __sub__(self, o): return self._bin(lib().flux2mi_array_sub, self, o)
    def __rsub__(self, o): return self._bin(lib().flux2mi_array_sub, o, self)
    def __mul__(self, o): return self._bin(lib().flux2mi_array_mul, self, o)
    def __rmul__(self, o): return self._bin(lib().flux2mi_array_mul, o, self)

    def eval(self):
        self.ctx.sync()
        return self

    def torch(self) -> torch.Tensor:
        """A device tensor with a copy of the values (test aid)."""
        shape = self.shape
        out = torch.empty(shape, dtype=torch.float32, device=self.ctx.device)
        self.ctx._enter()
        check(lib().flux2mi_array_copy_to(self.ctx.handle, self.handle, out.data_ptr(), out.numel()))
        self.ctx._exit()
        return out

    def tolist(self):
        n = 1
        for d in self.shape:
            n *= d
        buf = (C.c_float * n)()
        check(lib().flux2mi_array_to_host(self.ctx.handle, self.handle, buf))
        return list(buf)

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                lib().flux2mi_array_free(self.handle)
                self.handle = None
        except Exception:
            pass


class DiT:
    """Flux2Transformer2DModel (flux2mi_dit)."""

    def __init__(self, ctx: Context, config, weights: str = "bf16"):
        """`weights`: "bf16" (default) or "fp8" — e4m3 block-GEMM weights + per-token fp8 activations
        (the MI355X counterpart of the reference's quantised transformer variants,
        Configuration/QuantizationConfig.swift:40-73)."""
        self.ctx = ctx
        if weights not in ("bf16", "fp8"):
            raise ValueError(f"weights must be 'bf16' or 'fp8', got {weights!r}")
        cfg = L.DitConfig()
        if isinstance(config, str):
            check(lib().flux2mi_dit_preset(config.encode(), C.byref(cfg)))
        else:
            cfg.dim, cfg.heads = config["dim"], config["heads"]
            cfg.n_double, cfg.n_single = config["n_double"], config["n_single"]
            cfg.joint_dim, cfg.guidance_embeds = config["joint_dim"], int(bool(config["guidance"]))
        cfg.weight_format = 1 if weights == "fp8" else 0
        self.cfg = cfg
        h = C.c_void_p()
        check(lib().flux2mi_dit_create(ctx.handle, C.byref(cfg), C.byref(h)))
        self.handle = h
        ctx._children.append(weakref.ref(self))

    def load(self, tensors: Dict[str, torch.Tensor]):
        """Checkpoint tensors by key (BFL or diffusers naming), host or device memory."""
        _set_tensors(lib().flux2mi_dit_set_tensor, self.handle, tensors)
        return self

    def finalize(self):
        check(lib().flux2mi_dit_finalize(self.handle))
        return self

    def load_safetensors(self, path: str):
        """Native reader (csrc/loader.cpp): BFL or diffusers keys, bf16/f16/f32, quanto qint8 pairs.
        Returns (tensors ingested, tensors skipped)."""
        n, k = C.c_int(0), C.c_int(0)
        self.ctx._enter()
        check(lib().flux2mi_dit_load_safetensors(self.handle, str(path).encode(), C.byref(n), C.byref(k)))
        self.ctx._exit()
        return n.value, k.value

    def export_prequantized(self, path: str, source_fingerprint: str = ""):
        """`Flux2PrequantizedCheckpoint.save`: the resident (finalized) weights in the library's own layout, written
        atomically; see include/flux2mi.h."""
        self.ctx._enter()
        check(lib().flux2mi_dit_export_prequantized(self.handle, str(path).encode(), source_fingerprint.encode()))
        self.ctx._exit()
        return path

    def load_prequantized(self, path: str, expected_fingerprint: Optional[str] = None) -> bool:
        """`Flux2PrequantizedCheckpoint.load(into:)`: validates before touching the model (raises Flux2Error and leaves
        the model as it was), finalizes on success.  Returns True when the export was taken with a LoRA merged in."""
        baked = C.c_int(0)
        fp = None if expected_fingerprint is None else expected_fingerprint.encode()
        self.ctx._enter()
        check(lib().flux2mi_dit_load_prequantized(self.handle, str(path).encode(), fp, C.byref(baked)))
        self.ctx._exit()
        return bool(baked.value)

    def set_tensor_q8(self, name: str, data: torch.Tensor, scale: torch.Tensor):
        """One quanto qint8 tensor: int8 [out,in] x f32 scale [out] (WeightLoader.swift:347-363)."""
        d = data.contiguous()
        sc = scale.reshape(-1).to(torch.float32).contiguous()
        assert d.dtype == torch.int8 and d.dim() == 2 and sc.numel() == d.shape[0]
        check(lib().flux2mi_dit_set_tensor_q8(self.handle, name.encode(), d.data_ptr(), sc.data_ptr(), d.shape[0], d.shape[1]))
        return self

    def merge_lora(self, name: str, lora_a: torch.Tensor, lora_b: torch.Tensor, scale: float):
        """W[name] += scale * (B A); A [rank,in], B [out,rank], each in its own float dtype (WeightLoader.mergeLoRAWeights).
        The factors' shapes are checked against the weight by the library (Flux2MIError SHAPE for another model's LoRA)."""
        a, b = lora_a.contiguous(), lora_b.contiguous()
        dts = {torch.float32: L.F32, torch.bfloat16: L.BF16, torch.float16: L.F16}
        assert a.dim() == 2 and b.dim() == 2
        check(lib().flux2mi_dit_merge_lora_checked(self.handle, name.encode(), dts[a.dtype], a.data_ptr(), a.shape[0], a.shape[1],
                                                   dts[b.dtype], b.data_ptr(), b.shape[0], b.shape[1], float(scale)))
        return self

    def slot_shape(self, name: str):
        """[out, in] of the weight a checkpoint key addresses."""
        r, c = C.c_int64(0), C.c_int64(0)
        check(lib().flux2mi_dit_slot_shape(self.handle, name.encode(), C.byref(r), C.byref(c)))
        return int(r.value), int(c.value)

    @property
    def param_bytes(self) -> int:
        return int(lib().flux2mi_dit_param_bytes(self.handle))

    def set_attention_softmax(self, mode: str):
        """"auto": fixed-offset softmax in the blocks whose QK-RMSNorm weights bound the scores; "online": no bound assumed
        anywhere — the per-query data-offset form of the same kernel (what blocks with larger norm scales get anyway)."""
        check(lib().flux2mi_dit_set_attention_softmax(self.handle, {"auto": 0, "online": 1}[mode]))
        return self

    def attention_softmax_paths(self) -> dict:
        fx, on = C.c_int(0), C.c_int(0)
        check(lib().flux2mi_dit_attention_softmax_info(self.handle, C.byref(fx), C.byref(on)))
        return {"fixed_offset_blocks": fx.value, "online_blocks": on.value}

    def forward(self, latents: torch.Tensor, txt_emb: torch.Tensor, timestep: float, guidance: float,
                img_ids: torch.Tensor, txt_ids: torch.Tensor, s_out: Optional[int] = None) -> torch.Tensor:
        """latents f32 [S_img,128], txt_emb bf16 [S_txt,J], ids i32 [.,4] -> f32 [s_out,128] (async)."""
        lat = _req(latents.reshape(-1, 128), torch.float32, "latents")
        txt = _req(txt_emb.reshape(-1, txt_emb.shape[-1]), torch.bfloat16, "txt_emb")
        ii, ti = _req(img_ids, torch.int32, "img_ids"), _req(txt_ids, torch.int32, "txt_ids")
        s_img, s_txt = lat.shape[0], txt.shape[0]
        s_out = s_img if s_out is None else s_out
        out = torch.empty((s_out, 128), dtype=torch.float32, device=lat.device)
        self.ctx._enter()
        check(lib().flux2mi_dit_forward(self.handle, _ptr(lat), s_img, _ptr(txt), s_txt, timestep, guidance,
                                        _ptr(ii), _ptr(ti), _ptr(out), s_out))
        self.ctx._exit()
        return out

    def forward_block(self, layer: int, hs_in: Optional[torch.Tensor], txt_emb: torch.Tensor, timestep: float, guidance: float,
                      img_ids: torch.Tensor, txt_ids: torch.Tensor, with_final: bool = False, latents: Optional[torch.Tensor] = None):
        """ONE block on a given residual stream: hs_in f32 [S_txt + S_img, d] (text rows first) -> f32 of the same shape.
        layer: double blocks first, then single blocks (flux2mi_dit_forward_block; teacher-forced parity tests).  with_final: also
        the output layer applied to the result, f32 [S_img, 128].  hs_in=None + latents: OBSERVE the whole forward's own stream behind
        block `layer` (and, with_final, its prediction)."""
        txt = _req(txt_emb.reshape(-1, txt_emb.shape[-1]), torch.bfloat16, "txt_emb")
        ii, ti = _req(img_ids, torch.int32, "img_ids"), _req(txt_ids, torch.int32, "txt_ids")
        s_txt = txt.shape[0]
        if hs_in is not None:
            hs = _req(hs_in.reshape(-1, hs_in.shape[-1]), torch.float32, "hs_in")
            lat, s_img = None, hs.shape[0] - s_txt
            out = torch.empty_like(hs)
        else:
            hs, lat = None, _req(latents.reshape(-1, 128), torch.float32, "latents")
            s_img = lat.shape[0]
            out = torch.empty((s_txt + s_img, self.cfg.dim), dtype=torch.float32, device=lat.device)
        fin = torch.empty((s_img, 128), dtype=torch.float32, device=out.device) if with_final else None
        self.ctx._enter()
        check(lib().flux2mi_dit_forward_block(self.handle, layer, _ptr(hs), _ptr(lat), _ptr(out), s_img, _ptr(txt), s_txt, timestep, guidance,
                                              _ptr(ii), _ptr(ti), _ptr(fin)))
        self.ctx._exit()
        return (out, fin) if with_final else out

    def last_modulation(self) -> dict:
        """temb and the modulation vectors of the last forward (flux2mi_dit_last_modulation), as f32 CPU tensors."""
        d = self.cfg.dim
        buf = torch.empty(256 + 19 * d, dtype=torch.float32, device=self.ctx.device)
        self.ctx._enter()
        check(lib().flux2mi_dit_last_modulation(self.handle, _ptr(buf), buf.numel()))
        self.ctx._exit()
        b = buf.cpu()
        o = 256 + d
        return {"sinusoid": b[:256], "temb": b[o:o + d], "img": b[o + d:o + 7 * d], "txt": b[o + 7 * d:o + 13 * d], "single": b[o + 13 * d:o + 16 * d],
                "final": b[o + 16 * d:o + 18 * d]}

    def forward_kv_extract(self, cache: "KVCache", latents, ref_latents, txt_emb, timestep: float, guidance: float,
                           img_ids, txt_ids, ref_ids) -> torch.Tensor:
        """`forwardKVExtract` (Flux2Transformer.swift:346-446): fills `cache`, returns f32 [S_out,128]."""
        lat = _req(latents.reshape(-1, 128), torch.float32, "latents")
        ref = _req(ref_latents.reshape(-1, 128), torch.float32, "ref_latents")
        txt = _req(txt_emb.reshape(-1, txt_emb.shape[-1]), torch.bfloat16, "txt_emb")
        ii, ti, ri = (_req(x, torch.int32, n) for x, n in ((img_ids, "img_ids"), (txt_ids, "txt_ids"), (ref_ids, "ref_ids")))
        out = torch.empty((lat.shape[0], 128), dtype=torch.float32, device=lat.device)
        self.ctx._enter()
        check(lib().flux2mi_dit_forward_kv_extract(self.handle, cache.handle, _ptr(lat), lat.shape[0], _ptr(ref), ref.shape[0],
                                                   _ptr(txt), txt.shape[0], timestep, guidance, _ptr(ii), _ptr(ti), _ptr(ri),
                                                   _ptr(out)))
        self.ctx._exit()
        return out

    def forward_kv_cached(self, cache: "KVCache", latents, txt_emb, timestep: float, guidance: float, img_ids,
                          txt_ids) -> torch.Tensor:
        """`forwardKVCached` (Flux2Transformer.swift:459-546)."""
        lat = _req(latents.reshape(-1, 128), torch.float32, "latents")
        txt = _req(txt_emb.reshape(-1, txt_emb.shape[-1]), torch.bfloat16, "txt_emb")
        ii, ti = _req(img_ids, torch.int32, "img_ids"), _req(txt_ids, torch.int32, "txt_ids")
        out = torch.empty((lat.shape[0], 128), dtype=torch.float32, device=lat.device)
        self.ctx._enter()
        check(lib().flux2mi_dit_forward_kv_cached(self.handle, cache.handle, _ptr(lat), lat.shape[0], _ptr(txt), txt.shape[0],
                                                  timestep, guidance, _ptr(ii), _ptr(ti), _ptr(out)))
        self.ctx._exit()
        return out

    def close(self):
        if self.handle and self.ctx.handle:
            lib().flux2mi_dit_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class KVCache:
    """`TransformerKVCache`: reference-token K/V of every block (flux2mi_kv_cache)."""

    def __init__(self, dit: DiT, s_txt: int, s_ref: int):
        self.dit = dit
        h = C.c_void_p()
        check(lib().flux2mi_kv_cache_create(dit.handle, s_txt, s_ref, C.byref(h)))
        self.handle = h
        self.s_txt, self.s_ref = s_txt, s_ref

    def entry(self, layer: int):
        """`doubleStreamEntry(at:)` / `singleStreamEntry(at:)` (layer = double blocks, then single blocks): (keys bf16 [s_ref, dim]
        post-RoPE, values bf16 [s_ref, dim])."""
        d = self.dit.cfg.dim
        keys = torch.empty((self.s_ref, d), dtype=torch.bfloat16, device=self.dit.ctx.device)
        vt = torch.empty((d, self.s_ref), dtype=torch.bfloat16, device=self.dit.ctx.device)
        self.dit.ctx._enter()
        check(lib().flux2mi_kv_cache_entry(self.handle, layer, _ptr(keys), _ptr(vt)))
        self.dit.ctx._exit()
        return keys, vt.t().contiguous()

    def close(self):
        if self.handle and self.dit.handle and self.dit.ctx.handle:
            lib().flux2mi_kv_cache_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class VAE:
    """AutoencoderKLFlux2 decoder side (flux2mi_vae)."""

    def __init__(self, ctx: Context, config):
        self.ctx = ctx
        cfg = L.VaeConfig()
        if isinstance(config, str):
            check(lib().flux2mi_vae_preset(config.encode(), C.byref(cfg)))
        else:
            for i, c in enumerate(config["channels"]):
                cfg.channels[i] = c
            cfg.latent_channels, cfg.groups = config["latent"], config["groups"]
            cfg.layers_per_block = config["layers_per_block"]
        self.cfg = cfg
        h = C.c_void_p()
        check(lib().flux2mi_vae_create(ctx.handle, C.byref(cfg), C.byref(h)))
        self.handle = h
        ctx._children.append(weakref.ref(self))

    def load(self, tensors: Dict[str, torch.Tensor]):
        _set_tensors(lib().flux2mi_vae_set_tensor, self.handle, tensors)
        return self

    def finalize(self):
        check(lib().flux2mi_vae_finalize(self.handle))
        return self

    def _bn(self, fn) -> int:
        return fn(self.handle)

    @property
    def bn_mean_ptr(self) -> int:
        return lib().flux2mi_vae_bn_mean(self.handle)

    @property
    def bn_var_ptr(self) -> int:
        return lib().flux2mi_vae_bn_var(self.handle)

    def decode(self, z: torch.Tensor, want_f32: bool = True, want_u8: bool = True):
        """z f32 [1,32,h,w] -> (f32 [1,3,8h,8w] or None, u8 [8h,8w,3] or None)  (async)."""
        z = _req(z, torch.float32, "z")
        h, w = z.shape[-2], z.shape[-1]
        of = torch.empty((1, 3, 8 * h, 8 * w), dtype=torch.float32, device=z.device) if want_f32 else None
        ou = torch.empty((8 * h, 8 * w, 3), dtype=torch.uint8, device=z.device) if want_u8 else None
        self.ctx._enter()
        check(lib().flux2mi_vae_decode(self.handle, _ptr(z), h, w, _ptr(of), _ptr(ou)))
        self.ctx._exit()
        return of, ou

    def load_safetensors(self, path: str):
        n, k = C.c_int(0), C.c_int(0)
        self.ctx._enter()
        check(lib().flux2mi_vae_load_safetensors(self.handle, str(path).encode(), C.byref(n), C.byref(k)))
        self.ctx._exit()
        return n.value, k.value

    @property
    def has_encoder(self) -> bool:
        return bool(lib().flux2mi_vae_has_encoder(self.handle))

    def encode(self, image: torch.Tensor) -> torch.Tensor:
        """`vae.encode(x, samplePosterior: false)`: image f32 [1,3,H,W] in [-1,1] -> posterior mean
        f32 [1,32,H/8,W/8]  (async; no scaling factor, no BatchNorm — AutoencoderKL.swift:90-123)."""
        x = _req(image, torch.float32, "image")
        H, W = x.shape[-2], x.shape[-1]
        z = torch.empty((1, 32, H // 8, W // 8), dtype=torch.float32, device=x.device)
        self.ctx._enter()
        check(lib().flux2mi_vae_encode(self.handle, _ptr(x), H, W, _ptr(z)))
        self.ctx._exit()
        return z

    def encode_to_packed(self, image: torch.Tensor) -> torch.Tensor:
        """encode -> patchify -> BatchNorm-normalise -> pack: f32 [1,(H/16)(W/16),128]."""
        z = self.encode(image)
        h16, w16 = z.shape[-2] // 2, z.shape[-1] // 2
        out = torch.empty((1, h16 * w16, 128), dtype=torch.float32, device=z.device)
        self.ctx._enter()
        check(lib().flux2mi_vae_latents_to_packed(self.ctx.handle, _ptr(z), self.bn_mean_ptr, self.bn_var_ptr,
                                                  _ptr(out), h16, w16))
        self.ctx._exit()
        return out

    def close(self):
        if self.handle and self.ctx.handle:
            lib().flux2mi_vae_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
