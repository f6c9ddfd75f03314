/* flux2mi.h — C ABI of libflux2mi.so: the MI355X (gfx950) implementation of the
 * Flux.2 denoise + VAE-decode hot path.
 *
 * The reference (VincentGourbin/flux-2-swift-mlx v2.4.0) has no FFI seam on this
 * path: it sits behind the Swift class `Flux2Pipeline` and every tensor is an
 * `MLXArray`.  This header is what a Swift shim module binds (see INTEGRATION.md;
 * precedent for Swift -> C-ABI calls in the reference:
 * Sources/Flux2Core/Utils/MLXCheckpoint.swift:23-118).  Each entry point cites the
 * reference interface it replaces (paths relative to the reference's Sources/).
 *
 * Conventions
 *   - plain C11: opaque handles, pointers and sizes; no C++/torch types.
 *   - every function returns int status (FLUX2MI_OK == 0); flux2mi_last_error()
 *     gives the thread-local message.  Nothing aborts the process.
 *   - "device pointer" = HIP device memory of the context's GPU; the caller owns
 *     input/output buffers, the library owns weights and workspace.
 *   - one context = one GPU + one HIP stream = one generation in flight
 *     (Flux2Pipeline is one-generation-at-a-time as well: Flux2Pipeline.swift:135).
 *     All device work is enqueued on the context's stream; functions documented as
 *     "async" return before the GPU finishes (flux2mi_ctx_sync to wait).
 *   - batch is 1 per forward like the reference (Flux2Pipeline.swift:1937); batch
 *     data-parallelism is one image per GPU/process.
 */
#ifndef FLUX2MI_H
#define FLUX2MI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FLUX2MI_VERSION 100 /* 0.1.0 */

enum flux2mi_status {
    FLUX2MI_OK = 0,
    FLUX2MI_ERR_INVALID_ARG = 1,   /* Flux2Error.invalidConfiguration (Flux2Core.swift:14-40) */
    FLUX2MI_ERR_HIP = 2,           /* HIP runtime failure */
    FLUX2MI_ERR_NOT_LOADED = 3,    /* Flux2Error.modelNotLoaded */
    FLUX2MI_ERR_SHAPE = 4,         /* shape mismatch (e.g. initLatents, Flux2Pipeline.swift:1898-1901) */
    FLUX2MI_ERR_CANCELLED = 5,     /* Flux2Error.generationCancelled (Flux2Pipeline.swift:1940-1942) */
    FLUX2MI_ERR_UNSUPPORTED = 6,
    FLUX2MI_ERR_OOM = 7,           /* Flux2Error.insufficientMemory */
    FLUX2MI_ERR_UNKNOWN_KEY = 8    /* set_tensor / merge_lora: the model has no tensor of that name (a key of another
                                      component or of a layer this path does not run) — callers may skip it; every
                                      other argument error stays FLUX2MI_ERR_INVALID_ARG */
};

enum flux2mi_dtype { FLUX2MI_F32 = 0, FLUX2MI_BF16 = 1, FLUX2MI_F16 = 2 };

typedef struct flux2mi_ctx flux2mi_ctx;
typedef struct flux2mi_dit flux2mi_dit;
typedef struct flux2mi_vae flux2mi_vae;

int flux2mi_version(void);
const char* flux2mi_last_error(void);

/* ---- context ---------------------------------------------------------- */
/* `hip_stream` may be NULL (the library creates a non-blocking stream) or an
 * existing hipStream_t the caller wants the work ordered on. */
int flux2mi_ctx_create(int device, void* hip_stream, flux2mi_ctx** out);
int flux2mi_ctx_destroy(flux2mi_ctx* ctx);
int flux2mi_ctx_sync(flux2mi_ctx* ctx);                 /* == MLX eval(): stream synchronise */
/* MLX.Memory.activeMemory / peakMemory / cacheMemory (mlx-swift 0.31.6 `Memory`; called at Flux2CLI/InpaintCommand.swift:253-255 and
 * CompareEncoders.swift:112,142).  Device bytes the library holds on the context's GPU, from its own allocation table (every
 * allocation of the library is accounted; the caller's own hipMallocs are not): `active` = weights, flux2mi_array / flux2mi_buffer
 * objects, KV caches and derived weights; `workspace` = the grow-only activation / staging workspaces kept between calls for reuse
 * (MLX's "cache": memory held but given back on request); `peak` = the largest active + workspace seen since the process started or
 * since flux2mi_ctx_reset_peak (MLX: `Memory.peakMemory = 0`).  Any of the three pointers may be NULL. */
int flux2mi_ctx_memory_stats(flux2mi_ctx* ctx, size_t* active_bytes, size_t* peak_bytes, size_t* workspace_bytes);
int flux2mi_ctx_reset_peak(flux2mi_ctx* ctx);
/* MLX.Memory.clearCache() (Flux2CLI/InpaintCommand.swift:271): synchronises the device and releases every workspace of this GPU;
 * the next forward / decode reserves what it needs again (results unchanged).  *freed_bytes (may be NULL) = bytes given back.
 * Threading: callable from any thread at any time, like MLX's.  Every entry point of this header that reserves or hands out workspace
 * memory (generate, forwards, decode / encode, text encoders, ingestion / finalize / LoRA merge, imaging and flux2mi_op_* calls,
 * destroy) registers itself as a user of its GPU's workspaces for the duration of the call: a trim that finds a user on the GPU —
 * another thread's call, or the generation whose hook called it — or another trim in progress releases nothing and returns OK with
 * 0 bytes; while a trim releases, calls that want the workspaces wait for it to finish.  (One context is still one call at a time:
 * the guard protects the workspaces against trims, it does not make a context re-entrant.) */
int flux2mi_ctx_trim(flux2mi_ctx* ctx, size_t* freed_bytes);
/* Plain device buffers for a host without a HIP binding of its own (the Swift shim: prompt embeddings, token ids, image bytes).
 * upload: the host bytes may be reused on return; download: behind everything queued on the stream, synchronises. */
int flux2mi_buffer_alloc(flux2mi_ctx* ctx, size_t bytes, void** out);
int flux2mi_buffer_free(flux2mi_ctx* ctx, void* device_ptr);
int flux2mi_buffer_upload(flux2mi_ctx* ctx, void* dst_device, const void* src_host, size_t bytes);
int flux2mi_buffer_download(flux2mi_ctx* ctx, void* dst_host, const void* src_device, size_t bytes);

/* ---- batch-level data parallelism over RCCL / xGMI (the reference has no distributed code, SURVEY F2 — this
 * interface is the build's own).  One process per GPU, image i -> rank i mod world, a full replica per GPU, NO collective inside
 * the denoise loop: per batch one broadcast of the prompt embeddings (the output of flux2mi_textenc_forward on `root`, i.e. what
 * Flux2Pipeline.swift:1381-1404 computes once per prompt) and one all-gather of the decoded images (:2095, postprocessVAEOutput).
 * Both are enqueued on the context's stream, in stream order with flux2mi_generate — nothing to hand over, no host sync.
 * The 128-byte id is made on rank 0 and carried to the other ranks by the host (TCP store, file, environment). */
#define FLUX2MI_DP_UNIQUE_ID_BYTES 128
/* RCCL is bound by dlopen at the first flux2mi_dp_* call: $FLUX2MI_RCCL_LIB, else the librccl.so.1 the process has already loaded
 * (a host running torch.distributed: one RCCL per process), else $ROCM_PATH/lib, /opt/rocm/lib, the loader's search path.
 * Returns the file that was bound (NULL: none could be; the flux2mi_dp_* calls then return FLUX2MI_ERR_UNSUPPORTED). */
const char* flux2mi_dp_rccl_library(void);
int flux2mi_dp_unique_id(uint8_t* id /* [FLUX2MI_DP_UNIQUE_ID_BYTES] */);
int flux2mi_dp_init(flux2mi_ctx* ctx, int rank, int world, const uint8_t* id);
int flux2mi_dp_finalize(flux2mi_ctx* ctx);               /* also done by flux2mi_ctx_destroy */
int flux2mi_dp_rank(const flux2mi_ctx* ctx);             /* 0 without a communicator */
int flux2mi_dp_world(const flux2mi_ctx* ctx);            /* 1 without a communicator */
/* bf16 [n_images, s_txt, joint_dim] device, in place, from `root` */
int flux2mi_dp_broadcast_embeddings(flux2mi_ctx* ctx, void* embeddings, int64_t elems, int root);
/* u8 device: local_images [bytes_per_rank] -> all_images [world][bytes_per_rank] (rank-major) */
int flux2mi_dp_all_gather_images(flux2mi_ctx* ctx, const uint8_t* local_images, uint8_t* all_images, int64_t bytes_per_rank);
/* max over the ranks of one host f64, in place (the bench's max-over-ranks time; doubles as a barrier); synchronises */
int flux2mi_dp_all_reduce_max(flux2mi_ctx* ctx, double* value);
/* Timings of the two exchanges for the bench line (SURVEY.md §8e "Measurability"; the reference has no counterpart: single process,
 * Flux2Pipeline.swift:135).  With timing on, every broadcast / all-gather is bracketed by HIP events on the context's stream;
 * _stats folds the finished ones (synchronises the stream) and returns per kind (index 0 broadcast, 1 all-gather) the summed
 * milliseconds, the number of calls and the payload bytes since the last reset; any output pointer may be NULL. */
int flux2mi_dp_collective_timing(flux2mi_ctx* ctx, int on);
int flux2mi_dp_collective_stats(flux2mi_ctx* ctx, double ms[2], int64_t calls[2], int64_t bytes[2], int reset);
void* flux2mi_ctx_stream(flux2mi_ctx* ctx);             /* hipStream_t */

/* ---- scheduler (host) -------------------------------------------------- */
/* FlowMatchEulerScheduler.swift:9-28 */
float flux2mi_compute_empirical_mu(int image_seq_len, int num_steps);
/* FlowMatchEulerScheduler.swift:65-115.  image_seq_len <= 0 selects the reference's
 * default (4096).  Writes the strength-sliced sigma table (length *n_out, terminal
 * 0 included) and tStart. */
int flux2mi_set_timesteps(int num_inference_steps, int image_seq_len, float strength,
                          float* sigmas_out, int capacity, int* n_out, int* t_start_out);
/* FlowMatchEulerScheduler.swift:236-260 (terminal 0 appended unless already 0). */
int flux2mi_set_custom_sigmas(const float* sigmas_in, int n_in, float* sigmas_out, int capacity,
                              int* n_out);

/* ---- latent utilities (host) -- Pipeline/LatentUtils.swift ------------- */
void flux2mi_validate_dimensions(int height, int width, int* out_height, int* out_width); /* :382-394 */
/* Pipeline/Flux2Pipeline+ChainHelpers.swift:112-129 */
void flux2mi_resolve_chain_dimensions(int width, int height, int max_pixels, int* out_height,
                                      int* out_width);
int flux2mi_image_position_ids(int height, int width, int32_t* out_host, int capacity_rows);   /* :256-286 */
int flux2mi_text_position_ids(int length, int32_t* out_host, int capacity_rows);               /* :291-298 */
int flux2mi_reference_position_ids(const int* latent_heights, const int* latent_widths, int n_images,
                                   int scale, int32_t* out_host, int capacity_rows);            /* :324-346 */

/* ---- elementwise device ops (async) ------------------------------------ */
/* FlowMatchEulerScheduler.step (:136-156): x += dt * v, dt = sigma_next - sigma. */
int flux2mi_sched_step(flux2mi_ctx* ctx, float* x, const float* v, float dt, int64_t n);
/* FlowMatchEulerScheduler.scaleNoise (:195-204): out = (1-sigma)*sample + sigma*noise. */
int flux2mi_scale_noise(flux2mi_ctx* ctx, float* out, const float* sample, const float* noise,
                        float sigma, int64_t n);
/* ---- pre-quantized checkpoint: Flux2PrequantizedCheckpoint + Flux2Pipeline.exportPrequantizedTransformer
 * (Loading/PrequantizedCheckpoint.swift, Pipeline/Flux2Pipeline.swift:759-840).  One safetensors file with the
 * resident weights in the library's own layout and naming ("<buffer>.weight" F8_E4M3 + "<buffer>.scale" F32 per
 * output row for the block GEMMs of a weight_format 1 model, BF16 otherwise) and string metadata {format,
 * quantization, bits, component, config, lora_baked, source_fingerprint}.  export: the model must be finalized;
 * written atomically (temp file + rename).  load: metadata (and the fingerprint when expected_fingerprint != NULL),
 * payload size, key set, dtypes and shapes are validated BEFORE the model is touched — on an error the model is
 * exactly as it was, so the standard load path can run; on success the model is finalized.  *lora_baked (optional)
 * reports an export taken after flux2mi_dit_merge_lora. */
int flux2mi_dit_export_prequantized(flux2mi_dit* dit, const char* path, const char* source_fingerprint);
int flux2mi_dit_load_prequantized(flux2mi_dit* dit, const char* path, const char* expected_fingerprint, int* lora_baked);

/* ---- pixel-space pre/post-processing of the inpainting / outpainting chains (device buffers, u8 interleaved
 * [H,W,C] images).  The reference resamples with CoreGraphics' closed `.high` filter; this library uses the
 * antialiased triangle filter in Pillow's 8-bit fixed-point form (bit-exact against Pillow's
 * Image.resize(BILINEAR, box=)), see DESIGN.md.  `box` = {x0, y0, x1, y1} source rectangle in pixels or NULL for
 * the whole image.  These calls synchronise the context's stream (they upload small host tables). */
enum { FLUX2MI_MASK_GRAYSCALE_WHITE_INPAINT = 0, FLUX2MI_MASK_ALPHA_TRANSPARENT_INPAINT = 1 }; /* Flux2MaskConvention */
/* CGContext.draw(image, in: rect) of renderRGBA (Flux2InpaintCompositing.swift:166-182): channels 1..4, each
 * filtered independently. */
int flux2mi_image_resample_u8(flux2mi_ctx* ctx, const uint8_t* src, int src_h, int src_w, int channels, const float* box,
                              uint8_t* dst, int dst_h, int dst_w);
/* renderInpaintWeights (Flux2InpaintCompositing.swift:196-227) and, with out = (H/16, W/16),
 * Flux2Pipeline.packMaskForLatentBlending (Flux2Pipeline+ChainHelpers.swift:148-236): mask rendered to out_h x out_w
 * in 8 bits, then grayscale v/255 (white = inpaint) or 1 - alpha/255 (transparent = inpaint; RGBA mask).
 * weights f32 [out_h*out_w]. */
int flux2mi_mask_inpaint_weights(flux2mi_ctx* ctx, const uint8_t* mask, int mask_h, int mask_w, int channels, int convention,
                                 const float* box, float* weights, int out_h, int out_w);
/* maskBoundingBox (:23-65): scan at <= 512 px, weights > threshold (reference default 8/255), rect {x,y,w,h} in image
 * space rounded outward by one scan pixel; *found = 0 for an empty mask. */
int flux2mi_mask_bounding_box(flux2mi_ctx* ctx, const uint8_t* mask, int mask_h, int mask_w, int channels, int convention,
                              int image_width, int image_height, float threshold, int out_rect[4], int* found);
/* expandCropRegion (:69-108, diffusers get_crop_region): pad, grow to the image aspect, shift inside, clamp. Host. */
void flux2mi_expand_crop_region(const int bbox[4], int padding, int image_width, int image_height, int out_rect[4]);
/* composite (:112-162, diffusers apply_overlay): base [full_h,full_w,base_channels] in place; generated
 * [crop_h,crop_w,gen_channels]; weights f32 [crop_h*crop_w] (1 = take the generated pixel).
 * FLUX2MI_ERR_SHAPE when the crop rect leaves the image (the reference returns nil). */
int flux2mi_composite_u8(flux2mi_ctx* ctx, uint8_t* base, int full_h, int full_w, int base_channels, const uint8_t* generated,
                         int gen_channels, const int crop_rect[4], const float* weights);
/* preprocessImageForVAE (Flux2Pipeline.swift:2359-2418): u8 [H,W,3|4] -> f32 [3,H,W] = x / 127.5 - 1. */
int flux2mi_preprocess_image_for_vae(flux2mi_ctx* ctx, const uint8_t* image, int height, int width, int channels, float* out);

/* Flux2OutpaintingChain.buildOutpaintCanvas (Flux2Chains/Flux2OutpaintingChain.swift:331-395): canvas u8
 * [canvas_h,canvas_w,3] = grey Gaussian noise (mean 128, sigma 32, SplitMix64 + Box-Muller seeded by noise_seed)
 * with the source [src_h,src_w,3|4] pasted at (offset_x, offset_y), top-left origin. */
int flux2mi_outpaint_canvas(flux2mi_ctx* ctx, const uint8_t* source, int src_h, int src_w, int channels, uint8_t* canvas,
                            int canvas_h, int canvas_w, int offset_x, int offset_y, uint64_t noise_seed);
/* Flux2OutpaintingChain.buildSmartMask (:397-474): u8 [canvas_h,canvas_w], 255 in the strips, 0 deep in the keep
 * rectangle, linear ramps of `transition_pixels` (capped at half the keep) inside the keep on the padded sides. */
int flux2mi_outpaint_smart_mask(flux2mi_ctx* ctx, uint8_t* mask, int canvas_h, int canvas_w, int keep_x, int keep_y, int keep_w,
                                int keep_h, int transition_pixels);

/* ---- MLX-compatible random stream.  The reference draws its noise with `MLXRandom.seed(seed)` +
 * `MLXRandom.normal(shape)` (LatentUtils.swift:32-41, Flux2Pipeline.swift:1203-1204, 2313, 2403): Threefry-2x32
 * keyed by {seed>>32, seed&0xffffffff}; every keyless draw first splits the global key (state, sub) = split(state)
 * and draws with `sub`.  Key bookkeeping is host integer work; draws are device kernels on the context's stream.
 * n <= 2^32-1 elements per draw. */
void flux2mi_mlx_random_key(uint64_t seed, uint32_t key[2]);
void flux2mi_mlx_random_split(const uint32_t key[2], uint32_t first[2], uint32_t second[2]);
int flux2mi_mlx_random_bits(flux2mi_ctx* ctx, const uint32_t key[2], uint32_t* out, int64_t n);
int flux2mi_mlx_random_uniform(flux2mi_ctx* ctx, const uint32_t key[2], float* out, int64_t n, float lo, float hi);
int flux2mi_mlx_random_normal(flux2mi_ctx* ctx, const uint32_t key[2], float* out, int64_t n);
/* ---- MLXArray façade ("MLXArray in the public API") -------------------------------------------------
 * `Flux2StepHook = @Sendable (Flux2StepContext, MLXArray) -> MLXArray` (Pipeline/Flux2Pipeline.swift:64),
 * `precomputedEmbeddings` / `initLatents` and the return values of `encodeImageToPackedSequence` /
 * `packMaskForLatentBlending` hand MLX arrays to code outside Flux2Core.  What that code does with them
 * (Flux2Chains/Flux2MaskedInpaintingChain.swift:380-403, Flux2CLI InpaintCommand.swift:253-271): `MLXArray(Float)`,
 * `+ - *` with broadcasting, `.shape`, `eval(_)`, `MLXRandom.seed / normal`.  A Swift shim's `MLXArray` wraps one
 * flux2mi_array: an immutable f32 device tensor of <= 4 dims; ops are enqueued on the context's stream and return new
 * arrays; `eval` = flux2mi_ctx_sync.  Each op rounds to f32 on its own (no cross-op FMA contraction), which makes the
 * closure of :399-403 executed through these entry points bit-identical to flux2mi_repaint_blend. */
typedef struct flux2mi_array flux2mi_array;
int flux2mi_array_alloc(flux2mi_ctx* ctx, const int64_t* shape, int ndim, flux2mi_array** out);     /* uninitialised */
int flux2mi_array_from_scalar(flux2mi_ctx* ctx, float value, flux2mi_array** out);                   /* MLXArray(Float), shape [] */
int flux2mi_array_from_host(flux2mi_ctx* ctx, const float* host, const int64_t* shape, int ndim, flux2mi_array** out);
/* non-owning view over device memory of the caller or the library (the hook's latent buffer, encoder outputs) */
int flux2mi_array_wrap(flux2mi_ctx* ctx, float* device, const int64_t* shape, int ndim, flux2mi_array** out);
int flux2mi_array_free(flux2mi_array* a);
int flux2mi_array_ndim(const flux2mi_array* a);                                                       /* .shape */
int64_t flux2mi_array_dim(const flux2mi_array* a, int i);
float* flux2mi_array_data(flux2mi_array* a);
int flux2mi_array_add(flux2mi_ctx* ctx, const flux2mi_array* a, const flux2mi_array* b, flux2mi_array** out);
int flux2mi_array_sub(flux2mi_ctx* ctx, const flux2mi_array* a, const flux2mi_array* b, flux2mi_array** out);
int flux2mi_array_mul(flux2mi_ctx* ctx, const flux2mi_array* a, const flux2mi_array* b, flux2mi_array** out);
/* the hook's returned array -> the library's latent buffer (no-op when it is the same memory) */
int flux2mi_array_copy_to(flux2mi_ctx* ctx, const flux2mi_array* src, float* dst_device, int64_t n);
int flux2mi_array_to_host(flux2mi_ctx* ctx, const flux2mi_array* a, float* host);                     /* synchronises */
/* `precomputedEmbeddings: MLXArray?` f32 -> the bf16 device buffer flux2mi_generate_params.txt_emb takes (Flux2Pipeline.swift:1166) */
int flux2mi_array_to_bf16(flux2mi_ctx* ctx, const flux2mi_array* a, void* dst_bf16_device);
/* MLXRandom.normal(shape) with an explicit key (see flux2mi_mlx_random_* for the key bookkeeping) */
int flux2mi_array_random_normal(flux2mi_ctx* ctx, const uint32_t key[2], const int64_t* shape, int ndim, flux2mi_array** out);

/* classical CFG, Flux2Pipeline.swift:1955-1973: v = v_uncond + g*(v_cond - v_uncond), in place on v_cond. */
int flux2mi_cfg_combine(flux2mi_ctx* ctx, float* v_cond, const float* v_uncond, float guidance, int64_t n);
/* RePaint hook body, Flux2Chains/Flux2MaskedInpaintingChain.swift:399-403:
 * x = (1-m)*((1-sigma_next)*img + sigma_next*noise) + m*x ; m is [S,1] broadcast over c. */
int flux2mi_repaint_blend(flux2mi_ctx* ctx, float* x, const float* image_latents, const float* blend_noise,
                          const float* mask, float sigma_next, int seq, int channels);
/* LatentUtils.packPatchifiedToSequence (:76-86): [C,h,w] -> [h*w,C]. */
int flux2mi_pack_patchified_to_sequence(flux2mi_ctx* ctx, const float* patchified, float* seq, int c, int h, int w);
/* LatentUtils.unpackSequenceToPatchified (:95-110): [h*w,C] -> [C,h,w]. */
int flux2mi_unpack_sequence_to_patchified(flux2mi_ctx* ctx, const float* seq, float* patchified, int c, int h, int w);
/* Flux2Pipeline.swift:2059-2079 fused: unpack -> denormalizeLatentsWithBatchNorm
 * (LatentUtils.swift:483-496, eps 1e-4) -> unpatchifyLatents (:119-142).
 * packed [h16*w16,128] -> z [32, 2*h16, 2*w16]. */
int flux2mi_latents_to_vae_input(flux2mi_ctx* ctx, const float* packed, const float* bn_mean,
                                 const float* bn_var, float* z, int h16, int w16);
/* Inverse (encodeImageToPackedSequence tail, +ChainHelpers.swift:90-99):
 * z [32,2*h16,2*w16] -> patchify -> BN-normalise -> packed [h16*w16,128]. */
int flux2mi_vae_latents_to_packed(flux2mi_ctx* ctx, const float* z, const float* bn_mean,
                                  const float* bn_var, float* packed, int h16, int w16);

/* ---- DiT (Flux2Transformer2DModel) -------------------------------------- */
typedef struct flux2mi_dit_config {
    int dim;              /* inner dim = heads * 128 */
    int heads;
    int n_double;         /* numLayers */
    int n_single;         /* numSingleLayers */
    int joint_dim;        /* jointAttentionDim */
    int guidance_embeds;  /* Dev: 1, Klein: 0 */
    int weight_format;    /* 0: bf16 weights and activations (default).  1: fp8 — the large GEMM weights are
                             stored as OCP e4m3 with one f32 scale per output channel and their
                             activations are quantised per token on the fly (BASELINE.json configs[3]);
                             replaces the reference's MLX weight-only quantisation
                             (Flux2Pipeline.swift:573, Configuration/QuantizationConfig.swift:40-73) */
} flux2mi_dit_config;

/* presets "klein-4b" | "klein-9b" | "dev" (Configuration/Flux2Config.swift:258-329). */
int flux2mi_dit_preset(const char* name, flux2mi_dit_config* out);
int flux2mi_dit_create(flux2mi_ctx* ctx, const flux2mi_dit_config* cfg, flux2mi_dit** out);
int flux2mi_dit_destroy(flux2mi_dit* dit);
/* Weight ingestion; replaces WeightLoader.mapTransformerWeights + applyTransformerWeights
 * (Loading/WeightLoader.swift:81-295 BFL keys, :397-480 diffusers keys).  `name` is the
 * checkpoint key in either scheme; `data` may be a host or a device pointer; dtype
 * f32/bf16/f16 is converted to bf16 storage.  Fused layouts (qkv, SwiGLU interleave)
 * are built here. */
int flux2mi_dit_set_tensor(flux2mi_dit* dit, const char* name, int dtype, const int64_t* shape, int ndim,
                           const void* data);
int flux2mi_dit_finalize(flux2mi_dit* dit);

/* ---- on-disk formats (Loading/WeightLoader.swift) ---------------------------------------------- */
/* Read a .safetensors file (BFL or diffusers key scheme, F32/BF16/F16, optional "transformer." prefix,
 * quanto qint8 "<key>._data" + "<key>._scale" pairs de-quantised on the device — WeightLoader.swift:24-78,
 * 306-363) and ingest every tensor this model knows; tensors of other components are skipped and
 * counted.  Call once per shard, then flux2mi_dit_finalize(). */
int flux2mi_dit_load_safetensors(flux2mi_dit* dit, const char* path, int* n_loaded, int* n_skipped);
/* One quanto qint8 tensor: data int8 [rows,cols], scale f32 [rows]; weight = data * scale (:347-363). */
int flux2mi_dit_set_tensor_q8(flux2mi_dit* dit, const char* name, const int8_t* data, const float* scale,
                              int64_t rows, int64_t cols);
/* LoRA merge into the resident bf16 weight: W += scale * (B A) (WeightLoader.mergeLoRAWeights :736-856, LoRA/LoRAConfig.swift:7-35;
 * scale = alpha/rank x user scale).  `name` = checkpoint key of the target weight.  Not available for weight_format 1.
 * _checked: A [a_rows = rank, a_cols = in] of dtype_a, B [b_rows = out, b_cols = rank] of dtype_b, shapes as the FILE states them;
 * FLUX2MI_ERR_SHAPE (naming the layer) unless they fit the addressed [out, in] weight — what the reference's matmul reports for a
 * LoRA of another model width.  The short form trusts A / B to have the target's own in / out (flux2mi_dit_slot_shape). */
int flux2mi_dit_merge_lora_checked(flux2mi_dit* dit, const char* name, int dtype_a, const void* A, int64_t a_rows, int64_t a_cols,
                                   int dtype_b, const void* B, int64_t b_rows, int64_t b_cols, float scale);
int flux2mi_dit_merge_lora(flux2mi_dit* dit, const char* name, int dtype, const void* A, const void* B, int rank,
                           float scale);
/* [out, in] of the weight a checkpoint key addresses (what `WeightLoader.mergeLoRAWeights` reads off `module.weight.shape`, :760-790);
 * FLUX2MI_ERR_UNKNOWN_KEY when the model has no tensor of that name */
int flux2mi_dit_slot_shape(flux2mi_dit* dit, const char* name, int64_t* rows, int64_t* cols);
/* `loadLoRA(LoRAConfig)`, weights side (Flux2Pipeline.swift:618-663, LoRA/LoRALoader.swift:132-140, WeightLoader.swift:736-856):
 * every `<layer>.lora_A.weight` / `.lora_B.weight` pair of the file (each factor in its own float dtype), prefixes
 * `base_model.model.` / `transformer.` / `diffusion_model.` stripped in sequence, scale = (metadata lora_alpha / lora_rank,
 * default 1) x user_scale.  Layers the transformer does not have are skipped and counted; a file matching no layer is
 * FLUX2MI_ERR_INVALID_ARG; a pair whose shapes do not fit its layer is FLUX2MI_ERR_SHAPE. */
int flux2mi_dit_load_lora_safetensors(flux2mi_dit* dit, const char* path, float user_scale, int* n_merged, int* n_unmatched);
/* error if a tensor is missing */
int64_t flux2mi_dit_param_bytes(const flux2mi_dit* dit);
/* Flux2Transformer2DModel.callAsFunction (Transformer/Flux2Transformer.swift:123-327), async.
 *   latents   f32 [s_img, 128] device  (output tokens first, then reference tokens)
 *   txt_emb   bf16 [s_txt, joint_dim] device
 *   timestep  sigma in [0,1] (multiplied by 1000 inside, :145)
 *   guidance  used only when cfg.guidance_embeds (pass the scale, e.g. 4.0)
 *   img_ids   i32 [s_img,4], txt_ids i32 [s_txt,4] device, (T,H,W,L)
 *   out       f32 [s_out,128] device: prediction for the first s_out image tokens
 *             (I2I keeps only the output tokens, Flux2Pipeline.swift:1743). */
int flux2mi_dit_forward(flux2mi_dit* dit, const float* latents, int s_img, const void* txt_emb, int s_txt,
                        float timestep, float guidance, const int32_t* img_ids, const int32_t* txt_ids,
                        float* out, int s_out);
/* ONE block of that forward on a caller-supplied residual stream — Flux2TransformerBlock.callAsFunction
 * (Transformer/Flux2TransformerBlock.swift:80-168) for layer < n_double, Flux2SingleTransformerBlock.callAsFunction
 * (Transformer/Flux2SingleBlock.swift:59-98) after that; async.  hs_in / hs_out f32 [s_txt + s_img, dim] device, text rows first;
 * modulation vectors and RoPE tables from timestep / guidance / ids as in flux2mi_dit_forward.  final_out (may be NULL): f32
 * [s_img,128] = the output layer (:316-324) applied to hs_out.  hs_in == NULL: OBSERVE — the whole forward runs on `latents`
 * (f32 [s_img,128]; otherwise unused, may be NULL), the stream behind block `layer` goes to hs_out and final_out is the forward's
 * prediction.  For teacher-forced per-block parity tests: a host never needs it. */
int flux2mi_dit_forward_block(flux2mi_dit* dit, int layer, const float* hs_in, const float* latents, float* hs_out, int s_img,
                              const void* txt_emb, int s_txt, float timestep, float guidance, const int32_t* img_ids,
                              const int32_t* txt_ids, float* final_out);
/* The timestep embedding and modulation vectors of the last forward outside a generation — TimestepEmbedding + Modulation
 * (Transformer/Flux2Embeddings.swift:124-142, Flux2Modulation.swift:49-75): f32 [256 + 19 dim] = sinusoid | MLP hidden | temb |
 * img (shift, scale, gate) x 2 | txt x 2 | single (shift, scale, gate) | final (shift, scale).  Parity tests only; async. */
int flux2mi_dit_last_modulation(flux2mi_dit* dit, float* out, int64_t n_floats);

/* ---- KV-cached reference conditioning (Transformer/TransformerKVCache.swift,
 * Flux2Transformer.swift:330-546; model variant klein-9b-kv) -------------------------------------- */
typedef struct flux2mi_kv_cache flux2mi_kv_cache;
/* Storage for the post-RoPE keys and the values of s_ref reference tokens in every block. */
int flux2mi_kv_cache_create(flux2mi_dit* dit, int s_txt, int s_ref, flux2mi_kv_cache** out);
int flux2mi_kv_cache_destroy(flux2mi_kv_cache* cache);
/* LayerKVCacheEntry of one block (TransformerKVCache.swift:13-25, :59-67; layer = double blocks, then single blocks):
 * keys bf16 [s_ref, dim] post-RoPE (head h at columns h*128..), values_t bf16 [dim, s_ref] (V transposed); either may be NULL.
 * Device buffers, async on the context's stream; an unfilled cache is FLUX2MI_ERR_NOT_LOADED. */
int flux2mi_kv_cache_entry(flux2mi_kv_cache* cache, int layer, void* keys, void* values_t);
/* forwardKVExtract (:346-446): sequence [txt | ref | out]; reference queries attend to [txt | ref]
 * only (Flux2Attention.swift:296-306, :407-425), everything else to all keys; fills `cache`.
 * latents f32 [s_out,128], ref_latents f32 [s_ref,128], ids i32 [.,4]; out f32 [s_out,128].  Async. */
int flux2mi_dit_forward_kv_extract(flux2mi_dit* dit, flux2mi_kv_cache* cache, const float* latents, int s_out,
                                   const float* ref_latents, int s_ref, const void* txt_emb, int s_txt, float timestep,
                                   float guidance, const int32_t* img_ids, const int32_t* txt_ids, const int32_t* ref_ids,
                                   float* out);
/* forwardKVCached (:459-546): sequence [txt | out], keys/values [txt | cached ref | out], no mask.
 * An unfilled cache is an error status (the reference calls fatalError, :488). */
int flux2mi_dit_forward_kv_cached(flux2mi_dit* dit, flux2mi_kv_cache* cache, const float* latents, int s_out,
                                  const void* txt_emb, int s_txt, float timestep, float guidance, const int32_t* img_ids,
                                  const int32_t* txt_ids, float* out);

/* ---- text encoder: Qwen3 -> Klein prompt embeddings (FluxTextEncoders) ------------------------------
 * Replaces Qwen3Model.forwardWithHiddenStates (Model/Qwen3/Qwen3Model.swift:109-174) as driven by
 * KleinEmbeddingExtractor.extractKleinEmbeddings (Embeddings/KleinEmbeddingExtractor.swift:50-131).
 * Tokenisation and the chat template stay on the host (string work, :143-162). */
typedef struct flux2mi_textenc flux2mi_textenc;
typedef struct flux2mi_textenc_config {
    int vocab, hidden, intermediate;
    int n_layers;            /* of the checkpoint; only layers <= extract_layers[2] are stored and run */
    int heads, kv_heads, head_dim;   /* head_dim must be 128 */
    float rope_theta;        /* 1e6 */
    float rms_eps;           /* 1e-6 */
    int extract_layers[3];   /* 1-based "after layer i" indices, Klein: 9, 18, 27 (KleinConfig.hiddenStateLayers); Dev: 10, 20, 30 */
    int qk_norm;             /* 1: per-head RMSNorm of q and k before RoPE (Qwen3); 0: none (Mistral) */
} flux2mi_textenc_config;
/* presets "klein-4b" (Qwen3-4B) | "klein-9b" (Qwen3-8B) | "dev" (Mistral-Small-3.2-24B text model:
 * Model/MistralModel.swift, MistralAttention.swift, Embeddings/EmbeddingExtractor.swift:202-278) */
int flux2mi_textenc_preset(const char* name, flux2mi_textenc_config* out);
int flux2mi_textenc_create(flux2mi_ctx* ctx, const flux2mi_textenc_config* cfg, flux2mi_textenc** out);
int flux2mi_textenc_destroy(flux2mi_textenc* te);
/* HF keys ("model.layers.N.self_attn.q_proj.weight", ..., with or without the "model." prefix);
 * keys of layers that are not run return FLUX2MI_ERR_UNKNOWN_KEY and can be skipped by the caller. */
int flux2mi_textenc_set_tensor(flux2mi_textenc* te, const char* name, int dtype, const int64_t* shape, int ndim,
                               const void* data);
/* every tensor of a .safetensors shard through flux2mi_textenc_set_tensor (prefixes `language_model.` / `text_encoder.` stripped);
 * keys of layers that are not run, lm_head, vision towers are skipped and counted */
int flux2mi_textenc_load_safetensors(flux2mi_textenc* te, const char* path, int* n_loaded, int* n_skipped);
int flux2mi_textenc_finalize(flux2mi_textenc* te);
int64_t flux2mi_textenc_param_bytes(const flux2mi_textenc* te);
/* ids i32 [S] on the device; real tokens are [pad_left, pad_left + valid_len): Klein right-pads
 * (pad_left 0, KleinEmbeddingExtractor.swift:70-90), Dev left-pads (pad_left = S - valid_len,
 * EmbeddingExtractor.swift:226-247).  Attention mask as the reference builds it: causal, pad keys
 * masked (rows of left-pad tokens end up attending uniformly to the pads before them, the effect of
 * its finite -1e9 mask in f32).  Positions are the row indices in both cases.
 * out bf16 [S, 3*hidden] (device) = hidden states after the three extract layers, concatenated:
 * the DiT's txt_emb.  Async. */
int flux2mi_textenc_forward(flux2mi_textenc* te, const int32_t* ids, int S, int pad_left, int valid_len, void* out);

/* ---- VAE decoder (AutoencoderKLFlux2.decode) ---------------------------- */
typedef struct flux2mi_vae_config {
    int channels[4];       /* decoder block_out_channels, e.g. 128,256,512,512 (small: 96,192,384,384) */
    int latent_channels;   /* 32 */
    int groups;            /* 32 */
    int layers_per_block;  /* 2 (decoder uses +1) */
} flux2mi_vae_config;

/* presets "standard" | "small" (Configuration/VAEConfig.swift:44-81). */
int flux2mi_vae_preset(const char* name, flux2mi_vae_config* out);
int flux2mi_vae_create(flux2mi_ctx* ctx, const flux2mi_vae_config* cfg, flux2mi_vae** out);
int flux2mi_vae_destroy(flux2mi_vae* vae);
/* diffusers keys, conv weights OIHW as in the file (WeightLoader.swift:486-560);
 * encoder keys are accepted and ignored. */
int flux2mi_vae_set_tensor(flux2mi_vae* vae, const char* name, int dtype, const int64_t* shape, int ndim,
                           const void* data);
int flux2mi_vae_finalize(flux2mi_vae* vae);
int flux2mi_vae_load_safetensors(flux2mi_vae* vae, const char* path, int* n_loaded, int* n_skipped);
const float* flux2mi_vae_bn_mean(const flux2mi_vae* vae); /* device f32 [128] */
const float* flux2mi_vae_bn_var(const flux2mi_vae* vae);  /* device f32 [128] */
/* AutoencoderKLFlux2.decode (VAE/AutoencoderKL.swift:129-143) + postprocessVAEOutput
 * (Pipeline/Flux2Pipeline.swift:2425-2468), async.
 *   z        f32 [32,h,w] device (NCHW like the reference)
 *   out_f32  f32 [3,8h,8w] device or NULL
 *   out_u8   u8 [8h,8w,3] device or NULL: clip((x+1)*127.5,0,255), truncating cast */
int flux2mi_vae_decode(flux2mi_vae* vae, const float* z, int h, int w, float* out_f32, uint8_t* out_u8);

/* AutoencoderKLFlux2.encode(x, samplePosterior: false) (VAE/AutoencoderKL.swift:90-123,
 * VAE/VAEEncoder.swift:88-118): image f32 [3,H,W] in [-1,1] (device; H, W multiples of 16) ->
 * posterior mean f32 [32,H/8,W/8].  No scaling factor and no BatchNorm here: chain
 * flux2mi_vae_latents_to_packed() for `encodeImageToPackedSequence`
 * (Pipeline/Flux2Pipeline+ChainHelpers.swift:75-103).  The encoder tensors ("encoder.*",
 * "quant_conv.*") are optional at load time; encode fails with FLUX2MI_ERR_NOT_LOADED without them. */
int flux2mi_vae_encode(flux2mi_vae* vae, const float* image, int H, int W, float* z_mean);
int flux2mi_vae_has_encoder(const flux2mi_vae* vae);

/* ---- pipeline: Flux2Pipeline.generateWithResult hot section -------------- */
/* Flux2StepContext (Pipeline/Flux2Pipeline.swift:42-57) */
typedef struct flux2mi_step_context {
    int step_idx;
    int total_steps;
    float sigma;
    float sigma_next;
    int height;
    int width;
    int is_i2i;
} flux2mi_step_context;

/* Flux2StepHook (:64).  Called on the generating thread after the Euler update of
 * every step incl. the last (sigma_next == 0).  `latents` is the device buffer
 * f32 [s_img,128]; modify it in place with work enqueued on `stream` (e.g.
 * flux2mi_repaint_blend).  Non-zero return cancels (FLUX2MI_ERR_CANCELLED). */
typedef int (*flux2mi_step_hook)(const flux2mi_step_context* ctx, float* latents, void* stream, void* user);
/* Flux2ProgressCallback (:32): (currentStep, totalSteps), after the hook. */
typedef void (*flux2mi_progress_cb)(int step, int total, void* user);
/* Flux2CheckpointCallback (:35): decoded intermediate image, host u8 [H,W,3]. */
typedef void (*flux2mi_checkpoint_cb)(int step, const uint8_t* rgb, int height, int width, void* user);

typedef struct flux2mi_generate_params {
    /* inputs */
    int height, width;             /* rounded UP to x16 like validateDimensions */
    int steps;
    float guidance;                /* embedded guidance (Dev) */
    const void* txt_emb;           /* bf16 [s_txt, joint_dim] device (precomputedEmbeddings) */
    int s_txt;
    const void* neg_emb;           /* classical CFG (base models): bf16 [s_neg, joint_dim] or NULL */
    int s_neg;
    float cfg_scale;               /* > 1 with neg_emb enables two-pass CFG */
    const float* noise;            /* f32 [128,H/16,W/16] device: patchified-layout noise (explicit RNG) */
    const float* init_latents;     /* f32 [S_img,128] device or NULL (img2img init, :1897-1907) */
    float strength;                /* < 1 requires init_latents (:1210-1213) */
    const float* ref_latents;      /* f32 [s_ref,128] device or NULL (I2I conditioning tokens) */
    const int32_t* ref_ids;        /* i32 [s_ref,4] device */
    int s_ref;
    const float* custom_sigmas;    /* host, LoRA scheduler override (:1873-1878) or NULL */
    int n_custom_sigmas;
    int checkpoint_interval;       /* <= 0: none */
    flux2mi_step_hook on_step; void* on_step_user;
    flux2mi_progress_cb on_progress; void* on_progress_user;
    flux2mi_checkpoint_cb on_checkpoint; void* on_checkpoint_user;
    int skip_decode;               /* 1: stop after the denoise loop */
    int use_kv_cache;              /* 1 (needs ref_latents): KV-cached reference conditioning, the klein-9b-kv
                                      path (Flux2Pipeline.swift:1555-1682): step 0 = flux2mi_dit_forward_kv_extract,
                                      steps 1+ = flux2mi_dit_forward_kv_cached */
    /* outputs */
    uint8_t* out_image;            /* host u8 [H,W,3] or NULL */
    uint8_t* out_image_device;     /* device u8 [H,W,3] or NULL: the same bytes left in HBM (e.g. as the send buffer of
                                      the all-gather of a data-parallel batch) */
    float* out_latents;            /* device f32 [S_img,128] or NULL: final packed latents */
    int out_height, out_width;
    int effective_steps;
    float ms_denoise;              /* phase "6. Denoising Loop" (HIP events) */
    float ms_decode;               /* phases "7. VAE Decode" + "8. Post-processing" */
} flux2mi_generate_params;

/* Softmax form of the DiT's joint attention (replaces MLXFast.scaledDotProductAttention, Flux2Attention.swift:168,
 * Flux2ParallelAttention.swift:104 — the same softmax either way, any offset cancels in O / l).
 * mode 0 (default): blocks whose QK-RMSNorm weights bound the scores take the fixed zero offset, the others a per-query
 * offset taken from the data (exact for any scores; counted as "online"); mode 1: the data-offset form everywhere.
 * _info reports how many blocks take which form. */
int flux2mi_dit_set_attention_softmax(flux2mi_dit* dit, int mode);
int flux2mi_dit_attention_softmax_info(const flux2mi_dit* dit, int* n_fixed_offset, int* n_online);

int flux2mi_generate(flux2mi_ctx* ctx, flux2mi_dit* dit, flux2mi_vae* vae, flux2mi_generate_params* p);

/* ---- measurement aid ---------------------------------------------------- */
/* HIP-event timing of one kernel class on the context's stream while the hot path
 * runs (bench.py roofline leg).  kernel_class: 0 = single-block linear1 GEMM,
 * 1 = all DiT GEMMs, 2 = attention, 3 = VAE conv, 4 = linear2, 5 = LayerNorm + modulate, 6 = GroupNorm apply, 7 = proj + mlp_out,
 * 8 = the VAE mid block's attention (work = 4 P^2 C). */
int flux2mi_prof_begin(flux2mi_ctx* ctx, int kernel_class);
int flux2mi_prof_end(flux2mi_ctx* ctx, int* n_launches, double* total_ms, double* flops);
/* The matrix-pipe rate THIS card sustains (diagnostic; bench.py's roofline.mfma_stream): a register-only stream of the MFMA the
 * hot kernels use — fp8 = 0: v_mfma_f32_16x16x32_bf16, 1: the e4m3 v_mfma_scale_f32_16x16x128_f8f6f4 — on pseudo-random
 * operands, one wave per SIMD on every CU, for about `seconds`; *tflops = the rate of the second half.  The hot kernels run at the
 * board's power cap, so this — not the data-sheet peak — is what a perfect kernel could reach on the box at hand.  Synchronous. */
int flux2mi_debug_mfma_stream(flux2mi_ctx* ctx, int fp8, double seconds, double* tflops);

/* Standalone op entry points (tests / microbenchmarks; same kernels the forward uses). */
/* The VAE mid block's single-head attention (AttentionBlock, VAE/ResnetBlock.swift:281-313: softmax(q k^T / sqrt(C)) v) as the decoder
 * runs it: ONE streaming kernel, the P x P scores never in HBM.  q, k, v bf16 [P, C] device (the projections WITH their biases), out bf16
 * [P, C]; C = 512 or 384 (other widths: FLUX2MI_ERR_UNSUPPORTED — the decoder keeps the three-launch form for them). */
int flux2mi_op_vae_attention(flux2mi_ctx* ctx, const void* q, const void* k, const void* v, void* out, int P, int C);
/* C[M,N] (f32) = A[M,K] (bf16) * W[N,K]^T (bf16) */
int flux2mi_op_gemm_bf16(flux2mi_ctx* ctx, const void* A, const void* W, float* C, int M, int N, int K);
/* the same product stored as bf16 [M,N] (one rounding of the f32 accumulator), as the model's block GEMMs store it */
int flux2mi_op_gemm_bf16_out_bf16(flux2mi_ctx* ctx, const void* A, const void* W, void* C, int M, int N, int K);
/* X f32 [M,N] += gate f32 [N] (per column) * (A W^T): the gate + residual epilogue of the block output GEMMs
 * (Flux2TransformerBlock.swift:117-166 applyGate + residual). */
int flux2mi_op_gemm_gate_resid(flux2mi_ctx* ctx, const void* A, const void* W, const float* gate, float* X, int M,
                               int N, int K);
/* same product through the fp8 path: A and W (bf16) are quantised row-wise to OCP e4m3 with
 * per-row scales (what the fp8 model does to activations / weights), then
 * v_mfma_f32_16x16x32_fp8_fp8; K % 128 == 0. */
int flux2mi_op_gemm_fp8(flux2mi_ctx* ctx, const void* A, const void* W, float* C, int M, int N, int K);
/* the same with A quantised in 16-row x 128-column blocks with power-of-two (e8m0) scales — the form the fp8 models' SwiGLU and
   attention epilogues emit for proj / mlp_out / linear2 (no reference counterpart: weight_format 1 is this library's own quantiser) */
int flux2mi_op_gemm_fp8_blocks(flux2mi_ctx* ctx, const void* A, const void* W, float* C, int M, int N, int K);
/* Per-op entry points of the HBM-bound kernels (their own tolerance, SURVEY §8c: <= 4e-3 vs the fp32 oracle).
 * y bf16 [rows,dim] = LayerNorm(x f32, eps 1e-6, no affine) * (1 + scale) + shift; rows < split: set 0, else set 1
 * (Flux2TransformerBlock.swift:56-61, Flux2Modulation.swift:96-112). */
int flux2mi_op_layernorm_modulate(flux2mi_ctx* ctx, const float* x, void* y, int rows, int dim, int split, const float* shift0,
                                  const float* scale0, const float* shift1, const float* scale1);
/* in place on qkv bf16 [S, 3*heads*128]: per-head RMSNorm (eps 1e-6, bf16 weights [128], set 0 for rows < split) then
 * 4-axis RoPE from ids i32 [S,4] on q and k, q times q_mul — the standalone kernel (Flux2Attention.swift:137-157,
 * Flux2RoPE.swift:129-169) */
int flux2mi_op_qk_norm_rope(flux2mi_ctx* ctx, void* qkv, int S, int heads, int split, const void* qw0, const void* kw0,
                            const void* qw1, const void* kw1, const int32_t* ids, float q_mul);
/* out bf16 [S, 2*dim] = RoPE(RMSNorm(x W^T)), W bf16 [2*dim, dim] = the q | k rows: the same math fused into the GEMM epilogue */
int flux2mi_op_qk_proj_norm_rope(flux2mi_ctx* ctx, const void* x, const void* w, int S, int dim, const void* qw, const void* kw,
                                 const int32_t* ids, float q_mul, void* out);
/* y = [SiLU](GroupNorm(x)), NHWC bf16 [pixels, C], f32 statistics (VAE/ResnetBlock.swift:9-55, :168-186) */
int flux2mi_op_groupnorm_silu(flux2mi_ctx* ctx, const void* x, void* y, long long pixels, int C, int groups, const float* gamma,
                              const float* beta, float eps, int silu);
/* joint attention: qkv bf16 [S, 3*heads*128] (q|k|v, already normalised/rotated) -> out bf16 [S, heads*128] */
int flux2mi_op_attention(flux2mi_ctx* ctx, const void* qkv, void* out, int S, int heads);
/* the form the blocks run: q pre-multiplied by log2(e)/sqrt(128), |q.k| <= score_bound_log2 in that domain -> softmax against a
   fixed zero offset when 0 < bound <= 80 (Flux2Attention.swift:168-174 after the QK RMSNorm bounds the scores); bound <= 0
   (unknown) or > 80: a per-query offset taken from the data, exact for any scores (attention64b.hip, DOFF) */
int flux2mi_op_attention_prescaled(flux2mi_ctx* ctx, const void* qkv, void* out, int S, int heads, float score_bound_log2);
/* The launch plan of that attention on this GPU (Flux2Attention.swift:168-174 has no counterpart: MLX's SDPA is one call): *pieces = into
 * how many equal key ranges the (head, 256-query block) items of the partial last round are cut (1 = whole items only; fixed-offset
 * softmax, partial sums added in a fixed order), *n_cu = compute units.  Either pointer may be NULL. */
int flux2mi_op_attention_split_info(flux2mi_ctx* ctx, int S, int heads, int* pieces, int* n_cu);
/* the fp8 models' form (weight_format 1, plain forwards; q_exp, k_exp in [-60, 60], negative = operands scaled down into e4m3's range):
   Q K^T on e4m3 copies q8 = q 2^q_exp, k8 = k 2^k_exp (device bytes
   [S, heads*128] each, q pre-scaled as above; the powers of two are undone by the scale operands of the block-scaled MFMA), V bf16
   [S, heads*128]; softmax, P and P V as in the bf16 form.  Not a restatement of the reference (its quantised variants are weight-only). */
int flux2mi_op_attention_qk8(flux2mi_ctx* ctx, const void* q8, const void* k8, const void* v, void* out, int S, int heads, int q_exp, int k_exp,
                             float score_bound_log2);
/* the same with the output as block-scaled e4m3, the form proj / linear2 of an fp8 model consume: out8 bytes [S, heads*128] and one e8m0
   scale byte per 16 queries x one head's 128 columns at bs[head * bs_ld + row / 16] (bs_ld >= ceil(S / 16)); out8 NULL: bf16 `out` */
int flux2mi_op_attention_qk8_blocks(flux2mi_ctx* ctx, const void* q8, const void* k8, const void* v, void* out, void* out8, void* bs, int bs_ld,
                                    int S, int heads, int q_exp, int k_exp, float score_bound_log2);
/* the same with queries [q0, q0+nq) against the key PREFIX [0, nk) of a V^T image holding all S keys — the launch shape of the
   KV-extraction pass, whose reference queries see [text | reference] only (Flux2Attention.swift:407-425: the additive mask is a
   key-prefix limit; Flux2Transformer.swift:346-457).  Only out rows [q0, q0+nq) are written */
int flux2mi_op_attention_prescaled_range(flux2mi_ctx* ctx, const void* qkv, void* out, int S, int heads, float score_bound_log2,
                                         int q0, int nq, int nk);

#ifdef __cplusplus
}
#endif
#endif /* FLUX2MI_H */
