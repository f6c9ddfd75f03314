// dit.cpp — Flux2Transformer2DModel on MI355X: weight ingestion (BFL + diffusers key
// schemes), HBM-resident bf16 weights with the fused layouts the kernels want, and the
// forward orchestration (Transformer/Flux2Transformer.swift:123-327).
//
// Data layout in HBM (B = 1, S = S_txt + S_img, joint order [text ; image (; refs)]):
//   hs    f32  [S, d]      residual stream (f32 like the reference's activations)
//   xn    bf16 [S, d]      LayerNorm+modulate output = GEMM A operand
//   qkv   bf16 [S, 3d]     q | k | v, RMSNorm+RoPE applied in place on q,k
//   vt    bf16 [H,128,Sp]  V transposed + key-permuted for the attention PV operand
//   comb  bf16 [S, 4d]     single blocks: attention output (cols 0..d) | SwiGLU hidden (d..4d)
//                          double blocks: attention output [S,d] (ld d) and FF hidden [S,3d]
//   mods  f32              temb, per-stream (shift,scale,gate) vectors, final (shift,scale)
// weight_format 1 (fp8): the block GEMM weights (qkv, proj, mlp, linear1/2) are e4m3 bytes with a
// f32 scale per output channel; their A operands are e4m3 too, quantised per token:
//   xq    e4m3 [S, 4d] + xs f32 [S]   LN+modulate output / quantised comb rows
// Embedders, modulation GEMVs and proj_out stay bf16.
#include <errno.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <map>

#include <string>
#include <unordered_map>
#include <vector>

#include "common.h"
#include "loader.h"
#include "kernels.h"

namespace {

enum Transform { T_DIRECT = 0, T_SWIGLU = 1, T_LINEAR1 = 2, T_SWAP_HALVES = 3 };

struct Slot {
    int buf;         // index into bufs
    int row_off;     // destination row offset (T_DIRECT)
    int rows, cols;  // expected source shape
    int transform;
};

struct Buf {
    uint16_t* ptr = nullptr;  // bf16 [rows, cols]; nullptr for an fp8 buffer
    uint8_t* q = nullptr;     // weight_format 1: e4m3 [rows, cols]
    float* scale = nullptr;   //                  per output channel (row) scale
    bool fp8 = false;
    int rows = 0, cols = 0;
    // rows written so far, as [begin, end) destination ranges keyed by begin: loading a slot twice (BFL and
    // diffusers aliases in one file, a reload, duplicate shards) covers its rows once, and `to_q` three times
    // never stands in for `to_k` / `to_v`
    std::map<int, int> loaded;
    std::string label;
    void mark(int begin, int end) { int& e = loaded[begin]; e = std::max(e, end); }
    int got_rows() const {
        int n = 0, hi = 0;
        for (const auto& r : loaded) {   // ascending begin
            const int b = std::max(r.first, hi);
            if (r.second > b) { n += r.second - b; hi = r.second; }
        }
        return n;
    }
};

}  // namespace

struct flux2mi_dit {
    flux2mi_ctx* ctx = nullptr;
    flux2mi_dit_config cfg{};
    std::vector<Buf> bufs;
    std::unordered_map<std::string, Slot> slots;
    bool finalized = false;
    int attn_softmax_mode = 0;  // 0: fixed-offset softmax where score_bound allows it; 1: no bound assumed anywhere (data-offset form)
    bool lora_merged = false;  // a LoRA was merged into the resident weights (tagged in pre-quantized exports)
    int64_t param_bytes = 0;

    // resolved weight pointers
    uint16_t *img_in = nullptr, *txt_in = nullptr, *time_in = nullptr, *time_out = nullptr;
    uint16_t *guid_in = nullptr, *guid_out = nullptr;
    uint16_t *mod_img = nullptr, *mod_txt = nullptr, *mod_single = nullptr, *final_mod = nullptr, *final_lin = nullptr;
    struct W { const uint16_t* w = nullptr; const float* scale = nullptr; };  // bf16 pointer, or e4m3 bytes + scales
    // score_bound: upper bound of |q.k| / sqrt(128) * log2(e) after the block's QK-RMSNorm (see finalize)
    // q8_exp / k8_exp (fp8 models): e4m3 Q / K of the attention are q 2^q8_exp / k 2^k8_exp, the largest powers of two that keep the
    // RMSNorm-bounded elements (|x_d| <= sqrt(128) max|w|) inside e4m3's 448 (finalize)
    struct Dbl { W qkv[2], proj[2], mlp_in[2], mlp_out[2]; uint16_t *qn[2], *kn[2]; float score_bound = 0.f; int q8_exp = 0, k8_exp = 0; };  // [0]=txt, [1]=img
    struct Sgl { W lin1, lin2; uint16_t *qn, *kn; float score_bound = 0.f; int q8_exp = 0, k8_exp = 0; };
    std::vector<Dbl> dbl;
    std::vector<Sgl> sgl;

    // staging for ingestion
    DevBuf stage_raw, stage_bf16, stage_w, stage_d, stage_q8;
    // forward workspace
    DevBuf hs, xn, qkv, vt, comb, mods, rope, lat_bf16, xq, xs, qk8, xbs, xq2;
    // flux2mi_dit_step_cache (common.h): what the last plain forward of the running generation projected / tabulated
    struct StepCache {
        bool on = false, txt_ok = false, rope_ok = false;
        const void* txt = nullptr;
        const int32_t *txt_ids = nullptr, *img_ids = nullptr;
        int s_txt = 0, s_lat = 0;
        DevBuf hs_txt;   // f32 [s_txt, d]: the text rows of the residual stream right behind txt_in
        // flux2mi_dit_prepare_steps: rows of the forward's `mods` layout (256 + 19 d floats), one per timestep
        std::vector<float> ts;
        float guidance = 0.f;
        DevBuf mods_all;
    } step_cache;
    DevBuf splitk_ws;   // proj_out ([s_out, d] x [128, d]^T: 32 tiles of 128x128 at 1024^2, 48 K-tiles): split-K workspace (kernels.h)
    int ws_S = 0;
    // flux2mi_dit_forward_block (parity tests): run ONE block of a plain forward on a caller-supplied residual stream
    struct BlockTap { int layer = -1; const float* hs_in = nullptr; float* hs_out = nullptr; } tap;
};

namespace {

int add_buf(flux2mi_dit* m, int rows, int cols, const std::string& label, bool gemm_weight = false) {
    Buf b;
    b.fp8 = gemm_weight && m->cfg.weight_format == 1;
    b.rows = rows;
    b.cols = cols;
    b.label = label;
    m->bufs.push_back(b);
    return (int)m->bufs.size() - 1;
}
void add_slot(flux2mi_dit* m, const std::string& name, int buf, int row_off, int rows, int cols, int transform) {
    m->slots[name] = Slot{buf, row_off, rows, cols, transform};
}

// Build the name -> slot registry for both key schemes.
void build_registry(flux2mi_dit* m) {
    const int d = m->cfg.dim, J = m->cfg.joint_dim;
    auto simple = [&](int rows, int cols, const std::string& label, std::initializer_list<const char*> names,
                      int transform = T_DIRECT, bool gemm_weight = false) {
        const int b = add_buf(m, rows, cols, label, gemm_weight);
        for (const char* n : names) add_slot(m, n, b, 0, rows, cols, transform);
        return b;
    };
    // BFL name first, diffusers alias second (WeightLoader.swift:99-205 / :397-480)
    simple(d, 128, "img_in", {"img_in.weight", "x_embedder.weight"});
    simple(d, J, "txt_in", {"txt_in.weight", "context_embedder.weight"});
    simple(d, 256, "time_in.in", {"time_in.in_layer.weight", "time_guidance_embed.timestep_embedder.linear_1.weight"});
    simple(d, d, "time_in.out", {"time_in.out_layer.weight", "time_guidance_embed.timestep_embedder.linear_2.weight"});
    if (m->cfg.guidance_embeds) {
        simple(d, 256, "guidance_in.in",
               {"guidance_in.in_layer.weight", "time_guidance_embed.guidance_embedder.linear_1.weight"});
        simple(d, d, "guidance_in.out",
               {"guidance_in.out_layer.weight", "time_guidance_embed.guidance_embedder.linear_2.weight"});
    }
    simple(6 * d, d, "mod_img", {"double_stream_modulation_img.lin.weight", "double_stream_modulation_img.linear.weight"});
    simple(6 * d, d, "mod_txt", {"double_stream_modulation_txt.lin.weight", "double_stream_modulation_txt.linear.weight"});
    simple(3 * d, d, "mod_single", {"single_stream_modulation.lin.weight", "single_stream_modulation.linear.weight"});
    {
        // internal order = BFL order [shift | scale]; the diffusers file stores [scale | shift]
        // (WeightLoader.swift:182-202 swaps BFL -> diffusers; we swap the other way).
        const int b = add_buf(m, 2 * d, d, "final_mod");
        add_slot(m, "final_layer.adaLN_modulation.1.weight", b, 0, 2 * d, d, T_DIRECT);
        add_slot(m, "norm_out.linear.weight", b, 0, 2 * d, d, T_SWAP_HALVES);
    }
    simple(128, d, "final_lin", {"final_layer.linear.weight", "proj_out.weight"});

    char a[160], c[160];
    for (int i = 0; i < m->cfg.n_double; ++i) {
        for (int s = 0; s < 2; ++s) {  // 0 = txt, 1 = img
            const char* bs = s ? "img" : "txt";
            // fused qkv [3d, d]
            snprintf(a, sizeof a, "dbl%d.%s.qkv", i, bs);
            const int bq = add_buf(m, 3 * d, d, a, true);
            snprintf(a, sizeof a, "double_blocks.%d.%s_attn.qkv.weight", i, bs);
            add_slot(m, a, bq, 0, 3 * d, d, T_DIRECT);
            const char* dq[3] = {s ? "to_q" : "add_q_proj", s ? "to_k" : "add_k_proj", s ? "to_v" : "add_v_proj"};
            for (int j = 0; j < 3; ++j) {
                snprintf(a, sizeof a, "transformer_blocks.%d.attn.%s.weight", i, dq[j]);
                add_slot(m, a, bq, j * d, d, d, T_DIRECT);
            }
            snprintf(a, sizeof a, "double_blocks.%d.%s_attn.proj.weight", i, bs);
            snprintf(c, sizeof c, "transformer_blocks.%d.attn.%s.weight", i, s ? "to_out.0" : "to_add_out");
            simple(d, d, a, {a, c}, T_DIRECT, true);
            snprintf(a, sizeof a, "double_blocks.%d.%s_attn.norm.query_norm.scale", i, bs);
            snprintf(c, sizeof c, "transformer_blocks.%d.attn.%s.weight", i, s ? "norm_q" : "norm_added_q");
            simple(1, 128, a, {a, c});
            snprintf(a, sizeof a, "double_blocks.%d.%s_attn.norm.key_norm.scale", i, bs);
            snprintf(c, sizeof c, "transformer_blocks.%d.attn.%s.weight", i, s ? "norm_k" : "norm_added_k");
            simple(1, 128, a, {a, c});
            snprintf(a, sizeof a, "double_blocks.%d.%s_mlp.0.weight", i, bs);
            snprintf(c, sizeof c, "transformer_blocks.%d.%s.linear_in.weight", i, s ? "ff" : "ff_context");
            simple(6 * d, d, a, {a, c}, T_SWIGLU, true);
            snprintf(a, sizeof a, "double_blocks.%d.%s_mlp.2.weight", i, bs);
            snprintf(c, sizeof c, "transformer_blocks.%d.%s.linear_out.weight", i, s ? "ff" : "ff_context");
            simple(d, 3 * d, a, {a, c}, T_DIRECT, true);
        }
    }
    for (int i = 0; i < m->cfg.n_single; ++i) {
        snprintf(a, sizeof a, "single_blocks.%d.linear1.weight", i);
        snprintf(c, sizeof c, "single_transformer_blocks.%d.attn.to_qkv_mlp_proj.weight", i);
        simple(9 * d, d, a, {a, c}, T_LINEAR1, true);
        snprintf(a, sizeof a, "single_blocks.%d.linear2.weight", i);
        snprintf(c, sizeof c, "single_transformer_blocks.%d.attn.to_out.weight", i);
        simple(d, 4 * d, a, {a, c}, T_DIRECT, true);
        snprintf(a, sizeof a, "single_blocks.%d.norm.query_norm.scale", i);
        snprintf(c, sizeof c, "single_transformer_blocks.%d.attn.norm_q.weight", i);
        simple(1, 128, a, {a, c});
        snprintf(a, sizeof a, "single_blocks.%d.norm.key_norm.scale", i);
        snprintf(c, sizeof c, "single_transformer_blocks.%d.attn.norm_k.weight", i);
        simple(1, 128, a, {a, c});
    }
}

uint16_t* buf_by_label(flux2mi_dit* m, const std::string& label) {
    for (Buf& b : m->bufs)
        if (b.label == label) return b.ptr;
    return nullptr;
}
flux2mi_dit::W weight_by_label(flux2mi_dit* m, const std::string& label) {
    flux2mi_dit::W w;
    for (Buf& b : m->bufs)
        if (b.label == label) {
            w.w = b.fp8 ? reinterpret_cast<const uint16_t*>(b.q) : b.ptr;
            w.scale = b.scale;
        }
    return w;
}

// Put `rows` x `cols` bf16 rows (checkpoint order) of slot `s` where the kernels want them.  `dst` is
// the base of a bf16 image of the slot's buffer (the weight buffer itself, or a scratch image of it for
// the LoRA merge); nullptr = the buffer is fp8: the transform runs into a bf16 staging copy and the
// rows are quantised (scale = row amax / 448).
int place_rows(flux2mi_dit* m, const Slot& s, const uint16_t* src, uint16_t* dst) {
    hipStream_t st = m->ctx->stream;
    Buf& b = m->bufs[s.buf];
    const int64_t rows = s.rows, cols = s.cols;
    const size_t row_bytes = (size_t)cols * 2;
    if (dst == nullptr) {
        const uint16_t* rows_src = src;
        if (s.transform == T_SWIGLU || s.transform == T_LINEAR1) {
            F2_TRY(m->stage_w.reserve((size_t)rows * cols * 2));
            uint16_t* w = m->stage_w.as<uint16_t>();
            if (s.transform == T_SWIGLU) {
                F2_LAUNCH(launch_swiglu_interleave_rows(st, src, w, (int)(rows / 2), (int)cols), "swiglu_interleave");
            } else {
                const int d = m->cfg.dim;
                F2_CHECK_HIP(hipMemcpyAsync(w, src, (size_t)3 * d * row_bytes, hipMemcpyDeviceToDevice, st));
                F2_LAUNCH(launch_swiglu_interleave_rows(st, src + (size_t)3 * d * cols, w + (size_t)3 * d * cols, 3 * d,
                                                        (int)cols),
                          "swiglu_interleave(linear1)");
            }
            rows_src = w;
        } else {
            F2_REQUIRE(s.transform == T_DIRECT, FLUX2MI_ERR_UNSUPPORTED, "transform on an fp8 buffer");
        }
        F2_LAUNCH(launch_quantize_rows_fp8(st, rows_src, (int)cols, b.q + (size_t)s.row_off * cols, (int)cols,
                                           b.scale + s.row_off, (int)rows, (int)cols),
                  "quantize weight rows");
        return FLUX2MI_OK;
    }
    switch (s.transform) {
        case T_DIRECT:
            F2_CHECK_HIP(hipMemcpyAsync(dst + (size_t)s.row_off * cols, src, (size_t)rows * row_bytes,
                                        hipMemcpyDeviceToDevice, st));
            break;
        case T_SWAP_HALVES: {
            const size_t half = (size_t)(rows / 2) * cols;
            F2_CHECK_HIP(hipMemcpyAsync(dst, src + half, half * 2, hipMemcpyDeviceToDevice, st));
            F2_CHECK_HIP(hipMemcpyAsync(dst + half, src, half * 2, hipMemcpyDeviceToDevice, st));
            break;
        }
        case T_SWIGLU:
            F2_LAUNCH(launch_swiglu_interleave_rows(st, src, dst, (int)(rows / 2), (int)cols), "swiglu_interleave");
            break;
        case T_LINEAR1: {
            const int d = m->cfg.dim;
            F2_CHECK_HIP(hipMemcpyAsync(dst, src, (size_t)3 * d * row_bytes, hipMemcpyDeviceToDevice, st));
            F2_LAUNCH(launch_swiglu_interleave_rows(st, src + (size_t)3 * d * cols, dst + (size_t)3 * d * cols, 3 * d,
                                                    (int)cols),
                      "swiglu_interleave(linear1)");
            break;
        }
    }
    return FLUX2MI_OK;
}

size_t dtype_size(int dtype) { return dtype == FLUX2MI_F32 ? 4 : 2; }

}  // namespace

extern "C" int flux2mi_dit_preset(const char* name, flux2mi_dit_config* out) {
    F2_REQUIRE(name && out, FLUX2MI_ERR_INVALID_ARG, "dit_preset: NULL argument");
    // Configuration/Flux2Config.swift:258-329
    if (!strcmp(name, "klein-4b")) *out = flux2mi_dit_config{3072, 24, 5, 20, 7680, 0};
    else if (!strcmp(name, "klein-9b")) *out = flux2mi_dit_config{4096, 32, 8, 24, 12288, 0};
    else if (!strcmp(name, "dev")) *out = flux2mi_dit_config{6144, 48, 8, 48, 15360, 1};
    else {
        flux2mi_set_error("dit_preset: unknown model '%s'", name);
        return FLUX2MI_ERR_INVALID_ARG;
    }
    return FLUX2MI_OK;
}

extern "C" int flux2mi_dit_create(flux2mi_ctx* ctx, const flux2mi_dit_config* cfg, flux2mi_dit** out) {
    F2_REQUIRE(ctx && cfg && out, FLUX2MI_ERR_INVALID_ARG, "dit_create: NULL argument");
    F2_REQUIRE(cfg->dim == cfg->heads * 128 && cfg->dim > 0 && cfg->dim <= 8192 && cfg->n_double >= 0 &&
                   cfg->n_single >= 0 && cfg->joint_dim > 0 && cfg->joint_dim % 64 == 0,
               FLUX2MI_ERR_INVALID_ARG, "dit_create: invalid config (dim %d heads %d joint %d)", cfg->dim, cfg->heads,
               cfg->joint_dim);
    F2_REQUIRE(cfg->weight_format == 0 || cfg->weight_format == 1, FLUX2MI_ERR_UNSUPPORTED,
               "dit_create: weight_format %d (0 = bf16, 1 = fp8 e4m3)", cfg->weight_format);
    F2_REQUIRE(cfg->weight_format == 0 || cfg->dim % 256 == 0, FLUX2MI_ERR_UNSUPPORTED,
               "dit_create: fp8 needs dim %% 256 == 0 (dim %d)", cfg->dim);
    F2_CHECK_HIP(hipSetDevice(ctx->device));
    flux2mi_dit* m = new flux2mi_dit();
    m->ctx = ctx;
    m->cfg = *cfg;
    build_registry(m);
    for (Buf& b : m->bufs) {
        const size_t bytes = b.fp8 ? (size_t)b.rows * b.cols : (size_t)b.rows * b.cols * 2;
        hipError_t e = b.fp8 ? f2_malloc((void**)&b.q, bytes) : f2_malloc((void**)&b.ptr, bytes);
        if (e == hipSuccess && b.fp8) {
            e = f2_malloc((void**)&b.scale, (size_t)b.rows * 4);
            m->param_bytes += (int64_t)b.rows * 4;
        }
        if (e != hipSuccess) {
            flux2mi_set_error("dit_create: hipMalloc(%zu) for %s failed: %s", bytes, b.label.c_str(),
                              hipGetErrorString(e));
            flux2mi_dit_destroy(m);
            return FLUX2MI_ERR_OOM;
        }
        m->param_bytes += (int64_t)bytes;
    }
    *out = m;
    return FLUX2MI_OK;
}

extern "C" int flux2mi_dit_destroy(flux2mi_dit* m) {
    if (!m) return FLUX2MI_OK;
    F2_WORKSPACE_USE(m->ctx->device);   // explicit releases below: not beside a trim
    (void)hipSetDevice(m->ctx->device);
    (void)hipStreamSynchronize(m->ctx->stream);
    for (Buf& b : m->bufs) {
        if (b.ptr) f2_free(b.ptr);
        if (b.q) f2_free(b.q);
        if (b.scale) f2_free(b.scale);
    }
    DevBuf* ws[] = {&m->stage_raw, &m->stage_bf16, &m->stage_w, &m->stage_d, &m->stage_q8, &m->hs, &m->xn, &m->qkv, &m->vt, &m->comb, &m->mods,
                    &m->rope, &m->lat_bf16, &m->xq, &m->xs, &m->splitk_ws, &m->qk8, &m->xbs, &m->xq2, &m->step_cache.hs_txt, &m->step_cache.mods_all};
    for (DevBuf* w : ws) w->release();
    delete m;
    return FLUX2MI_OK;
}

extern "C" int64_t flux2mi_dit_param_bytes(const flux2mi_dit* m) { return m ? m->param_bytes : 0; }

extern "C" int flux2mi_dit_set_tensor(flux2mi_dit* m, const char* name, int dtype, const int64_t* shape, int ndim,
                                      const void* data) {
    F2_REQUIRE(m && name && shape && data, FLUX2MI_ERR_INVALID_ARG, "dit_set_tensor: NULL argument");
    F2_WORKSPACE_USE(m->ctx->device);
    F2_REQUIRE(dtype == FLUX2MI_F32 || dtype == FLUX2MI_BF16 || dtype == FLUX2MI_F16, FLUX2MI_ERR_INVALID_ARG,
               "dit_set_tensor(%s): unsupported dtype %d", name, dtype);
    auto it = m->slots.find(name);
    if (it == m->slots.end()) {
        flux2mi_set_error("dit_set_tensor: unknown key '%s'", name);
        return FLUX2MI_ERR_UNKNOWN_KEY;
    }
    const Slot& s = it->second;
    int64_t rows = 1, cols = 1;
    if (ndim == 1) { rows = 1; cols = shape[0]; }
    else if (ndim == 2) { rows = shape[0]; cols = shape[1]; }
    else { flux2mi_set_error("dit_set_tensor(%s): ndim %d", name, ndim); return FLUX2MI_ERR_SHAPE; }
    F2_REQUIRE(rows == s.rows && cols == s.cols, FLUX2MI_ERR_SHAPE, "dit_set_tensor(%s): shape [%lld,%lld], expected [%d,%d]",
               name, (long long)rows, (long long)cols, s.rows, s.cols);
    F2_CHECK_HIP(hipSetDevice(m->ctx->device));
    hipStream_t st = m->ctx->stream;
    const int64_t n = rows * cols;
    Buf& b = m->bufs[s.buf];
    // 1. bring the bytes onto the device (src may be host or device memory)
    F2_TRY(m->stage_raw.reserve((size_t)n * dtype_size(dtype)));
    F2_CHECK_HIP(hipMemcpyAsync(m->stage_raw.ptr, data, (size_t)n * dtype_size(dtype), hipMemcpyDefault, st));
    // 2. convert to bf16
    const uint16_t* src = nullptr;
    if (dtype == FLUX2MI_BF16) {
        src = m->stage_raw.as<uint16_t>();
    } else {
        F2_TRY(m->stage_bf16.reserve((size_t)n * 2));
        F2_LAUNCH(launch_convert_to_bf16(st, m->stage_raw.ptr, dtype, m->stage_bf16.as<uint16_t>(), n), "convert_to_bf16");
        src = m->stage_bf16.as<uint16_t>();
    }
    // 3. place with the layout transform
    F2_TRY(place_rows(m, s, src, b.fp8 ? nullptr : b.ptr));
    F2_CHECK_HIP(hipStreamSynchronize(st));  // the caller may free `data` on return
    // T_DIRECT slots land at their row offset; the transforming slots (SwiGLU interleave, linear1, half swap) cover
    // the whole buffer
    if (s.transform == T_DIRECT) b.mark(s.row_off, s.row_off + (int)rows);
    else b.mark(0, b.rows);
    m->finalized = false;
    return FLUX2MI_OK;
}

namespace {
GemmProblem mk(const uint16_t* A, int lda, const uint16_t* W, int ldw, void* C, int ldc, int M, int N, int K, int epi);
}

// quanto qint8 tensor pair (`<key>._data` int8 [out,in], `<key>._scale` f32 [out] or [out,1]):
// de-quantised on the device (data * scale, WeightLoader.swift:347-363) and ingested like any tensor.
// The reference keeps the product in f16; here it is rounded to the bf16 weight storage.
extern "C" int flux2mi_dit_set_tensor_q8(flux2mi_dit* m, const char* name, const int8_t* data, const float* scale,
                                         int64_t rows, int64_t cols) {
    F2_REQUIRE(m && name && data && scale && rows > 0 && cols > 0, FLUX2MI_ERR_INVALID_ARG, "dit_set_tensor_q8: bad argument");
    F2_WORKSPACE_USE(m->ctx->device);
    F2_CHECK_HIP(hipSetDevice(m->ctx->device));
    hipStream_t st = m->ctx->stream;
    const size_t n = (size_t)rows * cols;
    F2_TRY(m->stage_q8.reserve(n + (size_t)rows * 4 + n * 2 + 512));
    int8_t* dq = m->stage_q8.as<int8_t>();
    float* ds = reinterpret_cast<float*>(dq + ((n + 255) & ~(size_t)255));
    uint16_t* out = reinterpret_cast<uint16_t*>(reinterpret_cast<uint8_t*>(ds) + (((size_t)rows * 4 + 255) & ~(size_t)255));
    F2_CHECK_HIP(hipMemcpyAsync(dq, data, n, hipMemcpyDefault, st));
    F2_CHECK_HIP(hipMemcpyAsync(ds, scale, (size_t)rows * 4, hipMemcpyDefault, st));
    F2_LAUNCH(launch_dequant_i8_rows(st, dq, ds, out, (int)rows, (int)cols), "dequant_i8");
    const int64_t shape[2] = {rows, cols};
    return flux2mi_dit_set_tensor(m, name, FLUX2MI_BF16, shape, 2, out);
}

// LoRA merge into the resident weight: W += scale * (B A), A [rank, in], B [out, rank]
// (WeightLoader.mergeLoRAWeights, Loading/WeightLoader.swift:736-856, non-quantised branch: the
// delta is formed in the weight dtype and added in it).  `name` is any checkpoint key of the target
// weight (BFL or diffusers); the delta goes through the same layout transform as the weight did.
// The shapes come from the file, never from the target: A [a_rows = rank, a_cols = in], B [b_rows = out, b_cols = rank] must fit the
// weight `name` addresses ([out, in]) or the call fails with FLUX2MI_ERR_SHAPE naming the layer (the reference fails with a shape
// error in `matmul` for a LoRA trained at another width; reading rank x cols elements from a smaller tensor would run past it).
// A and B may have different float dtypes (each is converted to bf16 on its own).
extern "C" int flux2mi_dit_merge_lora_checked(flux2mi_dit* m, const char* name, int dtype_a, const void* A, int64_t a_rows, int64_t a_cols,
                                              int dtype_b, const void* B, int64_t b_rows, int64_t b_cols, float scale) {
    F2_REQUIRE(m && name && A && B && a_rows > 0 && a_rows <= INT32_MAX, FLUX2MI_ERR_INVALID_ARG, "dit_merge_lora: bad argument");
    F2_WORKSPACE_USE(m->ctx->device);
    for (int dtype : {dtype_a, dtype_b})
        F2_REQUIRE(dtype == FLUX2MI_F32 || dtype == FLUX2MI_BF16 || dtype == FLUX2MI_F16, FLUX2MI_ERR_INVALID_ARG,
                   "dit_merge_lora(%s): unsupported dtype %d", name, dtype);
    auto it = m->slots.find(name);
    if (it == m->slots.end()) {
        flux2mi_set_error("dit_merge_lora: unknown key '%s'", name);
        return FLUX2MI_ERR_UNKNOWN_KEY;
    }
    const Slot& s = it->second;
    Buf& b = m->bufs[s.buf];
    F2_REQUIRE(a_rows == b_cols, FLUX2MI_ERR_SHAPE, "dit_merge_lora(%s): lora_A [%lld, %lld] and lora_B [%lld, %lld] disagree on the rank", name,
               (long long)a_rows, (long long)a_cols, (long long)b_rows, (long long)b_cols);
    F2_REQUIRE(a_cols == s.cols && b_rows == s.rows, FLUX2MI_ERR_SHAPE,
               "dit_merge_lora(%s): LoRA is for a [%lld, %lld] weight, this model's is [%d, %d] (a LoRA of another model width?)", name,
               (long long)b_rows, (long long)a_cols, s.rows, s.cols);
    const int rank = (int)a_rows;
    F2_REQUIRE(!b.fp8, FLUX2MI_ERR_UNSUPPORTED,
               "dit_merge_lora(%s): fp8 weights (merge into the bf16 checkpoint before loading with weight_format 1)", name);
    F2_REQUIRE(b.got_rows() >= b.rows, FLUX2MI_ERR_NOT_LOADED, "dit_merge_lora(%s): base weight not loaded", name);
    F2_CHECK_HIP(hipSetDevice(m->ctx->device));
    hipStream_t st = m->ctx->stream;
    const int64_t rows = s.rows, cols = s.cols;
    // A, B -> bf16 on the device; A^T so that delta = B (A^T)^T is the library's C = X W^T form
    const size_t na = (size_t)rank * cols, nb = (size_t)rows * rank;
    const size_t raw_a = na * dtype_size(dtype_a), raw_b = nb * dtype_size(dtype_b);
    F2_TRY(m->stage_raw.reserve(raw_a > raw_b ? raw_a : raw_b));
    F2_TRY(m->stage_bf16.reserve((2 * na + nb) * 2 + 1024));
    uint16_t* a16 = m->stage_bf16.as<uint16_t>();
    uint16_t* at16 = a16 + ((na + 127) & ~(size_t)127);
    uint16_t* b16 = at16 + ((na + 127) & ~(size_t)127);
    F2_CHECK_HIP(hipMemcpyAsync(m->stage_raw.ptr, A, raw_a, hipMemcpyDefault, st));
    F2_LAUNCH(launch_convert_to_bf16(st, m->stage_raw.ptr, dtype_a, a16, (int64_t)na), "lora A->bf16");
    F2_LAUNCH(launch_transpose_bf16(st, a16, at16, rank, (int)cols), "lora A^T");
    F2_CHECK_HIP(hipMemcpyAsync(m->stage_raw.ptr, B, raw_b, hipMemcpyDefault, st));
    F2_LAUNCH(launch_convert_to_bf16(st, m->stage_raw.ptr, dtype_b, b16, (int64_t)nb), "lora B->bf16");
    // delta (bf16, checkpoint row order) = scale * B A
    F2_TRY(m->stage_w.reserve((size_t)rows * cols * 2));
    GemmProblem p = mk(b16, rank, at16, rank, m->stage_w.ptr, (int)cols, (int)rows, (int)cols, rank, EPI_BF16);
    p.alpha = scale;
    F2_LAUNCH(launch_gemm(st, &p, 1), "lora delta");
    // the delta in the layout of the weight buffer, then W += delta over the rows the slot covers
    F2_TRY(m->stage_d.reserve((size_t)b.rows * b.cols * 2));
    F2_TRY(place_rows(m, s, m->stage_w.as<uint16_t>(), m->stage_d.as<uint16_t>()));
    const size_t off = (s.transform == T_DIRECT) ? (size_t)s.row_off * cols : 0;
    F2_LAUNCH(launch_bf16_add_inplace(st, b.ptr + off, m->stage_d.as<uint16_t>() + off, rows * cols), "lora merge");
    F2_CHECK_HIP(hipStreamSynchronize(st));
    m->lora_merged = true;
    return FLUX2MI_OK;
}

// The unchecked form: A and B are TRUSTED to be [rank, in] and [out, rank] of the addressed weight's own in / out (callers that built
// the factors from flux2mi_dit_slot_shape, e.g. synthetic tests); files go through flux2mi_dit_merge_lora_checked.
extern "C" int flux2mi_dit_merge_lora(flux2mi_dit* m, const char* name, int dtype, const void* A, const void* B, int rank,
                                      float scale) {
    F2_REQUIRE(m && name && rank > 0, FLUX2MI_ERR_INVALID_ARG, "dit_merge_lora: bad argument");
    auto it = m->slots.find(name);
    if (it == m->slots.end()) {
        flux2mi_set_error("dit_merge_lora: unknown key '%s'", name);
        return FLUX2MI_ERR_UNKNOWN_KEY;
    }
    return flux2mi_dit_merge_lora_checked(m, name, dtype, A, rank, it->second.cols, dtype, B, it->second.rows, rank, scale);
}

// [out, in] of the weight a checkpoint key addresses (FLUX2MI_ERR_UNKNOWN_KEY when the model has none of that name)
extern "C" int flux2mi_dit_slot_shape(flux2mi_dit* m, const char* name, int64_t* rows, int64_t* cols) {
    F2_REQUIRE(m && name, FLUX2MI_ERR_INVALID_ARG, "dit_slot_shape: NULL argument");
    auto it = m->slots.find(name);
    if (it == m->slots.end()) {
        flux2mi_set_error("dit_slot_shape: unknown key '%s'", name);
        return FLUX2MI_ERR_UNKNOWN_KEY;
    }
    if (rows) *rows = it->second.rows;
    if (cols) *cols = it->second.cols;
    return FLUX2MI_OK;
}

extern "C" int flux2mi_dit_finalize(flux2mi_dit* m) {
    F2_REQUIRE(m, FLUX2MI_ERR_INVALID_ARG, "dit_finalize: NULL");
    F2_WORKSPACE_USE(m->ctx->device);
    for (const Buf& b : m->bufs) {
        if (b.got_rows() < b.rows) {
            flux2mi_set_error("dit_finalize: tensor '%s' missing (%d of %d rows loaded)", b.label.c_str(), b.got_rows(),
                              b.rows);
            return FLUX2MI_ERR_NOT_LOADED;
        }
    }
    char a[160];
    m->img_in = buf_by_label(m, "img_in");
    m->txt_in = buf_by_label(m, "txt_in");
    m->time_in = buf_by_label(m, "time_in.in");
    m->time_out = buf_by_label(m, "time_in.out");
    m->guid_in = buf_by_label(m, "guidance_in.in");
    m->guid_out = buf_by_label(m, "guidance_in.out");
    m->mod_img = buf_by_label(m, "mod_img");
    m->mod_txt = buf_by_label(m, "mod_txt");
    m->mod_single = buf_by_label(m, "mod_single");
    m->final_mod = buf_by_label(m, "final_mod");
    m->final_lin = buf_by_label(m, "final_lin");
    m->dbl.resize(m->cfg.n_double);
    for (int i = 0; i < m->cfg.n_double; ++i)
        for (int s = 0; s < 2; ++s) {
            const char* bs = s ? "img" : "txt";
            snprintf(a, sizeof a, "dbl%d.%s.qkv", i, bs);
            m->dbl[i].qkv[s] = weight_by_label(m, a);
            snprintf(a, sizeof a, "double_blocks.%d.%s_attn.proj.weight", i, bs);
            m->dbl[i].proj[s] = weight_by_label(m, a);
            snprintf(a, sizeof a, "double_blocks.%d.%s_attn.norm.query_norm.scale", i, bs);
            m->dbl[i].qn[s] = buf_by_label(m, a);
            snprintf(a, sizeof a, "double_blocks.%d.%s_attn.norm.key_norm.scale", i, bs);
            m->dbl[i].kn[s] = buf_by_label(m, a);
            snprintf(a, sizeof a, "double_blocks.%d.%s_mlp.0.weight", i, bs);
            m->dbl[i].mlp_in[s] = weight_by_label(m, a);
            snprintf(a, sizeof a, "double_blocks.%d.%s_mlp.2.weight", i, bs);
            m->dbl[i].mlp_out[s] = weight_by_label(m, a);
        }
    m->sgl.resize(m->cfg.n_single);
    for (int i = 0; i < m->cfg.n_single; ++i) {
        snprintf(a, sizeof a, "single_blocks.%d.linear1.weight", i);
        m->sgl[i].lin1 = weight_by_label(m, a);
        snprintf(a, sizeof a, "single_blocks.%d.linear2.weight", i);
        m->sgl[i].lin2 = weight_by_label(m, a);
        snprintf(a, sizeof a, "single_blocks.%d.norm.query_norm.scale", i);
        m->sgl[i].qn = buf_by_label(m, a);
        snprintf(a, sizeof a, "single_blocks.%d.norm.key_norm.scale", i);
        m->sgl[i].kn = buf_by_label(m, a);
    }
    // Score bound per block for the attention's fixed-offset softmax: after RMSNorm with weight w every
    // q (k) vector has |q| <= sqrt(128) max|w_q|, RoPE is a rotation, so |q.k| <= 128 max|w_q| max|w_k|
    // (Cauchy-Schwarz).  In the log2 domain with the 1/sqrt(128) scale: 16.32 max|w_q| max|w_k|; +3 % for
    // the bf16 roundings of q and k.
    {
        F2_CHECK_HIP(hipStreamSynchronize(m->ctx->stream));
        auto wmax = [&](const uint16_t* dev, float& out) -> int {
            uint16_t h[128];
            F2_CHECK_HIP(hipMemcpy(h, dev, sizeof h, hipMemcpyDeviceToHost));
            float mx = 0.f;
            for (int i = 0; i < 128; ++i) {
                uint32_t u = (uint32_t)h[i] << 16;
                float f;
                memcpy(&f, &u, 4);
                mx = fmaxf(mx, fabsf(f));
            }
            out = mx;
            return FLUX2MI_OK;
        };
        const float c = 128.0f / sqrtf(128.0f) * 1.4426950408889634f * 1.03f;
        // e4m3 exponents: |q_d| <= sqrt(128) max|w_q| q_mul (q carries the score scale log2(e) / sqrt(128)), |k_d| <= sqrt(128) max|w_k|
        auto e8 = [](float bound) { return bound > 0.f ? std::max(-30, std::min(30, (int)floorf(log2f(448.0f / bound)))) : 0; };
        const float qm = 1.4426950408889634f / sqrtf(128.0f), r128 = sqrtf(128.0f);
        for (auto& b : m->dbl) {
            float q0, q1, k0, k1;
            F2_TRY(wmax(b.qn[0], q0)); F2_TRY(wmax(b.qn[1], q1)); F2_TRY(wmax(b.kn[0], k0)); F2_TRY(wmax(b.kn[1], k1));
            b.score_bound = c * fmaxf(q0, q1) * fmaxf(k0, k1);
            b.q8_exp = e8(r128 * fmaxf(q0, q1) * qm);
            b.k8_exp = e8(r128 * fmaxf(k0, k1));
        }
        for (auto& b : m->sgl) {
            float q, k;
            F2_TRY(wmax(b.qn, q)); F2_TRY(wmax(b.kn, k));
            b.score_bound = c * q * k;
            b.q8_exp = e8(r128 * q * qm);
            b.k8_exp = e8(r128 * k);
        }
    }
    m->stage_raw.release();
    m->stage_bf16.release();
    m->stage_w.release();
    m->stage_q8.release();
    m->finalized = true;
    return FLUX2MI_OK;
}

// ------------------------------------------------------------------ pre-quantized checkpoint
// Flux2PrequantizedCheckpoint (Loading/PrequantizedCheckpoint.swift) + Flux2Pipeline.exportPrequantizedTransformer
// (Pipeline/Flux2Pipeline.swift:759-840): the resident weights in the library's OWN layout and naming — e4m3 bytes +
// per-row f32 scales for the block GEMMs, bf16 for the rest — so that a reload is a validated file -> HBM copy: no key
// mapping, no layout transform, no quantise pass, about half the bytes.  The save is atomic (temp file + rename); the
// load validates metadata, payload size, key set, dtypes and shapes BEFORE it touches the model.
namespace {
const char* const PREQ_FORMAT = "flux2mi-prequantized-v1";

std::string json_escape(const char* s) {
    std::string o;
    for (const char* p = s ? s : ""; *p; ++p) {
        if (*p == '"' || *p == '\\') { o += '\\'; o += *p; }
        else if ((unsigned char)*p < 0x20) o += ' ';
        else o += *p;
    }
    return o;
}
std::string cfg_string(const flux2mi_dit_config& c) {
    char b[200];
    snprintf(b, sizeof(b), "dim=%d,heads=%d,n_double=%d,n_single=%d,joint_dim=%d,guidance=%d", c.dim, c.heads, c.n_double, c.n_single,
             c.joint_dim, c.guidance_embeds);
    return b;
}
struct PreqEntry { std::string key, dtype; int rows, cols; const void* dev; size_t bytes; };
std::vector<PreqEntry> preq_entries(const flux2mi_dit* m) {
    std::vector<PreqEntry> v;
    for (const Buf& b : m->bufs) {
        if (b.fp8) {
            v.push_back({b.label + ".weight", "F8_E4M3", b.rows, b.cols, b.q, (size_t)b.rows * b.cols});
            v.push_back({b.label + ".scale", "F32", b.rows, 0, b.scale, (size_t)b.rows * 4});
        } else {
            v.push_back({b.label + ".weight", "BF16", b.rows, b.cols, b.ptr, (size_t)b.rows * b.cols * 2});
        }
    }
    return v;
}
}  // namespace

extern "C" int flux2mi_dit_export_prequantized(flux2mi_dit* m, const char* path, const char* source_fingerprint) {
    F2_REQUIRE(m && path && *path, FLUX2MI_ERR_INVALID_ARG, "dit_export_prequantized: bad argument");
    F2_WORKSPACE_USE(m->ctx->device);
    F2_REQUIRE(m->finalized, FLUX2MI_ERR_NOT_LOADED, "dit_export_prequantized: weights not finalized");
    F2_CHECK_HIP(hipSetDevice(m->ctx->device));
    F2_CHECK_HIP(hipStreamSynchronize(m->ctx->stream));
    const std::vector<PreqEntry> ent = preq_entries(m);
    // header
    std::string h = "{\"__metadata__\":{";
    h += "\"format\":\"" + std::string(PREQ_FORMAT) + "\",";
    h += "\"quantization\":\"" + std::string(m->cfg.weight_format == 1 ? "fp8" : "bf16") + "\",";
    h += "\"bits\":\"" + std::string(m->cfg.weight_format == 1 ? "8" : "16") + "\",";
    h += "\"component\":\"transformer\",";
    h += "\"config\":\"" + cfg_string(m->cfg) + "\",";
    h += "\"lora_baked\":\"" + std::string(m->lora_merged ? "true" : "false") + "\",";
    h += "\"source_fingerprint\":\"" + json_escape(source_fingerprint) + "\"}";
    uint64_t off = 0;
    size_t max_bytes = 0;
    for (const PreqEntry& e : ent) {
        char b[256];
        if (e.cols) snprintf(b, sizeof(b), ",\"%s\":{\"dtype\":\"%s\",\"shape\":[%d,%d],\"data_offsets\":[%llu,%llu]}", e.key.c_str(), e.dtype.c_str(),
                             e.rows, e.cols, (unsigned long long)off, (unsigned long long)(off + e.bytes));
        else snprintf(b, sizeof(b), ",\"%s\":{\"dtype\":\"%s\",\"shape\":[%d],\"data_offsets\":[%llu,%llu]}", e.key.c_str(), e.dtype.c_str(), e.rows,
                      (unsigned long long)off, (unsigned long long)(off + e.bytes));
        h += b;
        off += e.bytes;
        max_bytes = std::max(max_bytes, e.bytes);
    }
    h += "}";
    while (h.size() % 8) h += ' ';
    // atomic save: a crash or a full disk never leaves a truncated file at `path` (PrequantizedCheckpoint.swift, "Safety model")
    const std::string tmp = std::string(path) + ".tmp." + std::to_string((long)getpid());
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) {
        flux2mi_set_error("dit_export_prequantized: cannot create '%s': %s", tmp.c_str(), strerror(errno));
        return FLUX2MI_ERR_INVALID_ARG;
    }
    const uint64_t hlen = h.size();
    bool ok = fwrite(&hlen, 8, 1, f) == 1 && fwrite(h.data(), 1, h.size(), f) == h.size();
    std::vector<uint8_t> host(max_bytes);
    for (size_t i = 0; ok && i < ent.size(); ++i) {
        if (hipMemcpy(host.data(), ent[i].dev, ent[i].bytes, hipMemcpyDeviceToHost) != hipSuccess) {
            flux2mi_set_error("dit_export_prequantized: device read of '%s' failed", ent[i].key.c_str());
            fclose(f);
            unlink(tmp.c_str());
            return FLUX2MI_ERR_HIP;
        }
        ok = fwrite(host.data(), 1, ent[i].bytes, f) == ent[i].bytes;
    }
    ok = ok && fflush(f) == 0 && fsync(fileno(f)) == 0;
    ok = (fclose(f) == 0) && ok;
    if (!ok || rename(tmp.c_str(), path) != 0) {
        flux2mi_set_error("dit_export_prequantized: writing '%s' failed: %s", path, strerror(errno));
        unlink(tmp.c_str());
        return FLUX2MI_ERR_INVALID_ARG;
    }
    return FLUX2MI_OK;
}

extern "C" int flux2mi_dit_load_prequantized(flux2mi_dit* m, const char* path, const char* expected_fingerprint, int* lora_baked) {
    F2_REQUIRE(m && path, FLUX2MI_ERR_INVALID_ARG, "dit_load_prequantized: bad argument");
    F2_WORKSPACE_USE(m->ctx->device);
    MappedFile f;
    std::map<std::string, TensorInfo> ts;
    std::map<std::string, std::string> md;
    const uint8_t* data = nullptr;
    uint64_t hlen = 0;
    F2_TRY(parse_safetensors(path, f, ts, &data, &md, &hlen));
    // ---- validate everything before the model is touched
    const std::pair<const char*, std::string> expect[] = {
        {"format", PREQ_FORMAT}, {"quantization", m->cfg.weight_format == 1 ? "fp8" : "bf16"},
        {"bits", m->cfg.weight_format == 1 ? "8" : "16"}, {"component", "transformer"}, {"config", cfg_string(m->cfg)}};
    for (const auto& kv : expect) {
        auto it = md.find(kv.first);
        F2_REQUIRE(it != md.end() && it->second == kv.second, FLUX2MI_ERR_INVALID_ARG,
                   "dit_load_prequantized: '%s': metadata %s = '%s', this model needs '%s'", path, kv.first,
                   it == md.end() ? "<missing>" : it->second.c_str(), kv.second.c_str());
    }
    if (expected_fingerprint) {
        auto it = md.find("source_fingerprint");
        F2_REQUIRE(it != md.end() && it->second == expected_fingerprint, FLUX2MI_ERR_INVALID_ARG,
                   "dit_load_prequantized: '%s' was exported from different source weights (fingerprint '%s', expected '%s')", path,
                   it == md.end() ? "<missing>" : it->second.c_str(), expected_fingerprint);
    }
    uint64_t max_end = 0;
    for (const auto& kv : ts) max_end = std::max(max_end, kv.second.end);
    F2_REQUIRE(8 + hlen + max_end == f.size, FLUX2MI_ERR_INVALID_ARG,
               "dit_load_prequantized: '%s' is truncated or corrupt (%zu bytes on disk, header declares %llu)", path, f.size,
               (unsigned long long)(8 + hlen + max_end));
    const std::vector<PreqEntry> ent = preq_entries(m);
    F2_REQUIRE(ts.size() == ent.size(), FLUX2MI_ERR_INVALID_ARG, "dit_load_prequantized: '%s' holds %zu tensors, this model has %zu", path,
               ts.size(), ent.size());
    for (const PreqEntry& e : ent) {
        auto it = ts.find(e.key);
        F2_REQUIRE(it != ts.end(), FLUX2MI_ERR_INVALID_ARG, "dit_load_prequantized: '%s': tensor '%s' missing", path, e.key.c_str());
        const TensorInfo& t = it->second;
        const bool shape_ok = e.cols ? (t.shape.size() == 2 && t.shape[0] == e.rows && t.shape[1] == e.cols)
                                     : (t.shape.size() == 1 && t.shape[0] == e.rows);
        F2_REQUIRE(t.dtype == e.dtype && shape_ok && t.end - t.begin == e.bytes, FLUX2MI_ERR_SHAPE,
                   "dit_load_prequantized: '%s': tensor '%s' has the wrong dtype or shape", path, e.key.c_str());
    }
    // ---- copy (file mapping -> HBM)
    F2_CHECK_HIP(hipSetDevice(m->ctx->device));
    for (const PreqEntry& e : ent)
        F2_CHECK_HIP(hipMemcpy(const_cast<void*>(e.dev), data + ts[e.key].begin, e.bytes, hipMemcpyHostToDevice));
    for (Buf& b : m->bufs) b.mark(0, b.rows);
    m->lora_merged = md.count("lora_baked") && md["lora_baked"] == "true";
    if (lora_baked) *lora_baked = m->lora_merged ? 1 : 0;
    m->finalized = false;
    return flux2mi_dit_finalize(m);
}

// ------------------------------------------------------------------ forward
namespace {

// row pitch of the block-scale table: row blocks of 16 + the 12 bytes a scalar load may read behind the last one, 16-byte multiples
int bscale_ld(int S) { return (((S + 15) / 16 + 12) + 15) & ~15; }

int ensure_workspace(flux2mi_dit* m, int S, int s_img) {
    const int d = m->cfg.dim, H = m->cfg.heads;
    const size_t Sz = (size_t)S;
    F2_TRY(m->hs.reserve(Sz * d * 4));
    F2_TRY(m->xn.reserve(Sz * d * 2));
    F2_TRY(m->qkv.reserve(Sz * 3 * d * 2));
    const size_t vt_bytes = (size_t)H * 128 * attn_s_pad(S) * 2;
    // the columns of the padded keys must read as finite numbers (their probabilities are exactly 0).  The V^T
    // GEMM only writes the valid keys, so the image is cleared whenever the sequence length (= the padding) changes.
    if (m->vt.bytes < vt_bytes || m->ws_S != S) {
        F2_TRY(m->vt.reserve(vt_bytes));
        F2_CHECK_HIP(hipMemsetAsync(m->vt.ptr, 0, m->vt.bytes, m->ctx->stream));
    }
    F2_TRY(m->comb.reserve(Sz * 4 * d * 2));
    if (m->ctx->n_cu > 0) F2_TRY(m->ctx->attn_ws.reserve(attention64b_split_ws_bytes(m->ctx->n_cu)));   // KV-split pieces (attention64b.hip)
    F2_TRY(m->mods.reserve((size_t)(256 + 2 * d + 6 * d + 6 * d + 3 * d + 2 * d) * 4));
    F2_TRY(m->rope.reserve(Sz * 128 * 4 * 2 + Sz * 4 * 4));   // cos, sin f32 [S,128] + positions f32 [S,4]
    F2_TRY(m->lat_bf16.reserve((size_t)s_img * 128 * 2));
    if (m->cfg.weight_format == 1) {
        F2_TRY(m->xq.reserve(Sz * 4 * d));
        F2_TRY(m->xs.reserve(Sz * 4));
        F2_TRY(m->qk8.reserve(Sz * 2 * d));   // e4m3 [S][q | k] of the current block (attention64b.hip, QK8)
        // block scales of the block-quantised activations (GemmProblem::a_bscale): [4d / 128 K-tiles][row blocks of 16, padded]; the pad
        // bytes are read (never used) by the scalar loads of a tile's last wave: unit scales
        // (their bytes: a buffer of their own — the producing GEMM still reads its LayerNorm operand from xq while its epilogue writes them)
        F2_TRY(m->xq2.reserve(Sz * 4 * d));
        const size_t bs_bytes = (size_t)(4 * d / 128) * (size_t)bscale_ld(S);
        if (m->xbs.bytes < bs_bytes || m->ws_S != S) {
            F2_TRY(m->xbs.reserve(bs_bytes));
            F2_CHECK_HIP(hipMemsetAsync(m->xbs.ptr, 0x7F, m->xbs.bytes, m->ctx->stream));
        }
    }
    m->ws_S = S;
    return FLUX2MI_OK;
}

GemmProblem mk(const uint16_t* A, int lda, const uint16_t* W, int ldw, void* C, int ldc, int M, int N, int K, int epi) {
    GemmProblem p;
    memset(&p, 0, sizeof p);
    p.A = A; p.lda = lda; p.W = W; p.ldw = ldw; p.C = C; p.ldc = ldc;
    p.M = M; p.N = N; p.K = K; p.epi = epi; p.alpha = 1.0f;
    return p;
}

int run_gemm(flux2mi_dit* m, const GemmProblem* p_in, int n, int extra_cls, const char* what) {
    double flops = 0.0;
    for (int i = 0; i < n; ++i) flops += 2.0 * p_in[i].M * (double)p_in[i].N * p_in[i].K;
    // split-K workspaces for the launches that would leave most of the chip idle (the N = d GEMMs at small sequence lengths:
    // gemm_splitk_ws_bytes says 0 for everything else, the 1024^2 shapes of every model among them)
    GemmProblem pw[2];
    const GemmProblem* p = p_in;
    if (n <= 2 && (p_in[0].epi == EPI_GATE_RESID || p_in[0].epi == EPI_F32) && !p_in[0].splitk_ws) {
        size_t need[2] = {0, 0}, tot = 0;
        for (int i = 0; i < n; ++i) {
            need[i] = (gemm_splitk_ws_bytes(p_in[i].M, p_in[i].N, p_in[i].K, p_in[i].fp8 != 0) + 255) & ~(size_t)255;
            tot += need[i];
        }
        if (tot) {
            F2_TRY(m->splitk_ws.reserve(tot));
            size_t off = 0;
            for (int i = 0; i < n; ++i) {
                pw[i] = p_in[i];
                if (need[i]) {
                    pw[i].splitk_ws = reinterpret_cast<float*>(m->splitk_ws.as<uint8_t>() + off);
                    pw[i].splitk_ws_bytes = need[i];
                }
                off += need[i];
            }
            p = pw;
        }
    }
    prof_before(m->ctx, PROF_DIT_GEMM | extra_cls);
    const int r = launch_gemm(m->ctx->stream, p, n);
    prof_after(m->ctx, PROF_DIT_GEMM | extra_cls, flops);
    F2_LAUNCH(r, what);
    return FLUX2MI_OK;
}

}  // namespace

// Reference-token K/V of every layer, written by the extraction pass and read by the cached passes
// (Transformer/TransformerKVCache.swift).  Per layer: K post-RoPE [s_ref, d] bf16 and V^T
// [heads*128, s_ref] bf16 in key order.  Layers: double blocks first, then single blocks.
struct flux2mi_kv_cache {
    flux2mi_dit* dit = nullptr;
    int s_txt = 0, s_ref = 0;
    bool filled = false;
    uint16_t* data = nullptr;
    size_t layer_elems = 0;   // 2 * s_ref * d
    uint16_t* k(int layer) const { return data + (size_t)layer * layer_elems; }
    uint16_t* v(int layer) const { return data + (size_t)layer * layer_elems + layer_elems / 2; }
};

namespace {

enum FwdMode { FWD_PLAIN = 0, FWD_KV_EXTRACT = 1, FWD_KV_CACHED = 2 };

// Joint row layout of one forward:
//   plain       [ txt | image tokens (output first, appended references after) ]
//   kv extract  [ txt | ref | out ]   reference queries see keys [txt | ref] only (Flux2Attention.swift:296-306)
//   kv cached   [ txt | gap | out ]   the gap keeps the rows of the extraction pass: its K rows / V^T
//                                     columns are filled from the cache, nothing else touches it
// so position ids, RoPE rows and key indices are the same in both KV passes.
int dit_forward_impl(flux2mi_dit* m, int mode, flux2mi_kv_cache* cache, const float* latents, int s_lat,
                     const float* ref_latents, int s_ref, const void* txt_emb, int s_txt, float timestep,
                     float guidance, const int32_t* img_ids, const int32_t* txt_ids, const int32_t* ref_ids, float* out,
                     int s_out) {
    F2_CHECK_HIP(hipSetDevice(m->ctx->device));
    hipStream_t st = m->ctx->stream;
    const int d = m->cfg.dim, H = m->cfg.heads, J = m->cfg.joint_dim;
    const int gap = mode == FWD_PLAIN ? 0 : s_ref;            // rows between the text and the live image rows
    const int S = s_txt + gap + s_lat;                         // rows of the layout (= keys of the attention)
    const int img0 = mode == FWD_KV_CACHED ? s_txt + gap : s_txt;   // first live image-stream row
    const int n_img = S - img0;                                // live image-stream rows
    const int out0 = mode == FWD_PLAIN ? s_txt : s_txt + gap;  // first output row
    const int s_pad = attn_s_pad(S);
    F2_TRY(ensure_workspace(m, S, S - s_txt));

    float* hs = m->hs.as<float>();
    uint16_t* xn = m->xn.as<uint16_t>();
    uint16_t* qkv = m->qkv.as<uint16_t>();
    uint16_t* vt = m->vt.as<uint16_t>();
    uint16_t* comb = m->comb.as<uint16_t>();
    float* cos_tab = m->rope.as<float>();
    float* sin_tab = cos_tab + (size_t)S * 128;
    float* pos_tab = sin_tab + (size_t)S * 128;
    // mods layout (this forward's own buffer, or the row flux2mi_dit_prepare_steps computed for this timestep)
    const size_t mods_stride = 256 + (size_t)19 * d;
    int pre_row = -1;
    if (m->step_cache.on && mode == FWD_PLAIN && m->step_cache.guidance == guidance)
        for (size_t k = 0; k < m->step_cache.ts.size(); ++k)
            if (m->step_cache.ts[k] == timestep) { pre_row = (int)k; break; }
    float* emb = pre_row >= 0 ? m->step_cache.mods_all.as<float>() + (size_t)pre_row * mods_stride : m->mods.as<float>();   // [256]
    float* tmp = emb + 256;                    // [d]
    float* temb = tmp + d;                     // [d]
    float* mod_img = temb + d;                 // [6d]  set0 (shift,scale,gate), set1 (shift,scale,gate)
    float* mod_txt = mod_img + 6 * d;          // [6d]
    float* mod_sgl = mod_txt + 6 * d;          // [3d]
    float* mod_fin = mod_sgl + 3 * d;          // [2d]  (shift, scale)

    // ---- timestep (+guidance) embedding: sinusoid(1000 t) -> Linear -> SiLU -> Linear
    // (Flux2Transformer.swift:145-149, Flux2Embeddings.swift:124-142)
    if (pre_row < 0) {
    F2_LAUNCH(launch_sinusoid(st, timestep * 1000.0f, emb), "sinusoid");
    F2_LAUNCH(launch_gemv(st, m->time_in, emb, tmp, d, 256, 0, 0), "time_in.in");
    F2_LAUNCH(launch_gemv(st, m->time_out, tmp, temb, d, d, 1, 0), "time_in.out");
    if (m->cfg.guidance_embeds) {
        F2_LAUNCH(launch_sinusoid(st, guidance * 1000.0f, emb), "sinusoid(g)");
        F2_LAUNCH(launch_gemv(st, m->guid_in, emb, tmp, d, 256, 0, 0), "guidance_in.in");
        F2_LAUNCH(launch_gemv(st, m->guid_out, tmp, temb, d, d, 1, 1), "guidance_in.out");
    }
    // ---- modulation vectors, once per forward (Flux2Transformer.swift:160-161, 256; Flux2Modulation.swift:49-75)
    F2_LAUNCH(launch_gemv(st, m->mod_img, temb, mod_img, 6 * d, d, 1, 0), "mod_img");
    F2_LAUNCH(launch_gemv(st, m->mod_txt, temb, mod_txt, 6 * d, d, 1, 0), "mod_txt");
    F2_LAUNCH(launch_gemv(st, m->mod_single, temb, mod_sgl, 3 * d, d, 1, 0), "mod_single");
    F2_LAUNCH(launch_gemv(st, m->final_mod, temb, mod_fin, 2 * d, d, 1, 0), "final_mod");
    }
    // ---- RoPE tables from the ids of this call (Flux2Transformer.swift:153-154, :370-371)
    flux2mi_dit::StepCache& sc = m->step_cache;
    const bool sc_on = sc.on && mode == FWD_PLAIN;
    if (!sc_on) sc.txt_ok = sc.rope_ok = false;   // any other forward rewrites the tables / is not the generation's own
    if (mode == FWD_PLAIN) {
        const bool reuse = sc_on && sc.rope_ok && sc.txt_ids == txt_ids && sc.img_ids == img_ids && sc.s_txt == s_txt && sc.s_lat == s_lat;
        if (!reuse) F2_LAUNCH(launch_rope_tables(st, txt_ids, s_txt, img_ids, s_lat, cos_tab, sin_tab, pos_tab), "rope_tables");
        if (sc_on) { sc.rope_ok = true; sc.txt_ids = txt_ids; sc.img_ids = img_ids; if (sc.s_txt != s_txt || sc.s_lat != s_lat) sc.txt_ok = false; sc.s_txt = s_txt; sc.s_lat = s_lat; }
    } else {
        // [txt | ref] rows (the gap rows of a cached pass get their ids too when given; unused otherwise)
        if (mode == FWD_KV_EXTRACT) {
            F2_LAUNCH(launch_rope_tables(st, txt_ids, s_txt, ref_ids, s_ref, cos_tab, sin_tab, pos_tab), "rope_tables(txt,ref)");
        } else {
            F2_LAUNCH(launch_rope_tables(st, txt_ids, s_txt, img_ids, 0, cos_tab, sin_tab, pos_tab), "rope_tables(txt)");
        }
        F2_LAUNCH(launch_rope_tables(st, txt_ids, 0, img_ids, s_lat, cos_tab + (size_t)out0 * 128, sin_tab + (size_t)out0 * 128,
                                     pos_tab + (size_t)out0 * 4),
                  "rope_tables(out)");
    }

    // ---- input projections into the joint residual stream (:137-138, :357-362)
    F2_LAUNCH(launch_f32_to_bf16(st, latents, m->lat_bf16.as<uint16_t>(), (int64_t)s_lat * 128), "latents->bf16");
    {
        GemmProblem p[2];
        p[0] = mk(reinterpret_cast<const uint16_t*>(txt_emb), J, m->txt_in, J, hs, d, s_txt, d, J, EPI_F32);
        p[1] = mk(m->lat_bf16.as<uint16_t>(), 128, m->img_in, 128, hs + (size_t)out0 * d, d, s_lat, d, 128, EPI_F32);
        const size_t txt_bytes = (size_t)s_txt * d * 4;
        if (sc_on && sc.txt_ok && sc.txt == txt_emb) {   // the text rows as the previous step of this generation projected them
            F2_CHECK_HIP(hipMemcpyAsync(hs, sc.hs_txt.ptr, txt_bytes, hipMemcpyDeviceToDevice, st));
            F2_TRY(run_gemm(m, p + 1, 1, 0, "embedder(img)"));
        } else {
            F2_TRY(run_gemm(m, p, 2, 0, "embedders"));
            if (sc_on) {
                F2_TRY(sc.hs_txt.reserve(txt_bytes));
                F2_CHECK_HIP(hipMemcpyAsync(sc.hs_txt.ptr, hs, txt_bytes, hipMemcpyDeviceToDevice, st));
                sc.txt = txt_emb; sc.txt_ok = true;
            }
        }
    }
    if (mode == FWD_KV_EXTRACT) {
        uint16_t* rb = m->lat_bf16.as<uint16_t>() + (size_t)s_lat * 128;
        F2_LAUNCH(launch_f32_to_bf16(st, ref_latents, rb, (int64_t)s_ref * 128), "ref latents->bf16");
        GemmProblem p = mk(rb, 128, m->img_in, 128, hs + (size_t)s_txt * d, d, s_ref, d, 128, EPI_F32);
        F2_TRY(run_gemm(m, &p, 1, 0, "embedder(ref)"));
    }

    // fp8 mode: the block GEMMs read their A operand from xq (e4m3, same [row, ld] layout as the
    // bf16 buffer it shadows) with one scale per token in xs.
    const bool f8 = m->cfg.weight_format == 1;
    uint8_t* xq = f8 ? m->xq.as<uint8_t>() : nullptr;
    float* xs = f8 ? m->xs.as<float>() : nullptr;
    using W = flux2mi_dit::W;
    // rows [r0, r0+M) of the bf16 activation `a` (row stride ld) times weight w
    auto blk = [&](const uint16_t* a, int ld, int r0, const W& w, void* C, int ldc, int M, int N, int K, int epi) {
        GemmProblem p;
        if (f8) {
            p = mk(reinterpret_cast<const uint16_t*>(xq + (size_t)r0 * ld), ld, w.w, K, C, ldc, M, N, K, epi);
            p.fp8 = 1;
            p.a_scale = xs + r0;
            p.w_scale = w.scale;
        } else {
            p = mk(a + (size_t)r0 * ld, ld, w.w, K, C, ldc, M, N, K, epi);
        }
        return p;
    };
    // rows [r, ...) of a weight (bf16: 2 B per element, e4m3: 1 B) with their per-row scales
    auto wrows = [&](const W& w, int r, int K) {
        W o = w;
        o.w = reinterpret_cast<const uint16_t*>(reinterpret_cast<const uint8_t*>(w.w) + (size_t)r * K * (f8 ? 1 : 2));
        if (o.scale) o.scale += r;
        return o;
    };
    // V^T [d, rows] = W_v [d, K] (activation rows [r0, r0 + rows))^T into vt columns r0.. (key order permuted in the store)
    auto vt_problem = [&](const W& wv, int r0, int rows) {
        GemmProblem p;
        if (f8) {
            p = mk(wv.w, d, reinterpret_cast<const uint16_t*>(xq + (size_t)r0 * d), d, vt + r0, s_pad, d, rows, d, EPI_BF16);
            p.fp8 = 1;
            p.a_scale = wv.scale;        // the A operand is the weight here: its per-row scales
            p.w_scale = xs + r0;         // the W operand is the activation: its per-token scales
        } else {
            p = mk(wv.w, d, xn + (size_t)r0 * d, d, vt + r0, s_pad, d, rows, d, EPI_BF16);
        }
        p.col_perm16 = 1;
        return p;
    };
    // LayerNorm + modulate of the text rows with (sh0, sc0) and of the live image rows with (sh1, sc1)
    auto ln_mod_launch = [&](const float* sh0, const float* sc0, const float* sh1, const float* sc1) {
        if (img0 == s_txt) {
            return f8 ? launch_layernorm_modulate_fp8(st, hs, xq, xs, S, d, s_txt, sh0, sc0, sh1, sc1)
                      : launch_layernorm_modulate(st, hs, xn, S, d, s_txt, sh0, sc0, sh1, sc1);
        }
        const size_t o = (size_t)img0 * d;
        const int r = f8 ? launch_layernorm_modulate_fp8(st, hs, xq, xs, s_txt, d, s_txt, sh0, sc0, sh0, sc0)
                         : launch_layernorm_modulate(st, hs, xn, s_txt, d, s_txt, sh0, sc0, sh0, sc0);
        if (r) return r;
        return f8 ? launch_layernorm_modulate_fp8(st, hs + o, xq + o, xs + img0, n_img, d, 0, sh1, sc1, sh1, sc1)
                  : launch_layernorm_modulate(st, hs + o, xn + o, n_img, d, 0, sh1, sc1, sh1, sc1);
    };
    // algorithmic bytes of one call: the f32 residual rows read once, the normalised rows written once (bf16, or e4m3 + a scale)
    auto ln_mod = [&](const float* sh0, const float* sc0, const float* sh1, const float* sc1) {
        prof_before(m->ctx, PROF_LN_MODULATE);
        const int r = ln_mod_launch(sh0, sc0, sh1, sc1);
        prof_after(m->ctx, PROF_LN_MODULATE, (double)(s_txt + n_img) * ((double)d * (f8 ? 5.0 : 6.0) + (f8 ? 4.0 : 0.0)));
        return r;
    };
    // quantise rows of a bf16 activation into xq/xs before the GEMM that consumes it (fp8 only)
    auto quant = [&](const uint16_t* a, int ld, int cols) {
        return f8 ? launch_quantize_rows_fp8(st, a, ld, xq, ld, xs, S, cols) : 0;
    };
    // fp8 models, plain forwards: the activations of proj / mlp_out / linear2 in 16-row x 128-column e4m3 blocks with power-of-two
    // scales (kernels.h, GemmProblem::a_bscale), written by the producing epilogues instead of a row-quantise pass over bf16 copies.
    // Row ranges start at multiples of 64 (the scale table is read by aligned scalar loads).  FLUX2MI_FP8_BLOCKS (dev library): 0 = the
    // row-wise activations with their quantisation passes (A/B runs), 2 = blocks by stand-alone quantisation passes instead of the fused
    // epilogues (bisection aid)
    static const int bsa_mode = [] {
        const char* e = F2_DEV_ENV("FLUX2MI_FP8_BLOCKS");
        return e ? atoi(e) : 1;
    }();
    const bool bsa_on = bsa_mode != 0, bsa_fused = bsa_mode == 1;
    const bool bsa = f8 && bsa_on && mode == FWD_PLAIN && (s_txt & 63) == 0 && (img0 & 63) == 0 && d % 128 == 0;
    const int bs_ld = bscale_ld(S);
    uint8_t* xbs = bsa ? m->xbs.as<uint8_t>() : nullptr;
    uint8_t* xq2 = bsa ? m->xq2.as<uint8_t>() : nullptr;
    // the A operand of a consuming GEMM: rows from r0 of the block-quantised buffer (row pitch = the problem's lda)
    auto with_blocks = [&](GemmProblem& q, int r0) {
        q.A = reinterpret_cast<const uint16_t*>(xq2 + (size_t)r0 * q.lda);
        q.a_scale = nullptr;
        q.a_bscale = xbs + r0 / 16;
        q.a_bscale_ld = bs_ld;
    };
    // the SwiGLU hidden of a producing GEMM: rows from r0 of xq2 (row pitch ld), hidden column 0 at byte column c0 = K-tile c0 / 128
    auto hidden_blocks = [&](GemmProblem& q, int r0, int ld, int c0) {
        if (!bsa_fused) return;
        q.c2_q8 = xq2 + (size_t)r0 * ld + c0;
        q.c2_q8_ld = ld;
        q.c2_bs = xbs + (size_t)(c0 / 128) * bs_ld + r0 / 16;
        q.c2_bs_ld = bs_ld;
    };
    // the attention output (bf16 comb columns [0, d), row pitch ld) into blocks by its own pass: where the attention kernel does not write
    // them itself (e4m3 Q K^T switched off in the dev library)
    auto quant_attn_blocks = [&](int ld) { return launch_quantize_blocks_fp8(st, comb, ld, xq2, ld, xbs, bs_ld, S, d); };
    // joint attention per query range; keys = all S rows except for the reference queries of the
    // extraction pass, which see [txt | ref] only.  Also moves the reference K / V^T to / from the cache.
    // the score scale log2(e)/sqrt(128) rides on Q from the norm+RoPE kernel (one rounding, as before), so the
    // attention's softmax needs no multiply (FLUX2MI_ATTN_PRESCALE=0: scale inside the attention, for A/B runs)
    static const bool q_pre = [] {
        const char* e = F2_DEV_ENV("FLUX2MI_ATTN_PRESCALE");
        return e ? atoi(e) != 0 : true;
    }();
    const float q_mul = q_pre ? 1.4426950408889634f / sqrtf(128.0f) : 1.0f;
    // fp8 models, plain forwards: Q K^T on e4m3 copies of q and k (written by the norm + RoPE kernel in front of this call; the KV-cache
    // passes restore bf16 reference keys and keep the bf16 form).  FLUX2MI_ATTN_QK8=0 (dev library): bf16 Q K^T for A/B runs
    static const bool qk8_on = [] {
        const char* e = F2_DEV_ENV("FLUX2MI_ATTN_QK8");
        return e ? atoi(e) != 0 : true;
    }();
    const bool qk8 = f8 && qk8_on && q_pre && mode == FWD_PLAIN;
    const bool attn_blocks = bsa && bsa_fused && qk8;   // the attention writes its output as block-scaled e4m3 itself
    uint8_t* qk8_buf = qk8 ? m->qk8.as<uint8_t>() : nullptr;
    auto attend = [&](int layer, uint16_t* o_buf, int ldo, float score_bound, int q8_exp, int k8_exp) -> int {
        if (m->attn_softmax_mode == 1) score_bound = 0.f;   // no bound given: launch_attention takes the data-offset form
        if (qk8) {
            prof_before(m->ctx, PROF_ATTENTION);
            // (bsa: the output straight into proj's / linear2's block-scaled operand, row pitch ldo bytes, K-tiles 0 .. d / 128 - 1)
            const int ar = launch_attention64b_qk8(st, qk8_buf, qk8_buf + d, 2 * d, q8_exp, k8_exp, vt, s_pad, o_buf, ldo, S, S, H,
                                                   !attention_uses_fixed_offset(score_bound, true), attn_blocks ? xq2 : nullptr, ldo, xbs, bs_ld,
                                                   m->ctx->n_cu, m->ctx->attn_ws.as<float>());
            prof_after(m->ctx, PROF_ATTENTION, 4.0 * (double)S * S * d);
            F2_LAUNCH(ar, "attention(e4m3 QK)");
            return FLUX2MI_OK;
        }
        if (mode == FWD_KV_EXTRACT) {
            F2_CHECK_HIP(hipMemcpy2DAsync(cache->k(layer), (size_t)d * 2, qkv + (size_t)s_txt * 3 * d + d, (size_t)3 * d * 2,
                                          (size_t)d * 2, (size_t)s_ref, hipMemcpyDeviceToDevice, st));
            F2_LAUNCH(launch_vt_keys_copy(st, vt, s_pad, cache->v(layer), H, s_txt, s_ref, 1), "kv_cache(V^T out)");
        } else if (mode == FWD_KV_CACHED) {
            F2_CHECK_HIP(hipMemcpy2DAsync(qkv + (size_t)s_txt * 3 * d + d, (size_t)3 * d * 2, cache->k(layer), (size_t)d * 2,
                                          (size_t)d * 2, (size_t)s_ref, hipMemcpyDeviceToDevice, st));
            F2_LAUNCH(launch_vt_keys_copy(st, vt, s_pad, cache->v(layer), H, s_txt, s_ref, 0), "kv_cache(V^T in)");
        }
        struct Range { int q0, nq, nk; } rg[3];
        int nr = 0;
        if (mode == FWD_PLAIN) rg[nr++] = {0, S, S};
        else {
            rg[nr++] = {0, s_txt, S};
            if (mode == FWD_KV_EXTRACT) rg[nr++] = {s_txt, s_ref, s_txt + s_ref};
            rg[nr++] = {out0, s_lat, S};
        }
        for (int i = 0; i < nr; ++i) {
            prof_before(m->ctx, PROF_ATTENTION);
            const int ar = launch_attention(st, qkv + (size_t)rg[i].q0 * 3 * d, qkv + d, 3 * d, vt, s_pad,
                                            o_buf + (size_t)rg[i].q0 * ldo, ldo, rg[i].nq, rg[i].nk, H, score_bound, q_pre,
                                            mode == FWD_PLAIN ? m->ctx->attn_ws.as<float>() : nullptr, m->ctx->n_cu);   // KV-split: plain forwards only
            prof_after(m->ctx, PROF_ATTENTION, 4.0 * (double)rg[i].nq * rg[i].nk * d);
            F2_LAUNCH(ar, "attention");
        }
        return FLUX2MI_OK;
    };

    // FLUX2MI_QK_FUSE=0: QK norm + RoPE as a separate kernel after the qkv projection (A/B runs).  The fused form needs
    // whole heads inside a 256-column tile and the 256x256 bf16 kernel (>= 100 tiles, see launch_gemm).
    static const bool qk_fuse_on = [] {
        const char* e = F2_DEV_ENV("FLUX2MI_QK_FUSE");
        return e ? atoi(e) != 0 : true;
    }();
    const bool qk_fuse_ok = qk_fuse_on && d % 256 == 0 && !F2_DEV_ENV("FLUX2MI_GEMM");
    // decided on the whole sequence S, not on the live rows of this pass: a KV-cached pass (fewer live rows) must
    // treat q / k exactly like the extraction pass whose keys it reuses
    const bool qk_fuse_dbl = qk_fuse_ok && (long)((S + 255) / 256) * ((3 * d + 255) / 256) >= 100;
    // V^T straight out of a GEMM with the operands' roles swapped (A = the V rows of the fused weight, W = the
    // activation, C = vt, key order permuted in the store): no V round trip through qkv, no transpose kernel.
    // Plain forwards only (the KV-cache passes move reference columns of vt around on unaligned boundaries).
    static const bool vt_gemm_on = [] {
        const char* e = F2_DEV_ENV("FLUX2MI_VT_GEMM");
        return e ? atoi(e) != 0 : true;
    }();
    // (only when the V^T launch fills the chip on its own: at 256x256 it is 36 workgroups and the step got 12 % slower)
    const bool vt_gemm = vt_gemm_on && mode == FWD_PLAIN && (s_txt & 15) == 0 && (long)((d + 255) / 256) * ((S + 255) / 256) >= 160;
    // ---- double-stream blocks (Flux2TransformerBlock.swift:80-168)
    // flux2mi_dit_forward_block: only block tap.layer runs, from tap.hs_in to tap.hs_out ([S, d] f32, text rows first)
    const flux2mi_dit::BlockTap tap = mode == FWD_PLAIN ? m->tap : flux2mi_dit::BlockTap();
    // (hs_in == nullptr: observe only — every block runs on the forward's own stream, the stream behind tap.layer is copied out)
    auto tap_skip = [&](int layer) { return tap.layer >= 0 && tap.hs_in != nullptr && layer != tap.layer; };
    auto tap_in = [&](int layer) -> hipError_t {
        return layer == tap.layer && tap.hs_in ? hipMemcpyAsync(hs, tap.hs_in, (size_t)S * d * 4, hipMemcpyDeviceToDevice, st) : hipSuccess;
    };
    auto tap_out = [&](int layer) -> hipError_t {
        return layer == tap.layer ? hipMemcpyAsync(tap.hs_out, hs, (size_t)S * d * 4, hipMemcpyDeviceToDevice, st) : hipSuccess;
    };
    for (int i = 0; i < m->cfg.n_double; ++i) {
        if (tap_skip(i)) continue;
        F2_CHECK_HIP(tap_in(i));
        const flux2mi_dit::Dbl& w = m->dbl[i];
        // attention half: LN -> modulate(set 0) per stream
        F2_LAUNCH(ln_mod(mod_txt + 0 * d, mod_txt + 1 * d, mod_img + 0 * d, mod_img + 1 * d), "ln_mod(attn)");
        {
            GemmProblem p[2];
            const int nq = (qk_fuse_dbl && vt_gemm) ? 2 * d : 3 * d;   // [q|k] only: V^T comes from its own GEMM below
            p[0] = blk(xn, d, 0, w.qkv[0], qkv, 3 * d, s_txt, nq, d, EPI_BF16);
            p[1] = blk(xn, d, img0, w.qkv[1], qkv + (size_t)img0 * 3 * d, 3 * d, n_img, nq, d, EPI_BF16);
            if (qk_fuse_dbl) {   // per-stream norm weights: one set per problem
                for (int r = 0; r < 2; ++r) {
                    p[r].qk_qw = w.qn[r]; p[r].qk_kw = w.kn[r]; p[r].qk_pos = pos_tab;
                    p[r].qk_cols = 2 * d; p[r].qk_row0 = r ? img0 : 0; p[r].qk_qmul = q_mul;
                    p[r].qk_l2t16 = log2f(2000.0f) / 16.0f;
                    if (qk8) { p[r].qk8 = qk8_buf; p[r].qk8_ld = 2 * d; p[r].qk8_qmul = ldexpf(1.0f, w.q8_exp); p[r].qk8_kmul = ldexpf(1.0f, w.k8_exp); }
                }
            }
            F2_TRY(run_gemm(m, p, 2, 0, "dbl.qkv"));
        }
        if (qk_fuse_dbl && vt_gemm) {
            GemmProblem pv[2];
            pv[0] = vt_problem(wrows(w.qkv[0], 2 * d, d), 0, s_txt);
            pv[1] = vt_problem(wrows(w.qkv[1], 2 * d, d), img0, n_img);
            F2_TRY(run_gemm(m, pv, 2, 0, "dbl.vt"));
        } else if (qk_fuse_dbl) {
            F2_LAUNCH(launch_v_transpose_only(st, qkv, 3 * d, S, H, vt, s_pad), "v_transpose");
        } else
        F2_LAUNCH(launch_qk_norm_rope_vt(st, qkv, 3 * d, S, H, s_txt, w.qn[0], w.kn[0], w.qn[1], w.kn[1], cos_tab, sin_tab,
                                         vt, s_pad, q_mul, qk8_buf, w.q8_exp, w.k8_exp), "qk_norm_rope");
        F2_TRY(attend(i, comb, d, w.score_bound, w.q8_exp, w.k8_exp));
        if (bsa && !attn_blocks) F2_LAUNCH(quant_attn_blocks(d), "quantize(attn out, blocks)");
        else if (!bsa) F2_LAUNCH(quant(comb, d, d), "quantize(attn out)");
        {
            GemmProblem p[2];
            p[0] = blk(comb, d, 0, w.proj[0], hs, d, s_txt, d, d, EPI_GATE_RESID);
            p[0].vec = mod_txt + 2 * d;
            p[1] = blk(comb, d, img0, w.proj[1], hs + (size_t)img0 * d, d, n_img, d, d, EPI_GATE_RESID);
            p[1].vec = mod_img + 2 * d;
            if (bsa) { with_blocks(p[0], 0); with_blocks(p[1], img0); }
            F2_TRY(run_gemm(m, p, 2, PROF_PROJ_MLPOUT, "dbl.proj"));
        }
        // feed-forward half: LN -> modulate(set 1) -> Linear(d->6d) SwiGLU -> Linear(3d->d)
        F2_LAUNCH(ln_mod(mod_txt + 3 * d, mod_txt + 4 * d, mod_img + 3 * d, mod_img + 4 * d), "ln_mod(ff)");
        {
            GemmProblem p[2];
            p[0] = blk(xn, d, 0, w.mlp_in[0], nullptr, 0, s_txt, 6 * d, d, EPI_SWIGLU);
            p[0].C2 = comb; p[0].ldc2 = 3 * d;
            p[1] = blk(xn, d, img0, w.mlp_in[1], nullptr, 0, n_img, 6 * d, d, EPI_SWIGLU);
            p[1].C2 = comb + (size_t)img0 * 3 * d; p[1].ldc2 = 3 * d;
            if (bsa) { hidden_blocks(p[0], 0, 3 * d, 0); hidden_blocks(p[1], img0, 3 * d, 0); }   // straight into mlp_out's operand
            F2_TRY(run_gemm(m, p, 2, 0, "dbl.mlp_in"));
        }
        if (!bsa) F2_LAUNCH(quant(comb, 3 * d, 3 * d), "quantize(ff hidden)");
        else if (!bsa_fused) F2_LAUNCH(launch_quantize_blocks_fp8(st, comb, 3 * d, xq2, 3 * d, xbs, bs_ld, S, 3 * d), "quantize(ff hidden, blocks)");
        {
            GemmProblem p[2];
            p[0] = blk(comb, 3 * d, 0, w.mlp_out[0], hs, d, s_txt, d, 3 * d, EPI_GATE_RESID);
            p[0].vec = mod_txt + 5 * d;
            p[1] = blk(comb, 3 * d, img0, w.mlp_out[1], hs + (size_t)img0 * d, d, n_img, d, 3 * d, EPI_GATE_RESID);
            p[1].vec = mod_img + 5 * d;
            if (bsa) { with_blocks(p[0], 0); with_blocks(p[1], img0); }
            F2_TRY(run_gemm(m, p, 2, PROF_PROJ_MLPOUT, "dbl.mlp_out"));
        }
        F2_CHECK_HIP(tap_out(i));
    }

    // ---- single-stream blocks on the concatenated stream (Flux2SingleBlock.swift:59-98,
    // Flux2ParallelAttention.swift:72-123); with a gap the two live row ranges are two problems
    // on the same weights
    const int n_rng = img0 == s_txt ? 1 : 2;
    const int rng0[2] = {0, img0}, rngn[2] = {img0 == s_txt ? S : s_txt, n_img};
    const bool qk_fuse = qk_fuse_ok && (long)((S + 255) / 256) * ((9 * d + 255) / 256) >= 100;
    for (int i = 0; i < m->cfg.n_single; ++i) {
        if (tap_skip(m->cfg.n_double + i)) continue;
        F2_CHECK_HIP(tap_in(m->cfg.n_double + i));
        const flux2mi_dit::Sgl& w = m->sgl[i];
        F2_LAUNCH(ln_mod(mod_sgl, mod_sgl + d, mod_sgl, mod_sgl + d), "ln_mod(single)");
        if (qk_fuse && vt_gemm) {
            // linear1 as [q | k] (fused norm + RoPE) and the SwiGLU rows in one launch, V^T in a second one
            GemmProblem p[2];
            p[0] = blk(xn, d, 0, w.lin1, qkv, 3 * d, S, 2 * d, d, EPI_BF16);
            p[0].qk_qw = w.qn; p[0].qk_kw = w.kn; p[0].qk_pos = pos_tab;
            p[0].qk_cols = 2 * d; p[0].qk_row0 = 0; p[0].qk_qmul = q_mul;
            p[0].qk_l2t16 = log2f(2000.0f) / 16.0f;
            if (qk8) { p[0].qk8 = qk8_buf; p[0].qk8_ld = 2 * d; p[0].qk8_qmul = ldexpf(1.0f, w.q8_exp); p[0].qk8_kmul = ldexpf(1.0f, w.k8_exp); }
            p[1] = blk(xn, d, 0, wrows(w.lin1, 3 * d, d), nullptr, 0, S, 6 * d, d, EPI_SWIGLU);
            p[1].C2 = comb + d;
            p[1].ldc2 = 4 * d;
            if (bsa) hidden_blocks(p[1], 0, 4 * d, d);   // columns d..4d of linear2's operand
            F2_TRY(run_gemm(m, p, 2, PROF_LINEAR1, "single.linear1(qk|mlp)"));
            GemmProblem pv = vt_problem(wrows(w.lin1, 2 * d, d), 0, S);
            F2_TRY(run_gemm(m, &pv, 1, PROF_LINEAR1, "single.linear1(vt)"));
        } else
        {
            GemmProblem p[2];
            for (int r = 0; r < n_rng; ++r) {
                p[r] = blk(xn, d, rng0[r], w.lin1, qkv + (size_t)rng0[r] * 3 * d, 3 * d, rngn[r], 9 * d, d, EPI_QKV_MLP);
                p[r].n_split = 3 * d;
                p[r].C2 = comb + (size_t)rng0[r] * 4 * d + d;  // SwiGLU hidden goes to columns d..4d of comb
                p[r].ldc2 = 4 * d;
                if (bsa) hidden_blocks(p[r], rng0[r], 4 * d, d);
                if (qk_fuse) {   // per-head RMSNorm + RoPE of q | k in the GEMM epilogue (no separate pass over qkv)
                    p[r].qk_qw = w.qn; p[r].qk_kw = w.kn; p[r].qk_pos = pos_tab;
                    p[r].qk_cols = 2 * d; p[r].qk_row0 = rng0[r]; p[r].qk_qmul = q_mul;
                    p[r].qk_l2t16 = log2f(2000.0f) / 16.0f;   // theta 2000, 16 frequencies per axis (Flux2RoPE.swift:129-169)
                    if (qk8) { p[r].qk8 = qk8_buf; p[r].qk8_ld = 2 * d; p[r].qk8_qmul = ldexpf(1.0f, w.q8_exp); p[r].qk8_kmul = ldexpf(1.0f, w.k8_exp); }
                }
            }
            F2_TRY(run_gemm(m, p, n_rng, PROF_LINEAR1, "single.linear1"));
        }
        if (qk_fuse && vt_gemm) {
        } else if (qk_fuse) {
            F2_LAUNCH(launch_v_transpose_only(st, qkv, 3 * d, S, H, vt, s_pad), "v_transpose");
        } else
        F2_LAUNCH(launch_qk_norm_rope_vt(st, qkv, 3 * d, S, H, 0, w.qn, w.kn, w.qn, w.kn, cos_tab, sin_tab, vt, s_pad, q_mul, qk8_buf, w.q8_exp,
                                         w.k8_exp), "qk_norm_rope");
        F2_TRY(attend(m->cfg.n_double + i, comb, 4 * d, w.score_bound, w.q8_exp, w.k8_exp));
        // linear2's operand: nothing to do when both producers wrote block-scaled e4m3 themselves
        if (!bsa) F2_LAUNCH(quant(comb, 4 * d, 4 * d), "quantize(attn | ff hidden)");
        else if (!bsa_fused) F2_LAUNCH(launch_quantize_blocks_fp8(st, comb, 4 * d, xq2, 4 * d, xbs, bs_ld, S, 4 * d), "quantize(attn | ff hidden, blocks)");
        else if (!attn_blocks) F2_LAUNCH(quant_attn_blocks(4 * d), "quantize(attn out, blocks)");   // the hidden columns are in place already
        {
            GemmProblem p[2];
            for (int r = 0; r < n_rng; ++r) {
                p[r] = blk(comb, 4 * d, rng0[r], w.lin2, hs + (size_t)rng0[r] * d, d, rngn[r], d, 4 * d, EPI_GATE_RESID);
                p[r].vec = mod_sgl + 2 * d;
                if (bsa) with_blocks(p[r], rng0[r]);
            }
            F2_TRY(run_gemm(m, p, n_rng, PROF_LINEAR2, "single.linear2"));
        }
        F2_CHECK_HIP(tap_out(m->cfg.n_double + i));
    }

    // ---- output: drop text (and reference) tokens, AdaLayerNormContinuous, proj_out (:316-324, :444-452)
    F2_LAUNCH(launch_layernorm_modulate(st, hs + (size_t)out0 * d, xn, s_out, d, 0, mod_fin, mod_fin + d, mod_fin, mod_fin + d),
              "ln_mod(final)");
    {
        GemmProblem p = mk(xn, d, m->final_lin, d, out, 128, s_out, 128, d, EPI_F32);
        const size_t ws = gemm_splitk_ws_bytes(s_out, 128, d);
        if (ws) {
            F2_TRY(m->splitk_ws.reserve(ws));
            p.splitk_ws = m->splitk_ws.as<float>();
            p.splitk_ws_bytes = ws;
        }
        F2_TRY(run_gemm(m, &p, 1, 0, "proj_out"));
    }
    if (mode == FWD_KV_EXTRACT) cache->filled = true;
    return FLUX2MI_OK;
}

}  // namespace

int flux2mi_dit_prepare_steps(flux2mi_dit* m, const float* ts, int n, float guidance) {
    flux2mi_dit::StepCache& sc = m->step_cache;
    sc.ts.clear();
    if (!sc.on || n <= 0 || !m->finalized) return FLUX2MI_OK;
    F2_CHECK_HIP(hipSetDevice(m->ctx->device));
    hipStream_t st = m->ctx->stream;
    const int d = m->cfg.dim;
    const long ms = 256 + (long)19 * d;
    F2_TRY(sc.mods_all.reserve((size_t)n * ms * 4));
    float* base = sc.mods_all.as<float>();
    // the forward's sequence (dit_forward_impl) with V vectors per launch: emb | tmp | temb | mod_img | mod_txt | mod_sgl | mod_fin per row
    for (int k = 0; k < n; ++k) F2_LAUNCH(launch_sinusoid(st, ts[k] * 1000.0f, base + (size_t)k * ms), "sinusoid");
    for (int v0 = 0; v0 < n; v0 += 32) {
        const int V = n - v0 < 32 ? n - v0 : 32;
        float* emb = base + (size_t)v0 * ms;
        float *tmp = emb + 256, *temb = tmp + d, *mod_img = temb + d, *mod_txt = mod_img + 6 * d, *mod_sgl = mod_txt + 6 * d, *mod_fin = mod_sgl + 3 * d;
        F2_LAUNCH(launch_gemv_multi(st, m->time_in, emb, ms, tmp, ms, d, 256, V, 0, 0), "time_in.in");
        F2_LAUNCH(launch_gemv_multi(st, m->time_out, tmp, ms, temb, ms, d, d, V, 1, 0), "time_in.out");
        if (m->cfg.guidance_embeds) {   // one guidance value: its sinusoid once (the scratch of m->mods), read by every vector (stride 0)
            F2_TRY(m->mods.reserve((size_t)ms * 4));
            float* gemb = m->mods.as<float>();
            F2_LAUNCH(launch_sinusoid(st, guidance * 1000.0f, gemb), "sinusoid(g)");
            F2_LAUNCH(launch_gemv_multi(st, m->guid_in, gemb, 0, tmp, ms, d, 256, V, 0, 0), "guidance_in.in");
            F2_LAUNCH(launch_gemv_multi(st, m->guid_out, tmp, ms, temb, ms, d, d, V, 1, 1), "guidance_in.out");
        }
        F2_LAUNCH(launch_gemv_multi(st, m->mod_img, temb, ms, mod_img, ms, 6 * d, d, V, 1, 0), "mod_img");
        F2_LAUNCH(launch_gemv_multi(st, m->mod_txt, temb, ms, mod_txt, ms, 6 * d, d, V, 1, 0), "mod_txt");
        F2_LAUNCH(launch_gemv_multi(st, m->mod_single, temb, ms, mod_sgl, ms, 3 * d, d, V, 1, 0), "mod_single");
        F2_LAUNCH(launch_gemv_multi(st, m->final_mod, temb, ms, mod_fin, ms, 2 * d, d, V, 1, 0), "final_mod");
    }
    sc.ts.assign(ts, ts + n);
    sc.guidance = guidance;
    return FLUX2MI_OK;
}

void flux2mi_dit_step_cache(flux2mi_dit* m, bool on) {
    if (!m) return;
    static const bool allowed = [] {   // FLUX2MI_STEP_CACHE=0 (dev library): every step recomputes (A/B runs)
        const char* e = F2_DEV_ENV("FLUX2MI_STEP_CACHE");
        return e ? atoi(e) != 0 : true;
    }();
    m->step_cache.on = on && allowed;
    m->step_cache.ts.clear();
    m->step_cache.txt_ok = m->step_cache.rope_ok = false;
}

extern "C" int flux2mi_dit_forward(flux2mi_dit* m, const float* latents, int s_img, const void* txt_emb, int s_txt,
                                   float timestep, float guidance, const int32_t* img_ids, const int32_t* txt_ids,
                                   float* out, int s_out) {
    F2_REQUIRE(m && latents && txt_emb && img_ids && txt_ids && out, FLUX2MI_ERR_INVALID_ARG, "dit_forward: NULL argument");
    F2_WORKSPACE_USE(m->ctx->device);
    F2_REQUIRE(m->finalized, FLUX2MI_ERR_NOT_LOADED, "dit_forward: weights not finalized");
    F2_REQUIRE(s_img > 0 && s_txt > 0 && s_out > 0 && s_out <= s_img, FLUX2MI_ERR_SHAPE,
               "dit_forward: s_img %d s_txt %d s_out %d", s_img, s_txt, s_out);
    return dit_forward_impl(m, FWD_PLAIN, nullptr, latents, s_img, nullptr, 0, txt_emb, s_txt, timestep, guidance, img_ids,
                            txt_ids, nullptr, out, s_out);
}

// One block of a plain forward on a caller-supplied residual stream (teacher-forced per-block parity: tests feed the ORACLE's block
// input, so a block's own error is seen without what the blocks before it accumulated).  `layer`: double blocks 0..n_double-1, then
// the single blocks.  hs_in / hs_out f32 [s_txt + s_img, d] device, text rows first (Flux2TransformerBlock.swift:80-168 keeps the two
// streams apart, Flux2SingleBlock.swift:59-98 concatenates them [text ; image]: the same rows either way).  Everything a block takes
// from the step (modulation vectors, RoPE tables) is computed from timestep / guidance / ids exactly as flux2mi_dit_forward does.
// final_out (may be NULL): f32 [s_img, 128], the output layer (AdaLayerNormContinuous + proj_out, Flux2Transformer.swift:316-324) applied
// to hs_out's image rows — the forward's prediction had this block been its last.
// hs_in == NULL: OBSERVE — the whole forward runs on `latents` (f32 [s_img, 128], required then) as flux2mi_dit_forward would, the
// stream behind block `layer` is copied to hs_out, final_out is the forward's prediction.
extern "C" int flux2mi_dit_forward_block(flux2mi_dit* m, int layer, const float* hs_in, const float* latents, float* hs_out, int s_img,
                                         const void* txt_emb, int s_txt, float timestep, float guidance, const int32_t* img_ids,
                                         const int32_t* txt_ids, float* final_out) {
    F2_REQUIRE(m && hs_out && txt_emb && img_ids && txt_ids && (hs_in || latents), FLUX2MI_ERR_INVALID_ARG, "dit_forward_block: NULL argument");
    F2_WORKSPACE_USE(m->ctx->device);
    F2_REQUIRE(m->finalized, FLUX2MI_ERR_NOT_LOADED, "dit_forward_block: weights not finalized");
    F2_REQUIRE(layer >= 0 && layer < m->cfg.n_double + m->cfg.n_single, FLUX2MI_ERR_INVALID_ARG, "dit_forward_block: layer %d of %d", layer,
               m->cfg.n_double + m->cfg.n_single);
    F2_REQUIRE(s_img > 0 && s_txt > 0, FLUX2MI_ERR_SHAPE, "dit_forward_block: s_img %d s_txt %d", s_img, s_txt);
    F2_REQUIRE(!m->step_cache.on, FLUX2MI_ERR_UNSUPPORTED, "dit_forward_block: not inside a generation");
    F2_CHECK_HIP(hipSetDevice(m->ctx->device));
    // teacher-forced: the embedders still run (their rows are replaced by hs_in at the block) on zero latents; the prediction goes to
    // scratch unless asked for
    F2_TRY(m->ctx->scratch.reserve((size_t)s_img * 128 * 4 * 2));
    float* lat0 = m->ctx->scratch.as<float>();
    float* pred = final_out ? final_out : lat0 + (size_t)s_img * 128;
    if (!latents) F2_CHECK_HIP(hipMemsetAsync(lat0, 0, (size_t)s_img * 128 * 4, m->ctx->stream));
    m->tap.layer = layer;
    m->tap.hs_in = hs_in;
    m->tap.hs_out = hs_out;
    const int r = dit_forward_impl(m, FWD_PLAIN, nullptr, latents ? latents : lat0, s_img, nullptr, 0, txt_emb, s_txt, timestep, guidance, img_ids,
                                   txt_ids, nullptr, pred, s_img);
    m->tap = flux2mi_dit::BlockTap();
    return r;
}

// The timestep embedding and the modulation vectors the LAST forward outside a generation computed (Flux2Embeddings.swift:124-142,
// Flux2Modulation.swift:49-75): f32 [256 + 19 dim] = sinusoid(256) | MLP hidden(dim) | temb(dim) | img(6 dim: shift, scale, gate x 2) |
// txt(6 dim) | single(3 dim) | final(2 dim: shift, scale), copied to `out` (device or host) in stream order.  Parity tests only.
extern "C" int flux2mi_dit_last_modulation(flux2mi_dit* m, float* out, int64_t n_floats) {
    F2_REQUIRE(m && out, FLUX2MI_ERR_INVALID_ARG, "dit_last_modulation: NULL argument");
    F2_WORKSPACE_USE(m->ctx->device);
    const int64_t n = 256 + (int64_t)19 * m->cfg.dim;
    F2_REQUIRE(n_floats >= n && m->mods.ptr, FLUX2MI_ERR_SHAPE, "dit_last_modulation: %lld floats needed (and a forward first)", (long long)n);
    F2_CHECK_HIP(hipSetDevice(m->ctx->device));
    F2_CHECK_HIP(hipMemcpyAsync(out, m->mods.ptr, (size_t)n * 4, hipMemcpyDefault, m->ctx->stream));
    return FLUX2MI_OK;
}

// Which softmax form the attention of each block runs.  The fixed zero offset (attention64b.hip, DOFF = false) needs the block's
// QK-RMSNorm weights to bound the scores: 16.8 max|w_q| max|w_k| <= 80 in the log2 domain; blocks of a checkpoint with larger
// norm scales take the per-query data offset of the same kernel (DOFF = true: exact for any scores, +0.4-1.3 % attention time
// on the 1024^2 shapes, profiles/r02_l_attention_offset_bench.log).
extern "C" int flux2mi_dit_set_attention_softmax(flux2mi_dit* m, int mode) {
    F2_REQUIRE(m && (mode == 0 || mode == 1), FLUX2MI_ERR_INVALID_ARG, "dit_set_attention_softmax: mode 0 (auto) or 1 (online)");
    m->attn_softmax_mode = mode;
    return FLUX2MI_OK;
}

extern "C" int flux2mi_dit_attention_softmax_info(const flux2mi_dit* m, int* n_fixed_offset, int* n_online) {
    F2_REQUIRE(m && n_fixed_offset && n_online, FLUX2MI_ERR_INVALID_ARG, "dit_attention_softmax_info: NULL argument");
    F2_REQUIRE(m->finalized, FLUX2MI_ERR_NOT_LOADED, "dit_attention_softmax_info: weights not finalized");
    int fx = 0, on = 0;
    auto count = [&](float bound) { (m->attn_softmax_mode == 0 && attention_uses_fixed_offset(bound, true) ? fx : on) += 1; };
    for (const auto& b : m->dbl) count(b.score_bound);
    for (const auto& b : m->sgl) count(b.score_bound);
    *n_fixed_offset = fx;
    *n_online = on;
    return FLUX2MI_OK;
}

// ------------------------------------------------------------------ KV-cached reference conditioning
extern "C" int flux2mi_kv_cache_create(flux2mi_dit* m, int s_txt, int s_ref, flux2mi_kv_cache** out) {
    F2_REQUIRE(m && out && s_txt > 0 && s_ref > 0, FLUX2MI_ERR_INVALID_ARG, "kv_cache_create: bad argument");
    F2_WORKSPACE_USE(m->ctx->device);
    F2_CHECK_HIP(hipSetDevice(m->ctx->device));
    flux2mi_kv_cache* c = new flux2mi_kv_cache();
    c->dit = m;
    c->s_txt = s_txt;
    c->s_ref = s_ref;
    c->layer_elems = (size_t)2 * s_ref * m->cfg.dim;
    const size_t bytes = c->layer_elems * (size_t)(m->cfg.n_double + m->cfg.n_single) * 2;
    hipError_t e = f2_malloc((void**)&c->data, bytes);
    if (e != hipSuccess) {
        flux2mi_set_error("kv_cache_create: hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        delete c;
        return FLUX2MI_ERR_OOM;
    }
    *out = c;
    return FLUX2MI_OK;
}

// LayerKVCacheEntry of one block (TransformerKVCache.swift:13-25, doubleStreamEntry(at:) / singleStreamEntry(at:) :59-67): layer =
// double blocks first, then single blocks.  keys bf16 [s_ref, d] (post-RoPE, head h at columns h*128..), values_t bf16 [d, s_ref]
// (row h*128 + c = channel c of head h: the cache keeps V transposed, as the attention consumes it).  Device buffers; async.
extern "C" int flux2mi_kv_cache_entry(flux2mi_kv_cache* c, int layer, void* keys, void* values_t) {
    F2_REQUIRE(c && c->dit && layer >= 0 && layer < c->dit->cfg.n_double + c->dit->cfg.n_single, FLUX2MI_ERR_INVALID_ARG, "kv_cache_entry: bad argument");
    F2_REQUIRE(c->filled, FLUX2MI_ERR_NOT_LOADED, "kv_cache_entry: the cache has not been filled by an extraction pass");
    F2_CHECK_HIP(hipSetDevice(c->dit->ctx->device));
    const size_t bytes = c->layer_elems / 2 * 2;
    if (keys) F2_CHECK_HIP(hipMemcpyAsync(keys, c->k(layer), bytes, hipMemcpyDeviceToDevice, c->dit->ctx->stream));
    if (values_t) F2_CHECK_HIP(hipMemcpyAsync(values_t, c->v(layer), bytes, hipMemcpyDeviceToDevice, c->dit->ctx->stream));
    return FLUX2MI_OK;
}

extern "C" int flux2mi_kv_cache_destroy(flux2mi_kv_cache* c) {
    if (!c) return FLUX2MI_OK;
    (void)hipSetDevice(c->dit->ctx->device);
    (void)hipStreamSynchronize(c->dit->ctx->stream);
    if (c->data) f2_free(c->data);
    delete c;
    return FLUX2MI_OK;
}

extern "C" int flux2mi_dit_forward_kv_extract(flux2mi_dit* m, flux2mi_kv_cache* cache, const float* latents, int s_out,
                                              const float* ref_latents, int s_ref, const void* txt_emb, int s_txt,
                                              float timestep, float guidance, const int32_t* img_ids,
                                              const int32_t* txt_ids, const int32_t* ref_ids, float* out) {
    F2_REQUIRE(m && cache && latents && ref_latents && txt_emb && img_ids && txt_ids && ref_ids && out, FLUX2MI_ERR_INVALID_ARG,
               "dit_forward_kv_extract: NULL argument");
    F2_WORKSPACE_USE(m->ctx->device);
    F2_REQUIRE(m->finalized, FLUX2MI_ERR_NOT_LOADED, "dit_forward_kv_extract: weights not finalized");
    F2_REQUIRE(cache->dit == m && cache->s_txt == s_txt && cache->s_ref == s_ref && s_out > 0, FLUX2MI_ERR_SHAPE,
               "dit_forward_kv_extract: cache was created for s_txt %d s_ref %d, call has %d / %d", cache->s_txt,
               cache->s_ref, s_txt, s_ref);
    // the bf16 copy of the latents also holds the reference latents in this pass
    F2_TRY(m->lat_bf16.reserve((size_t)(s_out + s_ref) * 128 * 2));
    return dit_forward_impl(m, FWD_KV_EXTRACT, cache, latents, s_out, ref_latents, s_ref, txt_emb, s_txt, timestep, guidance,
                            img_ids, txt_ids, ref_ids, out, s_out);
}

extern "C" int flux2mi_dit_forward_kv_cached(flux2mi_dit* m, flux2mi_kv_cache* cache, const float* latents, int s_out,
                                             const void* txt_emb, int s_txt, float timestep, float guidance,
                                             const int32_t* img_ids, const int32_t* txt_ids, float* out) {
    F2_REQUIRE(m && cache && latents && txt_emb && img_ids && txt_ids && out, FLUX2MI_ERR_INVALID_ARG,
               "dit_forward_kv_cached: NULL argument");
    F2_WORKSPACE_USE(m->ctx->device);
    F2_REQUIRE(m->finalized, FLUX2MI_ERR_NOT_LOADED, "dit_forward_kv_cached: weights not finalized");
    // the reference aborts here (fatalError, Flux2Transformer.swift:488); this returns a status
    F2_REQUIRE(cache->filled, FLUX2MI_ERR_NOT_LOADED, "dit_forward_kv_cached: the cache has not been filled by an extraction pass");
    F2_REQUIRE(cache->dit == m && cache->s_txt == s_txt && s_out > 0, FLUX2MI_ERR_SHAPE,
               "dit_forward_kv_cached: cache was created for s_txt %d, call has %d", cache->s_txt, s_txt);
    return dit_forward_impl(m, FWD_KV_CACHED, cache, latents, s_out, nullptr, cache->s_ref, txt_emb, s_txt, timestep, guidance,
                            img_ids, txt_ids, nullptr, out, s_out);
}

// ------------------------------------------------------------------ standalone ops (tests, microbenchmarks)
// op entry points: give launch_gemm the context's scratch as split-K workspace where it would split (few 128x128 tiles, long K)
static int op_splitk_ws(flux2mi_ctx* ctx, GemmProblem& p) {
    const size_t need = gemm_splitk_ws_bytes(p.M, p.N, p.K);
    if (!need) return FLUX2MI_OK;
    F2_TRY(ctx->scratch.reserve(need));
    p.splitk_ws = ctx->scratch.as<float>();
    p.splitk_ws_bytes = need;
    return FLUX2MI_OK;
}

extern "C" int flux2mi_op_gemm_bf16(flux2mi_ctx* ctx, const void* A, const void* W, float* C, int M, int N, int K) {
    F2_REQUIRE(ctx && A && W && C, FLUX2MI_ERR_INVALID_ARG, "op_gemm: NULL argument");
    F2_WORKSPACE_USE(ctx->device);
    GemmProblem p = mk(reinterpret_cast<const uint16_t*>(A), K, reinterpret_cast<const uint16_t*>(W), K, C, N, M, N, K, EPI_F32);
    F2_TRY(op_splitk_ws(ctx, p));
    prof_before(ctx, PROF_DIT_GEMM);
    const int r = launch_gemm(ctx->stream, &p, 1);
    prof_after(ctx, PROF_DIT_GEMM, 2.0 * M * (double)N * K);
    F2_LAUNCH(r, "op_gemm");
    return FLUX2MI_OK;
}

// The same product rounded once to bf16 [M,N] (what the block GEMMs of the model store): the like-for-like form for the
// vendor comparison of tools/gemm_bench.py (hipBLASLt writes bf16 too) and for tests of the EPI_BF16 store path.
extern "C" int flux2mi_op_gemm_bf16_out_bf16(flux2mi_ctx* ctx, const void* A, const void* W, void* C, int M, int N, int K) {
    F2_REQUIRE(ctx && A && W && C && M > 0 && N > 0 && K > 0, FLUX2MI_ERR_INVALID_ARG, "op_gemm_bf16_out_bf16: bad argument");
    F2_WORKSPACE_USE(ctx->device);
    GemmProblem p = mk(reinterpret_cast<const uint16_t*>(A), K, reinterpret_cast<const uint16_t*>(W), K, C, N, M, N, K, EPI_BF16);
    F2_TRY(op_splitk_ws(ctx, p));
    prof_before(ctx, PROF_DIT_GEMM);
    const int r = launch_gemm(ctx->stream, &p, 1);
    prof_after(ctx, PROF_DIT_GEMM, 2.0 * M * (double)N * K);
    F2_LAUNCH(r, "op_gemm_bf16_out_bf16");
    return FLUX2MI_OK;
}

// x += gate (per column) * (A W^T): the epilogue of the proj / mlp_out / linear2 GEMMs, exposed for tests and
// microbenchmarks (tools/gate_resid_bench.py)
extern "C" int flux2mi_op_gemm_gate_resid(flux2mi_ctx* ctx, const void* A, const void* W, const float* gate, float* X,
                                          int M, int N, int K) {
    F2_REQUIRE(ctx && A && W && gate && X && M > 0 && N > 0 && K > 0, FLUX2MI_ERR_INVALID_ARG, "op_gemm_gate_resid: bad argument");
    F2_WORKSPACE_USE(ctx->device);
    GemmProblem p = mk(reinterpret_cast<const uint16_t*>(A), K, reinterpret_cast<const uint16_t*>(W), K, X, N, M, N, K, EPI_GATE_RESID);
    p.vec = gate;
    F2_TRY(op_splitk_ws(ctx, p));
    prof_before(ctx, PROF_DIT_GEMM);
    const int r = launch_gemm(ctx->stream, &p, 1);
    prof_after(ctx, PROF_DIT_GEMM, 2.0 * M * (double)N * K);
    F2_LAUNCH(r, "op_gemm_gate_resid");
    return FLUX2MI_OK;
}

extern "C" int flux2mi_op_gemm_fp8(flux2mi_ctx* ctx, const void* A, const void* W, float* C, int M, int N, int K) {
    F2_REQUIRE(ctx && A && W && C && M > 0 && N > 0 && K > 0 && K % 128 == 0, FLUX2MI_ERR_INVALID_ARG, "op_gemm_fp8: bad argument");
    F2_WORKSPACE_USE(ctx->device);
    const size_t need = (size_t)M * K + (size_t)N * K + ((size_t)M + N) * 4 + 1024;
    F2_TRY(ctx->scratch.reserve(need));
    uint8_t* aq = ctx->scratch.as<uint8_t>();
    uint8_t* wq = aq + (((size_t)M * K + 255) & ~(size_t)255);
    float* as = reinterpret_cast<float*>(wq + (((size_t)N * K + 255) & ~(size_t)255));
    float* ws = as + M;
    F2_LAUNCH(launch_quantize_rows_fp8(ctx->stream, reinterpret_cast<const uint16_t*>(A), K, aq, K, as, M, K), "quantize A");
    F2_LAUNCH(launch_quantize_rows_fp8(ctx->stream, reinterpret_cast<const uint16_t*>(W), K, wq, K, ws, N, K), "quantize W");
    GemmProblem p = mk(reinterpret_cast<const uint16_t*>(aq), K, reinterpret_cast<const uint16_t*>(wq), K, C, N, M, N, K, EPI_F32);
    p.fp8 = 1; p.a_scale = as; p.w_scale = ws;
    prof_before(ctx, PROF_DIT_GEMM);
    const int r = launch_gemm(ctx->stream, &p, 1);
    prof_after(ctx, PROF_DIT_GEMM, 2.0 * M * (double)N * K);
    F2_LAUNCH(r, "op_gemm_fp8");
    return FLUX2MI_OK;
}

// The same product with A quantised in 16-row x 128-column blocks with power-of-two scales (GemmProblem::a_bscale: what the SwiGLU and
// attention epilogues of the fp8 models emit): the block scales ride in the MFMA's scale operand.
extern "C" int flux2mi_op_gemm_fp8_blocks(flux2mi_ctx* ctx, const void* A, const void* W, float* C, int M, int N, int K) {
    F2_REQUIRE(ctx && A && W && C && M > 0 && N > 0 && K > 0 && K % 128 == 0, FLUX2MI_ERR_INVALID_ARG, "op_gemm_fp8_blocks: bad argument");
    F2_WORKSPACE_USE(ctx->device);
    const int lds = (((M + 15) / 16 + 12) + 15) & ~15;
    const size_t need = (size_t)M * K + (size_t)N * K + (size_t)N * 4 + (size_t)(K / 128) * lds + 2048;
    F2_TRY(ctx->scratch.reserve(need));
    uint8_t* aq = ctx->scratch.as<uint8_t>();
    uint8_t* wq = aq + (((size_t)M * K + 255) & ~(size_t)255);
    float* ws = reinterpret_cast<float*>(wq + (((size_t)N * K + 255) & ~(size_t)255));
    uint8_t* bs = reinterpret_cast<uint8_t*>(ws) + (((size_t)N * 4 + 255) & ~(size_t)255);
    F2_CHECK_HIP(hipMemsetAsync(bs, 0x7F, (size_t)(K / 128) * lds, ctx->stream));
    F2_LAUNCH(launch_quantize_blocks_fp8(ctx->stream, reinterpret_cast<const uint16_t*>(A), K, aq, K, bs, lds, M, K), "quantize A (blocks)");
    F2_LAUNCH(launch_quantize_rows_fp8(ctx->stream, reinterpret_cast<const uint16_t*>(W), K, wq, K, ws, N, K), "quantize W");
    GemmProblem p = mk(reinterpret_cast<const uint16_t*>(aq), K, reinterpret_cast<const uint16_t*>(wq), K, C, N, M, N, K, EPI_F32);
    p.fp8 = 1; p.a_scale = nullptr; p.w_scale = ws; p.a_bscale = bs; p.a_bscale_ld = lds;
    prof_before(ctx, PROF_DIT_GEMM);
    const int r = launch_gemm(ctx->stream, &p, 1);
    prof_after(ctx, PROF_DIT_GEMM, 2.0 * M * (double)N * K);
    F2_LAUNCH(r, "op_gemm_fp8_blocks");
    return FLUX2MI_OK;
}

extern "C" int flux2mi_op_attention(flux2mi_ctx* ctx, const void* qkv_in, void* out, int S, int heads) {
    F2_REQUIRE(ctx && qkv_in && out && S > 0 && heads > 0, FLUX2MI_ERR_INVALID_ARG, "op_attention: bad argument");
    F2_WORKSPACE_USE(ctx->device);
    const int d = heads * 128, s_pad = attn_s_pad(S);
    // scratch: a private copy of qkv is not needed (q,k are used as given); only V^T is built.
    F2_TRY(ctx->scratch.reserve((size_t)heads * 128 * s_pad * 2 + 256 * 4));
    uint16_t* vt = ctx->scratch.as<uint16_t>();
    F2_CHECK_HIP(hipMemsetAsync(vt, 0, (size_t)heads * 128 * s_pad * 2, ctx->stream));
    // V -> V^T only (reuse the transpose half of launch_qk_norm_rope_vt through a helper)
    F2_LAUNCH(launch_v_transpose_only(ctx->stream, reinterpret_cast<const uint16_t*>(qkv_in), 3 * d, S, heads, vt, s_pad),
              "v_transpose");
    const uint16_t* q = reinterpret_cast<const uint16_t*>(qkv_in);
    prof_before(ctx, PROF_ATTENTION);
    const int r = launch_attention(ctx->stream, q, q + d, 3 * d, vt, s_pad, reinterpret_cast<uint16_t*>(out), d, S, S, heads, 0.f, false,
                                   nullptr, 0);   // (not pre-scaled: the 8-wave kernels, no KV-split)
    prof_after(ctx, PROF_ATTENTION, 4.0 * (double)S * S * d);
    F2_LAUNCH(r, "op_attention");
    return FLUX2MI_OK;
}

// The form the model's blocks run (dit.cpp attention_block): q arrives multiplied by log2(e)/sqrt(128) (the norm + RoPE step folds
// it in before q's one rounding to bf16) and |q.k| <= score_bound_log2 in that domain, so the softmax may use a fixed zero offset
// (attention.hip / attention64b.hip).  score_bound_log2 <= 0 or > 80: the per-query data-offset form (exact for any scores).
extern "C" int flux2mi_op_attention_prescaled(flux2mi_ctx* ctx, const void* qkv_in, void* out, int S, int heads, float score_bound_log2) {
    F2_REQUIRE(ctx && qkv_in && out && S > 0 && heads > 0, FLUX2MI_ERR_INVALID_ARG, "op_attention_prescaled: bad argument");
    F2_WORKSPACE_USE(ctx->device);
    const int d = heads * 128, s_pad = attn_s_pad(S);
    F2_TRY(ctx->scratch.reserve((size_t)heads * 128 * s_pad * 2 + 256 * 4));
    uint16_t* vt = ctx->scratch.as<uint16_t>();
    F2_CHECK_HIP(hipMemsetAsync(vt, 0, (size_t)heads * 128 * s_pad * 2, ctx->stream));
    F2_LAUNCH(launch_v_transpose_only(ctx->stream, reinterpret_cast<const uint16_t*>(qkv_in), 3 * d, S, heads, vt, s_pad), "v_transpose");
    const uint16_t* q = reinterpret_cast<const uint16_t*>(qkv_in);
    if (ctx->n_cu > 0) F2_TRY(ctx->attn_ws.reserve(attention64b_split_ws_bytes(ctx->n_cu)));   // the blocks' form incl. the KV-split of a partial last round
    prof_before(ctx, PROF_ATTENTION);
    const int r = launch_attention(ctx->stream, q, q + d, 3 * d, vt, s_pad, reinterpret_cast<uint16_t*>(out), d, S, S, heads, score_bound_log2, true,
                                   ctx->attn_ws.as<float>(), ctx->n_cu);
    prof_after(ctx, PROF_ATTENTION, 4.0 * (double)S * S * d);
    F2_LAUNCH(r, "op_attention_prescaled");
    return FLUX2MI_OK;
}

// How the blocks' attention cuts the (head, 256-query block) items of [S queries x S keys] on this context's GPU: *pieces = the
// KV pieces per item of the partial last round (1 = whole items only), *n_cu = the compute units it plans for (tests, bench reports).
extern "C" int flux2mi_op_attention_split_info(flux2mi_ctx* ctx, int S, int heads, int* pieces, int* n_cu) {
    F2_REQUIRE(ctx && S > 0 && heads > 0, FLUX2MI_ERR_INVALID_ARG, "op_attention_split_info: bad argument");
    if (pieces) *pieces = attention64b_split_factor(S, S, heads, ctx->n_cu);
    if (n_cu) *n_cu = ctx->n_cu;
    return FLUX2MI_OK;
}

// The fp8 models' form: Q K^T on e4m3 q8 = q 2^q_exp, k8 = k 2^k_exp (device bytes [S, heads*128] each; q pre-scaled like above), V bf16
// [S, heads*128]; P, O as in the bf16 form (attention64b.hip, QK8).  score_bound_log2 as flux2mi_op_attention_prescaled.
extern "C" int flux2mi_op_attention_qk8(flux2mi_ctx* ctx, const void* q8, const void* k8, const void* v, void* out, int S, int heads, int q_exp, int k_exp,
                                        float score_bound_log2) {
    return flux2mi_op_attention_qk8_blocks(ctx, q8, k8, v, out, nullptr, nullptr, 0, S, heads, q_exp, k_exp, score_bound_log2);
}

// The same with the output as block-scaled e4m3 (what the fp8 models' proj / linear2 consume): out8 bytes [S, heads*128], bs one e8m0 byte
// per 16 queries x one head's 128 columns at bs[head * bs_ld + row / 16] (value = byte's power of two x the e4m3 value).  out8 = nullptr:
// the bf16 output `out`.
extern "C" int flux2mi_op_attention_qk8_blocks(flux2mi_ctx* ctx, const void* q8, const void* k8, const void* v, void* out, void* out8, void* bs, int bs_ld,
                                               int S, int heads, int q_exp, int k_exp, float score_bound_log2) {
    F2_REQUIRE(ctx && q8 && k8 && v && (out || (out8 && bs)) && S > 0 && heads > 0, FLUX2MI_ERR_INVALID_ARG, "op_attention_qk8: bad argument");
    F2_WORKSPACE_USE(ctx->device);
    const int d = heads * 128, s_pad = attn_s_pad(S);
    F2_TRY(ctx->scratch.reserve((size_t)heads * 128 * s_pad * 2 + 256 * 4));
    uint16_t* vt = ctx->scratch.as<uint16_t>();
    F2_CHECK_HIP(hipMemsetAsync(vt, 0, (size_t)heads * 128 * s_pad * 2, ctx->stream));
    // a V matrix of its own ([S, d], first V column 0), not the fused [q | k | v] row
    F2_LAUNCH(launch_v_transpose_only(ctx->stream, reinterpret_cast<const uint16_t*>(v), d, S, heads, vt, s_pad, /*v_col0=*/0), "v_transpose");
    if (ctx->n_cu > 0) F2_TRY(ctx->attn_ws.reserve(attention64b_split_ws_bytes(ctx->n_cu)));
    prof_before(ctx, PROF_ATTENTION);
    const int r = launch_attention64b_qk8(ctx->stream, reinterpret_cast<const uint8_t*>(q8), reinterpret_cast<const uint8_t*>(k8), d, q_exp, k_exp, vt, s_pad,
                                          reinterpret_cast<uint16_t*>(out), d, S, S, heads, !attention_uses_fixed_offset(score_bound_log2, true),
                                          reinterpret_cast<uint8_t*>(out8), d, reinterpret_cast<uint8_t*>(bs), bs_ld, ctx->n_cu, ctx->attn_ws.as<float>());
    prof_after(ctx, PROF_ATTENTION, 4.0 * (double)S * S * d);
    F2_LAUNCH(r, "op_attention_qk8");
    return FLUX2MI_OK;
}

// The KV-extraction pass's launch shape (attend(), FWD_KV_EXTRACT: Flux2Attention.swift:407-425's additive mask is a key-prefix limit):
// queries [q0, q0 + nq) against the key PREFIX [0, nk) of a V^T image that holds all S keys — real values behind key nk - 1, on an
// unaligned boundary when nk % 64 != 0.  Only out rows [q0, q0 + nq) are written.
extern "C" int flux2mi_op_attention_prescaled_range(flux2mi_ctx* ctx, const void* qkv_in, void* out, int S, int heads, float score_bound_log2,
                                                    int q0, int nq, int nk) {
    F2_REQUIRE(ctx && qkv_in && out && S > 0 && heads > 0 && q0 >= 0 && nq > 0 && q0 + nq <= S && nk > 0 && nk <= S, FLUX2MI_ERR_INVALID_ARG,
               "op_attention_prescaled_range: bad argument");
    F2_WORKSPACE_USE(ctx->device);
    const int d = heads * 128, s_pad = attn_s_pad(S);
    F2_TRY(ctx->scratch.reserve((size_t)heads * 128 * s_pad * 2 + 256 * 4));
    uint16_t* vt = ctx->scratch.as<uint16_t>();
    F2_CHECK_HIP(hipMemsetAsync(vt, 0, (size_t)heads * 128 * s_pad * 2, ctx->stream));
    F2_LAUNCH(launch_v_transpose_only(ctx->stream, reinterpret_cast<const uint16_t*>(qkv_in), 3 * d, S, heads, vt, s_pad), "v_transpose");
    const uint16_t* q = reinterpret_cast<const uint16_t*>(qkv_in);
    const int r = launch_attention(ctx->stream, q + (size_t)q0 * 3 * d, q + d, 3 * d, vt, s_pad, reinterpret_cast<uint16_t*>(out) + (size_t)q0 * d, d,
                                   nq, nk, heads, score_bound_log2, true, nullptr, 0);
    F2_LAUNCH(r, "op_attention_prescaled_range");
    return FLUX2MI_OK;
}

// ------------------------------------------------------------------ per-op parity entry points (HBM-bound kernels)
// SURVEY §8c gives LayerNorm+modulate / RMSNorm+RoPE / GroupNorm their own tolerance (<= 4e-3 vs the fp32 oracle): these
// run exactly the kernels the forward runs, on caller buffers.

// y bf16 [rows, dim] = LayerNorm(x f32 [rows, dim], eps 1e-6, no affine) * (1 + scale) + shift; rows < split take
// (shift0, scale0), the others (shift1, scale1)  (Flux2TransformerBlock.swift:56-61, Flux2Modulation.swift:96-112)
extern "C" int flux2mi_op_layernorm_modulate(flux2mi_ctx* ctx, const float* x, void* y, int rows, int dim, int split,
                                             const float* shift0, const float* scale0, const float* shift1, const float* scale1) {
    F2_REQUIRE(ctx && x && y && shift0 && scale0 && shift1 && scale1 && rows > 0 && dim > 0, FLUX2MI_ERR_INVALID_ARG,
               "op_layernorm_modulate: bad argument");
    F2_WORKSPACE_USE(ctx->device);
    F2_LAUNCH(launch_layernorm_modulate(ctx->stream, x, reinterpret_cast<uint16_t*>(y), rows, dim, split, shift0, scale0, shift1, scale1),
              "op_layernorm_modulate");
    return FLUX2MI_OK;
}

// In place on qkv bf16 [S, 3*heads*128] (q | k | v): per-head RMSNorm (eps 1e-6, bf16 weights [128]; rows < split use
// set 0, the others set 1) then the 4-axis RoPE of the position ids i32 [S,4] on q and k; q is multiplied by q_mul
// before its rounding.  The STANDALONE kernel (fp8 models, small problems; Flux2Attention.swift:137-157, Flux2RoPE.swift:129-169).
extern "C" int flux2mi_op_qk_norm_rope(flux2mi_ctx* ctx, void* qkv, int S, int heads, int split, const void* qw0, const void* kw0,
                                       const void* qw1, const void* kw1, const int32_t* ids, float q_mul) {
    F2_REQUIRE(ctx && qkv && qw0 && kw0 && qw1 && kw1 && ids && S > 0 && heads > 0, FLUX2MI_ERR_INVALID_ARG, "op_qk_norm_rope: bad argument");
    F2_WORKSPACE_USE(ctx->device);
    const int s_pad = attn_s_pad(S);
    const size_t tab = (size_t)S * 128 * 4;
    F2_TRY(ctx->scratch.reserve(2 * tab + (size_t)heads * 128 * s_pad * 2 + 256));
    float* cos_tab = ctx->scratch.as<float>();
    float* sin_tab = cos_tab + (size_t)S * 128;
    uint16_t* vt = reinterpret_cast<uint16_t*>(sin_tab + (size_t)S * 128);
    F2_LAUNCH(launch_rope_tables(ctx->stream, ids, 0, ids, S, cos_tab, sin_tab), "rope_tables");
    auto W = [](const void* p) { return reinterpret_cast<const uint16_t*>(p); };
    F2_LAUNCH(launch_qk_norm_rope_vt(ctx->stream, reinterpret_cast<uint16_t*>(qkv), 3 * heads * 128, S, heads, split, W(qw0), W(kw0), W(qw1),
                                     W(kw1), cos_tab, sin_tab, vt, s_pad, q_mul),
              "op_qk_norm_rope");
    return FLUX2MI_OK;
}

// out bf16 [S, 2*dim] = RoPE(RMSNorm(x W^T)) for the q | k rows of a fused projection weight W bf16 [2*dim, dim]: the
// FUSED form, norm + rotation in the epilogue of the 256x256 GEMM with the angles computed from the ids (bf16 models)
extern "C" int flux2mi_op_qk_proj_norm_rope(flux2mi_ctx* ctx, const void* x, const void* w, int S, int dim, const void* qw,
                                            const void* kw, const int32_t* ids, float q_mul, void* out) {
    F2_REQUIRE(ctx && x && w && qw && kw && ids && out && S > 0 && dim > 0 && dim % 256 == 0, FLUX2MI_ERR_INVALID_ARG,
               "op_qk_proj_norm_rope: bad argument (dim %% 256 == 0)");
    F2_WORKSPACE_USE(ctx->device);
    const size_t tab = (size_t)S * 128 * 4;
    F2_TRY(ctx->scratch.reserve(2 * tab + (size_t)S * 16 + 256));
    float* cos_tab = ctx->scratch.as<float>();
    float* sin_tab = cos_tab + (size_t)S * 128;
    float* pos_tab = sin_tab + (size_t)S * 128;
    F2_LAUNCH(launch_rope_tables(ctx->stream, ids, 0, ids, S, cos_tab, sin_tab, pos_tab), "rope_tables");
    GemmProblem p = mk(reinterpret_cast<const uint16_t*>(x), dim, reinterpret_cast<const uint16_t*>(w), dim, out, 2 * dim, S, 2 * dim, dim, EPI_BF16);
    p.qk_qw = reinterpret_cast<const uint16_t*>(qw);
    p.qk_kw = reinterpret_cast<const uint16_t*>(kw);
    p.qk_pos = pos_tab;
    p.qk_cols = 2 * dim;
    p.qk_row0 = 0;
    p.qk_qmul = q_mul;
    p.qk_l2t16 = log2f(2000.0f) / 16.0f;
    F2_LAUNCH(launch_gemm(ctx->stream, &p, 1), "op_qk_proj_norm_rope");
    return FLUX2MI_OK;
}

// y = [SiLU](GroupNorm(x)) on NHWC bf16 [pixels, C]: f32 statistics per (group), eps, per-channel gamma / beta f32
// (VAE/ResnetBlock.swift:9-55, :168-186)
extern "C" int flux2mi_op_groupnorm_silu(flux2mi_ctx* ctx, const void* x, void* y, long long pixels, int C, int groups, const float* gamma,
                                         const float* beta, float eps, int silu) {
    F2_REQUIRE(ctx && x && y && gamma && beta && pixels > 0 && C > 0 && groups > 0 && C % groups == 0, FLUX2MI_ERR_INVALID_ARG,
               "op_groupnorm_silu: bad argument");
    F2_WORKSPACE_USE(ctx->device);
    const size_t part = (size_t)groupnorm_partial_floats(pixels, groups) * 4;
    F2_TRY(ctx->scratch.reserve(part + (size_t)2 * C * 4 + 256));
    float* partial = ctx->scratch.as<float>();
    float* ab = reinterpret_cast<float*>(reinterpret_cast<uint8_t*>(partial) + ((part + 255) & ~(size_t)255));
    F2_LAUNCH(launch_groupnorm_stats(ctx->stream, reinterpret_cast<const uint16_t*>(x), pixels, C, groups, gamma, beta, eps, partial, ab),
              "groupnorm_stats");
    F2_LAUNCH(launch_groupnorm_apply(ctx->stream, reinterpret_cast<const uint16_t*>(x), reinterpret_cast<uint16_t*>(y), pixels, C, ab, silu),
              "groupnorm_apply");
    return FLUX2MI_OK;
}
