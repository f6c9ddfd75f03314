// textenc.cpp — Qwen3 text encoder -> Flux.2 Klein prompt embeddings.
//
// Replaces `Qwen3Model.forwardWithHiddenStates(_:layerIndices:attentionMask:)`
// (FluxTextEncoders/Model/Qwen3/Qwen3Model.swift:109-174) as used by
// `KleinEmbeddingExtractor.extractKleinEmbeddings` (Embeddings/KleinEmbeddingExtractor.swift:50-131):
// token ids (right-padded to S, `valid_len` real tokens) -> hidden states AFTER layers 9, 18, 27
// concatenated along the feature axis -> bf16 [S, 3*hidden] = the DiT's `txt_emb`.
// Only the layers up to the last extracted one are stored or run.
//
// Per layer (Qwen3DecoderLayer.swift, Qwen3Attention.swift, Qwen3MLP.swift):
//   x += o_proj( attn( rope( q_norm(q), k_norm(k) ), v ) ),  q|k|v = qkv_proj( RMSNorm(x) )   causal + padding mask, GQA
//   x += down_proj( silu(gate_proj(h)) * up_proj(h) ),       h = RMSNorm(x)
// HBM layout: residual stream f32 [S, hidden]; qkv bf16 [S, (hq + 2 hkv) * 128] from ONE fused GEMM
// ([q | k | v] rows); gate/up fused and row-interleaved so the GEMM epilogue applies SwiGLU; the same
// 256x256 / 128x128 GEMM kernels and the flash-attention kernel (causal instantiation) as the DiT.
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <unordered_map>
#include <vector>

#include "common.h"
#include "kernels.h"

namespace {
struct TSlot { int buf, row_off, rows, cols, kind; };   // kind 0 direct, 1 SwiGLU half (0 = gate rows, 1 = up rows)
struct TBuf { uint16_t* ptr = nullptr; int rows = 0, cols = 0, got = 0; std::string label; };
}  // namespace

struct flux2mi_textenc {
    flux2mi_ctx* ctx = nullptr;
    flux2mi_textenc_config cfg{};
    int n_run = 0;   // layers actually needed
    std::vector<TBuf> bufs;
    std::unordered_map<std::string, TSlot> slots;
    bool finalized = false;
    int64_t param_bytes = 0;
    struct Layer { uint16_t *qkv, *o, *gate_up, *down, *ln_in, *ln_post, *qn, *kn; };
    std::vector<Layer> layers;
    uint16_t* embed = nullptr;
    DevBuf stage_raw, stage_bf16, stage_gu;
    DevBuf hs, xn, qkv, vt, attn, hid, ones;
    DevBuf splitk_ws;                   // partial accumulators of the split-K projections (S = 512: few tiles, long K)
};

namespace {

int te_add_buf(flux2mi_textenc* m, int rows, int cols, const std::string& label) {
    TBuf b; b.rows = rows; b.cols = cols; b.label = label;
    m->bufs.push_back(b);
    return (int)m->bufs.size() - 1;
}
void te_slot(flux2mi_textenc* m, const std::string& name, int buf, int row_off, int rows, int cols, int kind) {
    m->slots[name] = TSlot{buf, row_off, rows, cols, kind};
    m->slots["model." + name] = TSlot{buf, row_off, rows, cols, kind};            // HF checkpoints: "model.layers..."
    m->slots["language_model.model." + name] = TSlot{buf, row_off, rows, cols, kind};
}

GemmProblem te_mk(const uint16_t* A, int lda, const uint16_t* W, int ldw, void* C, int ldc, int M, int N, int K, int epi) {
    GemmProblem p;
    memset(&p, 0, sizeof p);
    p.A = A; p.lda = lda; p.W = W; p.ldw = ldw; p.C = C; p.ldc = ldc;
    p.M = M; p.N = N; p.K = K; p.epi = epi; p.alpha = 1.0f;
    return p;
}

}  // namespace

extern "C" int flux2mi_textenc_preset(const char* name, flux2mi_textenc_config* out) {
    F2_REQUIRE(name && out, FLUX2MI_ERR_INVALID_ARG, "textenc_preset: NULL argument");
    // Qwen3-4B / Qwen3-8B as published (config.json: head_dim 128; the reference reads it at load,
    // Configuration/Qwen3Configuration.swift:52-70) with Klein's extraction layers (KleinConfig: 9, 18, 27)
    if (!strcmp(name, "klein-4b")) *out = flux2mi_textenc_config{151936, 2560, 9728, 36, 32, 8, 128, 1.0e6f, 1.0e-6f, {9, 18, 27}, 1};
    else if (!strcmp(name, "klein-9b")) *out = flux2mi_textenc_config{151936, 4096, 12288, 36, 32, 8, 128, 1.0e6f, 1.0e-6f, {9, 18, 27}, 1};
    // Dev: Mistral-Small-3.2-24B text model as published (config.json; the reference's in-code fallback
    // EncoderConfiguration.swift:86-103 has intermediate 14336 / theta 1e6 and is overridden by the file),
    // hidden states 10, 20, 30 (EmbeddingExtractor.swift:60); no q/k norm.  The position-dependent
    // "Llama-4" query scale (MistralAttention.swift:422-431) is 1 for positions < 16384, i.e. for all 512.
    else if (!strcmp(name, "dev")) *out = flux2mi_textenc_config{131072, 5120, 32768, 40, 32, 8, 128, 1.0e9f, 1.0e-5f, {10, 20, 30}, 0};
    else {
        flux2mi_set_error("textenc_preset: unknown model '%s'", name);
        return FLUX2MI_ERR_INVALID_ARG;
    }
    return FLUX2MI_OK;
}

extern "C" int flux2mi_textenc_create(flux2mi_ctx* ctx, const flux2mi_textenc_config* c, flux2mi_textenc** out) {
    F2_REQUIRE(ctx && c && out, FLUX2MI_ERR_INVALID_ARG, "textenc_create: NULL argument");
    F2_REQUIRE(c->head_dim == 128, FLUX2MI_ERR_UNSUPPORTED, "textenc_create: head_dim %d (the attention kernel is built for 128)", c->head_dim);
    F2_REQUIRE(c->vocab > 0 && c->hidden > 0 && c->hidden % 64 == 0 && c->intermediate > 0 && c->intermediate % 32 == 0 &&
                   c->n_layers > 0 && c->heads > 0 && c->kv_heads > 0 && c->heads % c->kv_heads == 0 && c->rope_theta > 1.0f,
               FLUX2MI_ERR_INVALID_ARG, "textenc_create: invalid config");
    int last = 0;
    for (int i = 0; i < 3; ++i) {
        F2_REQUIRE(c->extract_layers[i] >= 1 && c->extract_layers[i] < c->n_layers && c->extract_layers[i] > last, FLUX2MI_ERR_INVALID_ARG,
                   "textenc_create: extract_layers must be ascending in [1, n_layers) (the final-norm layer is not supported)");
        last = c->extract_layers[i];
    }
    F2_CHECK_HIP(hipSetDevice(ctx->device));
    flux2mi_textenc* m = new flux2mi_textenc();
    m->ctx = ctx;
    m->cfg = *c;
    m->n_run = last;
    const int d = c->hidden, I = c->intermediate, nq = c->heads * 128, nkv = c->kv_heads * 128;
    const int be = te_add_buf(m, c->vocab, d, "embed_tokens");
    te_slot(m, "embed_tokens.weight", be, 0, c->vocab, d, 0);
    char a[160];
    for (int l = 0; l < m->n_run; ++l) {
        auto nm = [&](const char* s) { snprintf(a, sizeof a, "layers.%d.%s", l, s); return std::string(a); };
        const int bq = te_add_buf(m, nq + 2 * nkv, d, nm("qkv"));
        te_slot(m, nm("self_attn.q_proj.weight"), bq, 0, nq, d, 0);
        te_slot(m, nm("self_attn.k_proj.weight"), bq, nq, nkv, d, 0);
        te_slot(m, nm("self_attn.v_proj.weight"), bq, nq + nkv, nkv, d, 0);
        te_slot(m, nm("self_attn.o_proj.weight"), te_add_buf(m, d, nq, nm("o_proj")), 0, d, nq, 0);
        const int bg = te_add_buf(m, 2 * I, d, nm("gate_up"));
        te_slot(m, nm("mlp.gate_proj.weight"), bg, 0, I, d, 1);
        te_slot(m, nm("mlp.up_proj.weight"), bg, 1, I, d, 1);
        te_slot(m, nm("mlp.down_proj.weight"), te_add_buf(m, d, I, nm("down_proj")), 0, d, I, 0);
        te_slot(m, nm("input_layernorm.weight"), te_add_buf(m, 1, d, nm("ln_in")), 0, 1, d, 0);
        te_slot(m, nm("post_attention_layernorm.weight"), te_add_buf(m, 1, d, nm("ln_post")), 0, 1, d, 0);
        // always allocated so that the buffer order per layer is fixed; without qk_norm they stay unused
        const int bqn = te_add_buf(m, 1, 128, nm("q_norm")), bkn = te_add_buf(m, 1, 128, nm("k_norm"));
        if (c->qk_norm) {
            te_slot(m, nm("self_attn.q_norm.weight"), bqn, 0, 1, 128, 0);
            te_slot(m, nm("self_attn.k_norm.weight"), bkn, 0, 1, 128, 0);
        } else {
            m->bufs[bqn].got = m->bufs[bkn].got = 1;
        }
    }
    for (TBuf& b : m->bufs) {
        const size_t bytes = (size_t)b.rows * b.cols * 2;
        hipError_t e = f2_malloc((void**)&b.ptr, bytes);
        if (e != hipSuccess) {
            flux2mi_set_error("textenc_create: hipMalloc(%zu) for %s failed: %s", bytes, b.label.c_str(), hipGetErrorString(e));
            flux2mi_textenc_destroy(m);
            return FLUX2MI_ERR_OOM;
        }
        m->param_bytes += (int64_t)bytes;
    }
    *out = m;
    return FLUX2MI_OK;
}

extern "C" int flux2mi_textenc_destroy(flux2mi_textenc* m) {
    if (!m) return FLUX2MI_OK;
    F2_WORKSPACE_USE(m->ctx->device);   // explicit releases below: not beside a trim
    (void)hipSetDevice(m->ctx->device);
    (void)hipStreamSynchronize(m->ctx->stream);
    for (TBuf& b : m->bufs)
        if (b.ptr) f2_free(b.ptr);
    DevBuf* ws[] = {&m->stage_raw, &m->stage_bf16, &m->stage_gu, &m->hs, &m->xn, &m->qkv, &m->vt, &m->attn, &m->hid, &m->ones, &m->splitk_ws};
    for (DevBuf* w : ws) w->release();
    delete m;
    return FLUX2MI_OK;
}

extern "C" int64_t flux2mi_textenc_param_bytes(const flux2mi_textenc* m) { return m ? m->param_bytes : 0; }

extern "C" int flux2mi_textenc_set_tensor(flux2mi_textenc* m, const char* name, int dtype, const int64_t* shape, int ndim,
                                          const void* data) {
    F2_REQUIRE(m && name && shape && data, FLUX2MI_ERR_INVALID_ARG, "textenc_set_tensor: NULL argument");
    F2_WORKSPACE_USE(m->ctx->device);
    F2_REQUIRE(dtype == FLUX2MI_F32 || dtype == FLUX2MI_BF16 || dtype == FLUX2MI_F16, FLUX2MI_ERR_INVALID_ARG,
               "textenc_set_tensor(%s): unsupported dtype %d", name, dtype);
    auto it = m->slots.find(name);
    if (it == m->slots.end()) {
        flux2mi_set_error("textenc_set_tensor: unknown key '%s'", name);
        return FLUX2MI_ERR_UNKNOWN_KEY;   // layers past the last extracted one, lm_head, final norm: not part of this path
    }
    const TSlot& s = it->second;
    int64_t rows = 1, cols = 1;
    if (ndim == 1) cols = shape[0];
    else if (ndim == 2) { rows = shape[0]; cols = shape[1]; }
    else { flux2mi_set_error("textenc_set_tensor(%s): ndim %d", name, ndim); return FLUX2MI_ERR_SHAPE; }
    F2_REQUIRE(rows == s.rows && cols == s.cols, FLUX2MI_ERR_SHAPE, "textenc_set_tensor(%s): shape [%lld,%lld], expected [%d,%d]",
               name, (long long)rows, (long long)cols, s.rows, s.cols);
    F2_CHECK_HIP(hipSetDevice(m->ctx->device));
    hipStream_t st = m->ctx->stream;
    const int64_t n = rows * cols;
    const size_t esz = dtype == FLUX2MI_F32 ? 4 : 2;
    TBuf& b = m->bufs[s.buf];
    F2_TRY(m->stage_raw.reserve((size_t)n * esz));
    F2_CHECK_HIP(hipMemcpyAsync(m->stage_raw.ptr, data, (size_t)n * esz, hipMemcpyDefault, st));
    const uint16_t* src = m->stage_raw.as<uint16_t>();
    if (dtype != FLUX2MI_BF16) {
        F2_TRY(m->stage_bf16.reserve((size_t)n * 2));
        F2_LAUNCH(launch_convert_to_bf16(st, m->stage_raw.ptr, dtype, m->stage_bf16.as<uint16_t>(), n), "convert_to_bf16");
        src = m->stage_bf16.as<uint16_t>();
    }
    if (s.kind == 0) {
        F2_CHECK_HIP(hipMemcpyAsync(b.ptr + (size_t)s.row_off * cols, src, (size_t)n * 2, hipMemcpyDeviceToDevice, st));
    } else {
        // gate (half 0) or up (half 1) of the fused SwiGLU weight: block g of 32 rows goes to rows
        // 64 g + 32 half .. + 32 (the interleave the GEMM's SwiGLU epilogue expects, gemm.hip)
        const int half = s.row_off;
        F2_CHECK_HIP(hipMemcpy2DAsync(b.ptr + (size_t)half * 32 * cols, (size_t)64 * cols * 2, src, (size_t)32 * cols * 2,
                                      (size_t)32 * cols * 2, (size_t)(rows / 32), hipMemcpyDeviceToDevice, st));
    }
    F2_CHECK_HIP(hipStreamSynchronize(st));
    b.got += (int)rows;
    m->finalized = false;
    return FLUX2MI_OK;
}

extern "C" int flux2mi_textenc_finalize(flux2mi_textenc* m) {
    F2_REQUIRE(m, FLUX2MI_ERR_INVALID_ARG, "textenc_finalize: NULL");
    F2_WORKSPACE_USE(m->ctx->device);
    for (const TBuf& b : m->bufs)
        if (b.got < b.rows) {
            flux2mi_set_error("textenc_finalize: tensor '%s' missing (%d of %d rows loaded)", b.label.c_str(), b.got, b.rows);
            return FLUX2MI_ERR_NOT_LOADED;
        }
    m->embed = m->bufs[0].ptr;
    m->layers.resize(m->n_run);
    for (int l = 0; l < m->n_run; ++l) {
        const int b0 = 1 + l * 8;   // order of te_add_buf calls in create
        m->layers[l] = {m->bufs[b0].ptr, m->bufs[b0 + 1].ptr, m->bufs[b0 + 2].ptr, m->bufs[b0 + 3].ptr,
                        m->bufs[b0 + 4].ptr, m->bufs[b0 + 5].ptr, m->bufs[b0 + 6].ptr, m->bufs[b0 + 7].ptr};
    }
    m->stage_raw.release();
    m->stage_bf16.release();
    m->finalized = true;
    return FLUX2MI_OK;
}

extern "C" int flux2mi_textenc_forward(flux2mi_textenc* m, const int32_t* ids, int S, int pad_left, int valid_len, void* out) {
    F2_REQUIRE(m && ids && out, FLUX2MI_ERR_INVALID_ARG, "textenc_forward: NULL argument");
    F2_WORKSPACE_USE(m->ctx->device);
    F2_REQUIRE(m->finalized, FLUX2MI_ERR_NOT_LOADED, "textenc_forward: weights not finalized");
    F2_REQUIRE(S > 0 && pad_left >= 0 && valid_len >= 1 && pad_left + valid_len <= S, FLUX2MI_ERR_SHAPE,
               "textenc_forward: S %d pad_left %d valid_len %d", S, pad_left, valid_len);
    F2_CHECK_HIP(hipSetDevice(m->ctx->device));
    hipStream_t st = m->ctx->stream;
    const flux2mi_textenc_config& c = m->cfg;
    const int d = c.hidden, I = c.intermediate, hq = c.heads, hkv = c.kv_heads;
    const int nq = hq * 128, nqkv = (hq + 2 * hkv) * 128, s_pad = attn_s_pad(S);
    const size_t Sz = (size_t)S;
    F2_TRY(m->hs.reserve(Sz * d * 4));
    F2_TRY(m->xn.reserve(Sz * d * 2));
    F2_TRY(m->qkv.reserve(Sz * nqkv * 2));
    F2_TRY(m->attn.reserve(Sz * nq * 2));
    F2_TRY(m->hid.reserve(Sz * I * 2));
    const size_t vt_bytes = (size_t)hkv * 128 * s_pad * 2;
    if (m->vt.bytes < vt_bytes) {
        F2_TRY(m->vt.reserve(vt_bytes));
        F2_CHECK_HIP(hipMemsetAsync(m->vt.ptr, 0, m->vt.bytes, st));
    }
    if (m->ones.bytes < (size_t)d * 4) {   // gate vector of the `x += 1 * y` epilogue
        F2_TRY(m->ones.reserve((size_t)d * 4));
        std::vector<float> one((size_t)d, 1.0f);
        F2_CHECK_HIP(hipMemcpyAsync(m->ones.ptr, one.data(), (size_t)d * 4, hipMemcpyHostToDevice, st));
        F2_CHECK_HIP(hipStreamSynchronize(st));
    }
    // the qkv / o / down projections have M = S rows: at 512 tokens their 128x128 tiles fill a fifth to a third of the chip, so
    // launch_gemm splits their K (deterministic, kernels.h GemmProblem::splitk_ws) when given a workspace
    const size_t ws_bytes = std::max(gemm_splitk_ws_bytes(S, nqkv, d), std::max(gemm_splitk_ws_bytes(S, d, nq), gemm_splitk_ws_bytes(S, d, I)));
    if (ws_bytes) F2_TRY(m->splitk_ws.reserve(ws_bytes));
    auto with_ws = [&](GemmProblem& p) {
        p.splitk_ws = ws_bytes ? m->splitk_ws.as<float>() : nullptr;
        p.splitk_ws_bytes = ws_bytes;
    };
    float* hs = m->hs.as<float>();
    uint16_t* xn = m->xn.as<uint16_t>();
    uint16_t* qkv = m->qkv.as<uint16_t>();
    uint16_t* vt = m->vt.as<uint16_t>();
    uint16_t* attn = m->attn.as<uint16_t>();
    uint16_t* hid = m->hid.as<uint16_t>();
    uint16_t* o16 = reinterpret_cast<uint16_t*>(out);

    F2_LAUNCH(launch_embed_rows(st, ids, m->embed, hs, S, d, c.vocab), "embed_tokens");
    int next = 0;
    for (int l = 0; l < m->n_run; ++l) {
        const flux2mi_textenc::Layer& w = m->layers[l];
        F2_LAUNCH(launch_rmsnorm_rows(st, hs, w.ln_in, xn, S, d, c.rms_eps), "input_layernorm");
        {
            GemmProblem p = te_mk(xn, d, w.qkv, d, qkv, nqkv, S, nqkv, d, EPI_BF16);
            with_ws(p);
            F2_LAUNCH(launch_gemm(st, &p, 1), "qkv_proj");
        }
        F2_LAUNCH(launch_qk_norm_rope_llm_vt(st, qkv, S, hq, hkv, c.qk_norm ? w.qn : nullptr, c.qk_norm ? w.kn : nullptr,
                                             c.rms_eps, c.rope_theta, vt, s_pad), "qk_norm_rope");
        F2_LAUNCH(launch_attention_causal(st, qkv, qkv + nq, nqkv, vt, s_pad, attn, nq, S, pad_left, pad_left + valid_len, hq, hkv),
                  "attention");
        {
            GemmProblem p = te_mk(attn, nq, w.o, nq, hs, d, S, d, nq, EPI_GATE_RESID);
            p.vec = m->ones.as<float>();
            with_ws(p);
            F2_LAUNCH(launch_gemm(st, &p, 1), "o_proj");
        }
        F2_LAUNCH(launch_rmsnorm_rows(st, hs, w.ln_post, xn, S, d, c.rms_eps), "post_attention_layernorm");
        {
            GemmProblem p = te_mk(xn, d, w.gate_up, d, nullptr, 0, S, 2 * I, d, EPI_SWIGLU);
            p.C2 = hid; p.ldc2 = I;
            F2_LAUNCH(launch_gemm(st, &p, 1), "gate_up_proj");
        }
        {
            GemmProblem p = te_mk(hid, I, w.down, I, hs, d, S, d, I, EPI_GATE_RESID);
            p.vec = m->ones.as<float>();
            with_ws(p);
            F2_LAUNCH(launch_gemm(st, &p, 1), "down_proj");
        }
        if (next < 3 && c.extract_layers[next] == l + 1) {   // 1-based: "after layer l + 1" (Qwen3Model.swift:147-165)
            F2_LAUNCH(launch_f32_rows_to_bf16_cols(st, hs, o16, S, d, 3 * d, next * d), "extract hidden state");
            ++next;
        }
    }
    return FLUX2MI_OK;
}
