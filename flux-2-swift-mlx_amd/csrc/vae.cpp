// vae.cpp — AutoencoderKLFlux2.decode (and .encode, posterior mean) on MI355X (VAE/AutoencoderKL.swift:129-143,
// VAE/VAEDecoder.swift:91-121, VAE/ResnetBlock.swift).  NHWC bf16 activations, OHWI bf16
// conv weights (same layouts the reference uses with MLX), f32 bias / GroupNorm params.
//
// Decoder graph:  z -> post_quant_conv 1x1 -> conv_in 3x3 -> mid(Resnet, Attn, Resnet)
//   -> 4 up blocks x 3 Resnets (+ nearest-2x + conv3x3 after blocks 0-2)
//   -> GroupNorm -> SiLU -> conv_out 3x3 -> [3,H,W] f32 and/or u8 HWC.
// Encoder graph (VAE/VAEEncoder.swift:88-118, AutoencoderKL.swift:90-123): image [3,H,W] in [-1,1]
//   -> conv_in 3x3 (3 -> c0; the 3 input channels are zero-padded to 32 for the MFMA K-step)
//   -> 4 down blocks x 2 Resnets (+ stride-2 conv with bottom/right zero padding after blocks 0-2)
//   -> mid(Resnet, Attn, Resnet) -> GroupNorm -> SiLU -> conv_out 3x3 (-> 64) -> quant_conv 1x1
//   -> posterior MEAN = channels 0..31 (Flux2Pipeline.swift:2201 uses the mean, no scaling factor).
// Encoder tensors are optional: a checkpoint loaded without them can decode but not encode.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <unordered_map>
#include <vector>

#include <memory>
#include "common.h"
#include "kernels.h"

namespace {

enum ParamKind { PK_CONV_W, PK_LINEAR_W, PK_F32 };

struct Param {
    std::string name;
    int kind;
    int64_t shape[4];
    int ndim;
    void* ptr = nullptr;  // conv: bf16 [O_pad][KK][I_pad]; linear: bf16 [O][I]; f32 vectors
    bool got = false;
    bool encoder = false;  // encode-only tensor: not required by finalize
    int i_pad = 0;         // conv: input channels padded to this count (conv_in of the encoder)
};

struct ConvW { int w = -1, b = -1, cin = 0, cout = 0, ks = 0; int seg_w = -1, seg = 0; };   // seg_w: index into flux2mi_vae::taprow_w (cin % 64 != 0)
struct GNW { int g = -1, b = -1, c = 0; };
struct LinW { int w = -1, b = -1; };
struct ResnetW { GNW n1, n2; ConvW c1, c2, sc; bool has_sc = false; };
struct AttnW { GNW gn; LinW q, k, v, o; };

}  // namespace

struct flux2mi_vae {
    flux2mi_ctx* ctx = nullptr;
    flux2mi_vae_config cfg{};
    std::vector<Param> params;
    std::unordered_map<std::string, int> index;
    bool finalized = false;

    ConvW post_quant, conv_in, conv_out;
    ResnetW mid0, mid1;
    GNW norm_out;
    AttnW attn;
    ResnetW up[4][3];
    ConvW upsample[3];
    DevBuf up_phase[3];                 // the upsample convolutions as four 2x2 convolutions (launch_upsample_phase_weights), built by finalize
    // 3x3 convolutions whose channel count is not a multiple of 64 (the small decoder's 96): weights re-laid out per kernel ROW (3 * cin values
    // + zeros to a whole number of K-tiles) so that they run on the GEMM kernel too (conv_gemm); built by finalize, one per such convolution
    std::vector<std::unique_ptr<DevBuf>> taprow_w;
    int bn_mean = -1, bn_var = -1;
    // encoder side
    ConvW e_conv_in, e_conv_out, e_quant, e_down[3];
    ResnetW e_res[4][2], e_mid0, e_mid1;
    AttnW e_attn;
    GNW e_norm_out;
    int first_encoder_param = -1;
    DevBuf img_in;

    DevBuf stage_raw, stage_bf16;
    DevBuf act[4];                      // A, B, C, D ping-pong activations
    DevBuf aq, ak, av, avt, scores, probs;
    DevBuf gn_partial, gn_ab, zin;
    DevBuf gn_chan;                     // per-channel GroupNorm partial sums written by the GEMM convolutions' epilogues
};

namespace {

int add_param(flux2mi_vae* v, const std::string& name, int kind, std::initializer_list<int64_t> shape) {
    Param p;
    p.name = name;
    p.kind = kind;
    p.ndim = (int)shape.size();
    int i = 0;
    for (int64_t s : shape) p.shape[i++] = s;
    v->params.push_back(p);
    v->index[name] = (int)v->params.size() - 1;
    return (int)v->params.size() - 1;
}
ConvW add_conv(flux2mi_vae* v, const std::string& name, int cout, int cin, int ks) {
    ConvW c;
    c.cin = cin; c.cout = cout; c.ks = ks;
    c.w = add_param(v, name + ".weight", PK_CONV_W, {cout, cin, ks, ks});
    c.b = add_param(v, name + ".bias", PK_F32, {cout});
    return c;
}
GNW add_gn(flux2mi_vae* v, const std::string& name, int c) {
    GNW g;
    g.c = c;
    g.g = add_param(v, name + ".weight", PK_F32, {c});
    g.b = add_param(v, name + ".bias", PK_F32, {c});
    return g;
}
LinW add_lin(flux2mi_vae* v, const std::string& name, int out, int in) {
    LinW l;
    l.w = add_param(v, name + ".weight", PK_LINEAR_W, {out, in});
    l.b = add_param(v, name + ".bias", PK_F32, {out});
    return l;
}
ResnetW add_resnet(flux2mi_vae* v, const std::string& pre, int cin, int cout) {
    ResnetW r;
    r.n1 = add_gn(v, pre + "norm1", cin);
    r.c1 = add_conv(v, pre + "conv1", cout, cin, 3);
    r.n2 = add_gn(v, pre + "norm2", cout);
    r.c2 = add_conv(v, pre + "conv2", cout, cout, 3);
    r.has_sc = cin != cout;
    if (r.has_sc) r.sc = add_conv(v, pre + "conv_shortcut", cout, cin, 1);
    return r;
}

AttnW add_attn(flux2mi_vae* v, const std::string& a, int c) {
    AttnW w;
    w.gn = add_gn(v, a + "group_norm", c);
    w.q = add_lin(v, a + "to_q", c, c);
    w.k = add_lin(v, a + "to_k", c, c);
    w.v = add_lin(v, a + "to_v", c, c);
    w.o = add_lin(v, a + "to_out.0", c, c);
    return w;
}

void build(flux2mi_vae* v) {
    const flux2mi_vae_config& c = v->cfg;
    const int L = c.latent_channels, mid = c.channels[3];
    v->post_quant = add_conv(v, "post_quant_conv", L, L, 1);
    v->conv_in = add_conv(v, "decoder.conv_in", mid, L, 3);
    v->mid0 = add_resnet(v, "decoder.mid_block.resnets.0.", mid, mid);
    v->attn = add_attn(v, "decoder.mid_block.attentions.0.", mid);
    v->mid1 = add_resnet(v, "decoder.mid_block.resnets.1.", mid, mid);
    int prev = mid;
    char buf[128];
    for (int i = 0; i < 4; ++i) {
        const int co = c.channels[3 - i];  // reversed channel order (VAEDecoder.swift:49-78)
        for (int j = 0; j < c.layers_per_block + 1; ++j) {
            snprintf(buf, sizeof buf, "decoder.up_blocks.%d.resnets.%d.", i, j);
            v->up[i][j] = add_resnet(v, buf, j == 0 ? prev : co, co);
        }
        prev = co;
        if (i < 3) {
            snprintf(buf, sizeof buf, "decoder.up_blocks.%d.upsamplers.0.conv", i);
            v->upsample[i] = add_conv(v, buf, co, co, 3);
        }
    }
    v->norm_out = add_gn(v, "decoder.conv_norm_out", c.channels[0]);
    v->conv_out = add_conv(v, "decoder.conv_out", 3, c.channels[0], 3);
    v->bn_mean = add_param(v, "bn.running_mean", PK_F32, {4 * L});
    v->bn_var = add_param(v, "bn.running_var", PK_F32, {4 * L});

    // ---- encoder (diffusers keys, WeightLoader.swift:486-560)
    v->first_encoder_param = (int)v->params.size();
    v->e_conv_in = add_conv(v, "encoder.conv_in", c.channels[0], 3, 3);
    v->params[v->e_conv_in.w].i_pad = 32;
    v->e_conv_in.cin = 32;
    prev = c.channels[0];
    for (int i = 0; i < 4; ++i) {
        const int co = c.channels[i];
        for (int j = 0; j < c.layers_per_block; ++j) {
            snprintf(buf, sizeof buf, "encoder.down_blocks.%d.resnets.%d.", i, j);
            v->e_res[i][j] = add_resnet(v, buf, j == 0 ? prev : co, co);
        }
        prev = co;
        if (i < 3) {
            snprintf(buf, sizeof buf, "encoder.down_blocks.%d.downsamplers.0.conv", i);
            v->e_down[i] = add_conv(v, buf, co, co, 3);
        }
    }
    v->e_mid0 = add_resnet(v, "encoder.mid_block.resnets.0.", mid, mid);
    v->e_attn = add_attn(v, "encoder.mid_block.attentions.0.", mid);
    v->e_mid1 = add_resnet(v, "encoder.mid_block.resnets.1.", mid, mid);
    v->e_norm_out = add_gn(v, "encoder.conv_norm_out", mid);
    v->e_conv_out = add_conv(v, "encoder.conv_out", 2 * L, mid, 3);
    v->e_quant = add_conv(v, "quant_conv", 2 * L, 2 * L, 1);
    for (size_t i = (size_t)v->first_encoder_param; i < v->params.size(); ++i) v->params[i].encoder = true;
}

size_t dsize(int dtype) { return dtype == FLUX2MI_F32 ? 4 : 2; }
inline int pad32(int x) { return ((x + 31) / 32) * 32; }

}  // namespace

extern "C" int flux2mi_vae_preset(const char* name, flux2mi_vae_config* out) {
    F2_REQUIRE(name && out, FLUX2MI_ERR_INVALID_ARG, "vae_preset: NULL argument");
    // Configuration/VAEConfig.swift:44-81
    if (!strcmp(name, "standard")) *out = flux2mi_vae_config{{128, 256, 512, 512}, 32, 32, 2};
    else if (!strcmp(name, "small")) *out = flux2mi_vae_config{{96, 192, 384, 384}, 32, 32, 2};
    else {
        flux2mi_set_error("vae_preset: unknown variant '%s'", name);
        return FLUX2MI_ERR_INVALID_ARG;
    }
    return FLUX2MI_OK;
}

extern "C" int flux2mi_vae_create(flux2mi_ctx* ctx, const flux2mi_vae_config* cfg, flux2mi_vae** out) {
    F2_REQUIRE(ctx && cfg && out, FLUX2MI_ERR_INVALID_ARG, "vae_create: NULL argument");
    for (int i = 0; i < 4; ++i)
        F2_REQUIRE(cfg->channels[i] > 0 && cfg->channels[i] % 32 == 0 && cfg->channels[i] % cfg->groups == 0,
                   FLUX2MI_ERR_INVALID_ARG, "vae_create: channels[%d]=%d must be a multiple of 32 and of groups", i,
                   cfg->channels[i]);
    F2_REQUIRE(cfg->latent_channels == 32 && cfg->layers_per_block == 2 && cfg->groups > 0 && cfg->groups <= 64,
               FLUX2MI_ERR_INVALID_ARG, "vae_create: unsupported config");
    F2_CHECK_HIP(hipSetDevice(ctx->device));
    flux2mi_vae* v = new flux2mi_vae();
    v->ctx = ctx;
    v->cfg = *cfg;
    build(v);
    for (Param& p : v->params) {
        size_t bytes = 0;
        if (p.kind == PK_CONV_W) bytes = (size_t)pad32((int)p.shape[0]) * (p.i_pad ? p.i_pad : p.shape[1]) * p.shape[2] * p.shape[3] * 2;
        else if (p.kind == PK_LINEAR_W) bytes = (size_t)p.shape[0] * p.shape[1] * 2;
        else bytes = (size_t)p.shape[0] * 4;
        hipError_t e = f2_malloc(&p.ptr, bytes);
        if (e != hipSuccess) {
            flux2mi_set_error("vae_create: hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
            flux2mi_vae_destroy(v);
            return FLUX2MI_ERR_OOM;
        }
    }
    *out = v;
    return FLUX2MI_OK;
}

extern "C" int flux2mi_vae_destroy(flux2mi_vae* v) {
    if (!v) return FLUX2MI_OK;
    F2_WORKSPACE_USE(v->ctx->device);   // explicit releases below: not beside a trim
    (void)hipSetDevice(v->ctx->device);
    (void)hipStreamSynchronize(v->ctx->stream);
    for (Param& p : v->params)
        if (p.ptr) f2_free(p.ptr);
    DevBuf* bufs[] = {&v->stage_raw, &v->stage_bf16, &v->act[0], &v->act[1], &v->act[2], &v->act[3], &v->aq, &v->ak,
                      &v->av, &v->avt, &v->scores, &v->probs, &v->gn_partial, &v->gn_ab, &v->zin, &v->img_in, &v->gn_chan};
    for (DevBuf* b : bufs) b->release();
    for (DevBuf& b : v->up_phase) b.release();
    for (auto& b : v->taprow_w) b->release();
    delete v;
    return FLUX2MI_OK;
}

extern "C" int flux2mi_vae_set_tensor(flux2mi_vae* v, const char* name, int dtype, const int64_t* shape, int ndim,
                                      const void* data) {
    F2_REQUIRE(v && name && shape && data, FLUX2MI_ERR_INVALID_ARG, "vae_set_tensor: NULL argument");
    F2_WORKSPACE_USE(v->ctx->device);
    if (strstr(name, "num_batches_tracked")) return FLUX2MI_OK;  // bookkeeping tensor of the checkpoint
    auto it = v->index.find(name);
    if (it == v->index.end()) {
        flux2mi_set_error("vae_set_tensor: unknown key '%s'", name);
        return FLUX2MI_ERR_UNKNOWN_KEY;
    }
    Param& p = v->params[it->second];
    F2_REQUIRE(ndim == p.ndim, FLUX2MI_ERR_SHAPE, "vae_set_tensor(%s): ndim %d, expected %d", name, ndim, p.ndim);
    int64_t n = 1;
    for (int i = 0; i < ndim; ++i) {
        F2_REQUIRE(shape[i] == p.shape[i], FLUX2MI_ERR_SHAPE, "vae_set_tensor(%s): dim %d is %lld, expected %lld", name, i,
                   (long long)shape[i], (long long)p.shape[i]);
        n *= shape[i];
    }
    F2_CHECK_HIP(hipSetDevice(v->ctx->device));
    hipStream_t st = v->ctx->stream;
    F2_TRY(v->stage_raw.reserve((size_t)n * dsize(dtype)));
    F2_CHECK_HIP(hipMemcpyAsync(v->stage_raw.ptr, data, (size_t)n * dsize(dtype), hipMemcpyDefault, st));
    if (p.kind == PK_F32) {
        F2_LAUNCH(launch_convert_to_f32(st, v->stage_raw.ptr, dtype, reinterpret_cast<float*>(p.ptr), n), "convert_to_f32");
    } else {
        F2_TRY(v->stage_bf16.reserve((size_t)n * 2));
        F2_LAUNCH(launch_convert_to_bf16(st, v->stage_raw.ptr, dtype, v->stage_bf16.as<uint16_t>(), n), "convert_to_bf16");
        if (p.kind == PK_LINEAR_W) {
            F2_CHECK_HIP(hipMemcpyAsync(p.ptr, v->stage_bf16.ptr, (size_t)n * 2, hipMemcpyDeviceToDevice, st));
        } else {
            // OIHW (file) -> OHWI, O zero-padded to a multiple of 32 (WeightLoader.swift:495-497)
            F2_LAUNCH(launch_oihw_to_ohwi_bf16(st, v->stage_bf16.as<uint16_t>(), reinterpret_cast<uint16_t*>(p.ptr),
                                               (int)p.shape[0], (int)p.shape[1], (int)p.shape[2], (int)p.shape[3],
                                               pad32((int)p.shape[0]), p.i_pad), "oihw_to_ohwi");
        }
    }
    F2_CHECK_HIP(hipStreamSynchronize(st));
    p.got = true;
    v->finalized = false;
    return FLUX2MI_OK;
}

extern "C" int flux2mi_vae_finalize(flux2mi_vae* v) {
    F2_REQUIRE(v, FLUX2MI_ERR_INVALID_ARG, "vae_finalize: NULL");
    F2_WORKSPACE_USE(v->ctx->device);
    for (const Param& p : v->params)
        if (!p.got && !p.encoder) {
            flux2mi_set_error("vae_finalize: tensor '%s' missing", p.name.c_str());
            return FLUX2MI_ERR_NOT_LOADED;
        }
    v->stage_raw.release();
    v->stage_bf16.release();
    // nearest-2x upsample + 3x3 convolution = four 2x2 convolutions on the source image (4/9 of the MACs): combined weights, once
    for (int i = 0; i < 3; ++i) {
        const ConvW& c = v->upsample[i];
        if (c.w < 0 || c.ks != 3 || c.cin % 64 != 0) continue;
        const int cout_pad = pad32(c.cout);
        v->up_phase[i].keep = true;   // derived weights, built once: not a workspace flux2mi_ctx_trim may release
        F2_TRY(v->up_phase[i].reserve((size_t)16 * cout_pad * c.cin * 2));
        F2_LAUNCH(launch_upsample_phase_weights(v->ctx->stream, reinterpret_cast<const uint16_t*>(v->params[c.w].ptr), cout_pad, c.cin,
                                                v->up_phase[i].as<uint16_t>()), "upsample_phase_weights");
    }
    // the decoder's 3x3 convolutions with cin % 64 != 0 (small decoder, 96 channels): tap-row weights for the GEMM kernel
    {
        static const bool on = [] {
            const char* e = F2_DEV_ENV("FLUX2MI_VAE_TAPROW");
            return e ? atoi(e) != 0 : true;
        }();
        v->taprow_w.clear();
        auto prep = [&](ConvW& c) -> int {
            c.seg_w = -1;
            if (!on || c.w < 0 || c.ks != 3 || c.cin % 64 == 0 || c.cin % 32 != 0 || c.cin < 64 || v->params[c.w].i_pad) return FLUX2MI_OK;
            const int cout_pad = pad32(c.cout);
            c.seg = ((3 * c.cin + 63) / 64) * 64;
            auto buf = std::make_unique<DevBuf>();
            buf->keep = true;
            F2_TRY(buf->reserve((size_t)cout_pad * 3 * c.seg * 2));
            F2_LAUNCH(launch_conv_w_taprow(v->ctx->stream, reinterpret_cast<const uint16_t*>(v->params[c.w].ptr), cout_pad, c.cin, c.seg,
                                           buf->as<uint16_t>()), "conv_w_taprow");
            c.seg_w = (int)v->taprow_w.size();
            v->taprow_w.push_back(std::move(buf));
            return FLUX2MI_OK;
        };
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 3; ++j) {
                F2_TRY(prep(v->up[i][j].c1));
                F2_TRY(prep(v->up[i][j].c2));
            }
    }
    F2_CHECK_HIP(hipStreamSynchronize(v->ctx->stream));
    v->finalized = true;
    return FLUX2MI_OK;
}

extern "C" const float* flux2mi_vae_bn_mean(const flux2mi_vae* v) {
    return (v && v->bn_mean >= 0) ? reinterpret_cast<const float*>(v->params[v->bn_mean].ptr) : nullptr;
}
extern "C" const float* flux2mi_vae_bn_var(const flux2mi_vae* v) {
    return (v && v->bn_var >= 0) ? reinterpret_cast<const float*>(v->params[v->bn_var].ptr) : nullptr;
}

// ------------------------------------------------------------------ decode
namespace {

struct Dec {
    flux2mi_vae* v;
    hipStream_t st;
    // GroupNorm statistics that came out of a GEMM convolution's epilogue: valid for the tensor at `of` until it is rewritten
    const uint16_t* stats_of = nullptr;
    int stats_rows = 0;
    const uint16_t* W(int i) const { return reinterpret_cast<const uint16_t*>(v->params[i].ptr); }
    const float* F(int i) const { return reinterpret_cast<const float*>(v->params[i].ptr); }

    // resample: 0 none, 1 nearest-2x upsample folded into the gather, -1 stride-2 downsample
    int conv(const ConvW& c, const uint16_t* x, int Hin, int Win, int resample, const uint16_t* resid, uint16_t* y,
             float* y_f32 = nullptr, uint8_t* y_u8 = nullptr, const uint16_t* w_phase = nullptr) {
        ConvArgs a;
        memset(&a, 0, sizeof a);
        a.x = x; a.w = W(c.w); a.bias = F(c.b); a.resid = resid; a.y = y; a.y_f32_nchw = y_f32; a.y_u8_hwc = y_u8;
        a.Hin = Hin; a.Win = Win; a.Cin = c.cin; a.Cout = c.cout; a.ks = c.ks;
        a.upsample = resample > 0; a.downsample = resample < 0;
        a.w_phase = resample > 0 ? w_phase : nullptr;
        if (y == stats_of) stats_of = nullptr;
        const double ho = resample > 0 ? 2.0 * Hin : resample < 0 ? 0.5 * Hin : Hin;
        const double wo = resample > 0 ? 2.0 * Win : resample < 0 ? 0.5 * Win : Win;
        prof_before(v->ctx, PROF_VAE_CONV);
        const int r = launch_conv2d(st, a);
        prof_after(v->ctx, PROF_VAE_CONV, 2.0 * ho * wo * c.cout * c.cin * (a.w_phase ? 4 : c.ks * c.ks));
        F2_LAUNCH(r, "conv2d");
        return FLUX2MI_OK;
    }
    // the per-channel affine of GroupNorm(x): from the producing convolution's epilogue sums when x has them, else two passes
    int gn_affine(const GNW& g, const uint16_t* x, int64_t pixels) {
        F2_TRY(v->gn_ab.reserve((size_t)2 * g.c * 4));
        static const bool fuse = [] {
            const char* e = F2_DEV_ENV("FLUX2MI_VAE_GN_FUSE");
            return e ? atoi(e) != 0 : true;
        }();
        if (fuse && stats_of == x && stats_rows > 0) {
            F2_LAUNCH(launch_groupnorm_finalize_channels(st, v->gn_chan.as<float>(), stats_rows, pixels, g.c, v->cfg.groups, F(g.g), F(g.b), 1e-6f,
                                                         v->gn_ab.as<float>()), "groupnorm_finalize_channels");
            return FLUX2MI_OK;
        }
        F2_TRY(v->gn_partial.reserve((size_t)groupnorm_partial_floats(pixels, v->cfg.groups) * 4));
        F2_LAUNCH(launch_groupnorm_stats(st, x, pixels, g.c, v->cfg.groups, F(g.g), F(g.b), 1e-6f,
                                         v->gn_partial.as<float>(), v->gn_ab.as<float>()), "groupnorm_stats");
        return FLUX2MI_OK;
    }
    int gn(const GNW& g, const uint16_t* x, uint16_t* y, int64_t pixels, int silu) {
        F2_TRY(gn_affine(g, x, pixels));
        F2_LAUNCH(launch_groupnorm_apply(st, x, y, pixels, g.c, v->gn_ab.as<float>(), silu), "groupnorm_apply");
        if (y == stats_of) stats_of = nullptr;
        return FLUX2MI_OK;
    }
    // room for the epilogue sums of `problems` convolution problems of `rows` partial rows x n channels each
    int stats_room(int rows, int n, int problems, float** out) {
        F2_TRY(v->gn_chan.reserve((size_t)rows * problems * n * 2 * 4));
        *out = v->gn_chan.as<float>();
        return FLUX2MI_OK;
    }
    // ---- the heavy 3x3 convolutions as plain GEMMs on the one-wave-per-SIMD kernel (gemm256s.hip, GemmProblem::conv_*): the input is
    // a ZERO-HALOED image [(H + 2)][(W + 2)][Cin] (written by gn_padded / by the producing convolution), GEMM rows = the padded pixels
    // from the first to the last interior one, K = (ky, kx, c) = the OHWI weight rows as they are.  Taken when the 256-column tile is
    // at least 3/4 used and a K-tile is 64 whole channels; the other layers (Cout 128 / 96, conv_in, conv_out, stride 2, 1x1) stay on
    // vae_kernels.hip.  FLUX2MI_VAE_GEMM=0 (dev library) keeps everything there (A/B runs).
    bool gemm_ok(const ConvW& c, int H, int Wd) const {
        static const bool on = [] {
            const char* e = F2_DEV_ENV("FLUX2MI_VAE_GEMM");
            return e ? atoi(e) != 0 : true;
        }();
        // 256-column tiles at least 3/4 used, or <= 128 columns on the 512 x 128 tile (at least 3/4 used as well)
        const bool n_ok = c.cout >= 192 || (c.cout <= 128 && c.cout >= 96);
        // the kernel's per-lane source offsets are 32-bit: the zero-haloed operand must stay under 4 GB (2048^2 x 256 channels does; larger
        // images take the gather kernels)
        const bool off_ok = (unsigned long long)(H + 2) * (unsigned long long)(Wd + 2) * (unsigned long long)c.cin * 2ull < (1ull << 32);
        return on && c.ks == 3 && (c.cin % 64 == 0 || c.seg_w >= 0) && c.cout % 4 == 0 && n_ok && off_ok && H >= 8 && Wd >= 8 && (long)H * Wd >= 4096;
    }
    // GroupNorm(+SiLU) of plain x -> zero-haloed y
    int gn_padded(const GNW& g, const uint16_t* x, uint16_t* y, int H, int Wd, int silu) {
        F2_TRY(gn_affine(g, x, (int64_t)H * Wd));
        prof_before(v->ctx, PROF_GN_APPLY);
        const int r = launch_groupnorm_apply_padded(st, x, y, H, Wd, g.c, v->gn_ab.as<float>(), silu);
        // algorithmic bytes: the plain bf16 image read once, the zero-haloed bf16 image written once
        prof_after(v->ctx, PROF_GN_APPLY, 2.0 * g.c * ((double)H * Wd + (double)(H + 2) * (Wd + 2)));
        F2_LAUNCH(r, "groupnorm_apply_padded");
        if (y == stats_of) stats_of = nullptr;
        return FLUX2MI_OK;
    }
    // 3x3 convolution of the zero-haloed xp -> y (plain [H][W][Cout], or zero-haloed when y_padded); resid plain
    int conv_gemm(const ConvW& c, const uint16_t* xp, int H, int Wd, const uint16_t* resid, uint16_t* y, bool y_padded) {
        const int wp = Wd + 2;
        GemmProblem p;
        memset(&p, 0, sizeof p);
        if (y == stats_of) stats_of = nullptr;
        p.A = xp; p.lda = c.cin; p.W = W(c.w); p.ldw = 9 * c.cin; p.C = y; p.ldc = c.cout;
        p.M = H * wp - 2; p.N = c.cout; p.K = 9 * c.cin; p.epi = EPI_CONV; p.alpha = 1.0f;
        p.vec = F(c.b); p.resid = resid; p.ldr = c.cout;
        p.conv_seg = 3 * (c.cin / 64); p.conv_row_jump = (long long)(wp - 3) * c.cin * 2;
        if (c.cin % 64 != 0) {
            // cin % 64 != 0 (96): the K segment is a kernel ROW — its 3 * cin values are contiguous in the zero-haloed image — padded to whole
            // K-tiles; the pad columns of the weights are zero, the image bytes under them are the next pixel's first channels (finite), and
            // behind the image's last pixel 128 zero bytes (the buffer has the room: it is sized for the widest level)
            p.W = v->taprow_w[c.seg_w]->as<uint16_t>(); p.ldw = 3 * c.seg; p.K = 3 * c.seg;
            p.conv_seg = c.seg / 64; p.conv_row_jump = ((long long)wp * c.cin - c.seg) * 2;
            F2_CHECK_HIP(hipMemsetAsync(const_cast<uint16_t*>(xp) + (size_t)(H + 2) * wp * c.cin, 0, 128, st));
        }
        p.conv_wp = wp; p.conv_w = Wd; p.conv_out_padded = y_padded ? 1 : 0;
        int srows = 0;
        if (!y_padded) {   // a plain output feeds a GroupNorm next: its statistics come out of this launch's epilogue
            srows = (p.M + gemm256s_conv_tile_rows(&p, 1) - 1) / gemm256s_conv_tile_rows(&p, 1);
            F2_TRY(stats_room(srows, c.cout, 1, &p.gn_part));
        }
        if (y_padded) {   // the halo pixels in front of the first / behind the last GEMM row: rows 0 and H + 1, plus one pixel each
            const size_t edge = (size_t)(wp + 1) * c.cout * 2;
            F2_CHECK_HIP(hipMemsetAsync(y, 0, edge, st));
            F2_CHECK_HIP(hipMemsetAsync(y + ((size_t)(H + 1) * wp - 1) * c.cout, 0, edge, st));
        }
        prof_before(v->ctx, PROF_VAE_CONV);
        const int r = launch_gemm(st, &p, 1);
        prof_after(v->ctx, PROF_VAE_CONV, 2.0 * H * Wd * (double)c.cout * c.cin * 9);
        F2_LAUNCH(r, "conv2d(gemm)");
        if (srows > 0) { stats_of = y; stats_rows = srows; }
        return FLUX2MI_OK;
    }
    // nearest-2x + 3x3 convolution = four 2x2 convolutions on the zero-haloed SOURCE xp (w_phase), two phases per launch -> plain [2H][2W][Cout]
    int upsample_gemm(const ConvW& c, const uint16_t* xp, int H, int Wd, const uint16_t* w_phase, uint16_t* y) {
        const int wp = Wd + 2, cout_pad = pad32(c.cout);
        if (y == stats_of) stats_of = nullptr;
        float* part = nullptr;
        int srows = 0;
        {   // the four phase problems are the same size: two per launch, one tile height for both launches
            GemmProblem probe[2];
            memset(probe, 0, sizeof probe);
            for (int k = 0; k < 2; ++k) { probe[k].M = H * wp - 2; probe[k].N = c.cout; probe[k].K = 4 * c.cin; probe[k].epi = EPI_CONV; }
            const int tr = gemm256s_conv_tile_rows(probe, 2);
            srows = (H * wp - 2 + tr - 1) / tr;
            F2_TRY(stats_room(srows, c.cout, 4, &part));
        }
        for (int half = 0; half < 2; ++half) {
            GemmProblem p[2];
            memset(p, 0, sizeof p);
            for (int k = 0; k < 2; ++k) {
                const int ph = half * 2 + k, py = ph >> 1, px = ph & 1;
                GemmProblem& q = p[k];
                q.A = xp + ((size_t)py * wp + px) * c.cin; q.lda = c.cin;
                q.W = w_phase + (size_t)ph * cout_pad * 4 * c.cin; q.ldw = 4 * c.cin;
                q.C = y; q.ldc = c.cout; q.M = H * wp - 2; q.N = c.cout; q.K = 4 * c.cin; q.epi = EPI_CONV; q.alpha = 1.0f;
                q.vec = F(c.b); q.ldr = c.cout;
                q.conv_seg = 2 * (c.cin / 64); q.conv_row_jump = (long long)(wp - 2) * c.cin * 2;
                q.conv_wp = wp; q.conv_w = Wd; q.conv_up = 1; q.conv_py = py; q.conv_px = px;
                q.gn_part = part + (size_t)ph * srows * c.cout * 2;
            }
            prof_before(v->ctx, PROF_VAE_CONV);
            const int r = launch_gemm(st, p, 2);
            prof_after(v->ctx, PROF_VAE_CONV, 2.0 * 2 * H * Wd * (double)c.cout * c.cin * 4);
            F2_LAUNCH(r, "upsample(gemm)");
        }
        stats_of = y; stats_rows = 4 * srows;
        return FLUX2MI_OK;
    }
    // x (A) -> A in place (or, with out_padded, -> out_padded as a zero-haloed image: the operand of a following upsample_gemm);
    // scratch B, C, D
    int resnet(const ResnetW& r, uint16_t* A, uint16_t* B, uint16_t* C, uint16_t* D, int H, int Wd, uint16_t* out_padded = nullptr) {
        const int64_t px = (int64_t)H * Wd;
        if (gemm_ok(r.c1, H, Wd)) {
            F2_TRY(gn_padded(r.n1, A, B, H, Wd, 1));
            F2_TRY(conv_gemm(r.c1, B, H, Wd, nullptr, C, false));
        } else {
            F2_TRY(gn(r.n1, A, B, px, 1));
            F2_TRY(conv(r.c1, B, H, Wd, 0, nullptr, C));
        }
        const bool g2 = gemm_ok(r.c2, H, Wd);
        if (g2) F2_TRY(gn_padded(r.n2, C, B, H, Wd, 1));
        else F2_TRY(gn(r.n2, C, B, px, 1));
        const uint16_t* shortcut = A;
        if (r.has_sc) {
            F2_TRY(conv(r.sc, A, H, Wd, 0, nullptr, D));
            shortcut = D;
        }
        if (g2) F2_TRY(conv_gemm(r.c2, B, H, Wd, shortcut, out_padded ? out_padded : A, out_padded != nullptr));
        else {
            F2_REQUIRE(!out_padded, FLUX2MI_ERR_UNSUPPORTED, "vae: padded resnet output needs the GEMM convolution");
            F2_TRY(conv(r.c2, B, H, Wd, 0, shortcut, A));
        }
        return FLUX2MI_OK;
    }
    int lin(const LinW& l, const uint16_t* x, uint16_t* y, int M, int C, const uint16_t* resid) {
        if (y == stats_of) stats_of = nullptr;
        GemmProblem p;
        memset(&p, 0, sizeof p);
        p.A = x; p.lda = C; p.W = W(l.w); p.ldw = C; p.C = y; p.ldc = C; p.M = M; p.N = C; p.K = C;
        p.epi = EPI_BIAS_BF16; p.vec = F(l.b); p.resid = resid; p.ldr = C; p.alpha = 1.0f;
        F2_LAUNCH(launch_gemm(st, &p, 1), "vae.linear");
        return FLUX2MI_OK;
    }
    // single-head attention over P = H*W tokens (ResnetBlock.swift:257-313); A in place
    int attention(const AttnW& w, uint16_t* A, uint16_t* B, int H, int Wd) {
        const int C = w.gn.c;
        const int P = H * Wd;
        const size_t pc = (size_t)P * C * 2;
        static const bool flash_on = [] {
            const char* e = F2_DEV_ENV("FLUX2MI_VAE_FLASH");
            return e ? atoi(e) != 0 : true;
        }();
        if (flash_on && vae_attention_supported(C)) return attention_streaming(w, A, B, P, C);
        F2_TRY(v->aq.reserve(pc)); F2_TRY(v->ak.reserve(pc)); F2_TRY(v->av.reserve(pc)); F2_TRY(v->avt.reserve(pc));
        F2_TRY(v->scores.reserve((size_t)P * P * 4));
        F2_TRY(v->probs.reserve((size_t)P * P * 2));
        F2_TRY(gn(w.gn, A, B, P, 0));
        F2_TRY(lin(w.q, B, v->aq.as<uint16_t>(), P, C, nullptr));
        F2_TRY(lin(w.k, B, v->ak.as<uint16_t>(), P, C, nullptr));
        F2_TRY(lin(w.v, B, v->av.as<uint16_t>(), P, C, nullptr));
        prof_before(v->ctx, PROF_VAE_ATTN);
        GemmProblem s;
        memset(&s, 0, sizeof s);
        s.A = v->aq.as<uint16_t>(); s.lda = C; s.W = v->ak.as<uint16_t>(); s.ldw = C; s.C = v->scores.ptr; s.ldc = P;
        s.M = P; s.N = P; s.K = C; s.epi = EPI_F32; s.alpha = 1.0f / sqrtf((float)C);
        F2_LAUNCH(launch_gemm(st, &s, 1), "vae.scores");
        F2_LAUNCH(launch_softmax_rows(st, v->scores.as<float>(), v->probs.as<uint16_t>(), P, P), "vae.softmax");
        F2_LAUNCH(launch_transpose_bf16(st, v->av.as<uint16_t>(), v->avt.as<uint16_t>(), P, C), "vae.v_transpose");
        GemmProblem o;
        memset(&o, 0, sizeof o);
        o.A = v->probs.as<uint16_t>(); o.lda = P; o.W = v->avt.as<uint16_t>(); o.ldw = P; o.C = v->aq.ptr; o.ldc = C;
        o.M = P; o.N = C; o.K = P; o.epi = EPI_BF16; o.alpha = 1.0f;
        F2_LAUNCH(launch_gemm(st, &o, 1), "vae.pv");
        prof_after(v->ctx, PROF_VAE_ATTN, 4.0 * (double)P * (double)P * C);
        F2_TRY(lin(w.o, v->aq.as<uint16_t>(), A, P, C, A));
        return FLUX2MI_OK;
    }
    // The production widths (C = 512 standard decoder / encoder, 384 small decoder): one streaming kernel (vae_attention.hip), the P x P
    // scores never in HBM (the form above: P*P*6 bytes = 1.5 GiB at 1024^2, 25.8 GiB at 2048^2).  V^T comes straight out of a role-swapped
    // GEMM (W_v as the row operand: C = W_v x^T, [C][ldv]) — no transpose pass; its bias is added by the attention's epilogue (softmax
    // rows sum to 1).
    int attention_streaming(const AttnW& w, uint16_t* A, uint16_t* B, int P, int C) {
        const size_t pc = (size_t)P * C * 2;
        const int ldv = (P + 31) & ~31;
        F2_TRY(v->aq.reserve(pc)); F2_TRY(v->ak.reserve((size_t)ldv * C * 2)); F2_TRY(v->av.reserve(pc)); F2_TRY(v->avt.reserve((size_t)C * ldv * 2));   // K: ceil32(P) readable rows
        F2_TRY(gn(w.gn, A, B, P, 0));
        F2_TRY(lin(w.q, B, v->aq.as<uint16_t>(), P, C, nullptr));
        F2_TRY(lin(w.k, B, v->ak.as<uint16_t>(), P, C, nullptr));
        if (ldv != P) F2_CHECK_HIP(hipMemsetAsync(v->avt.ptr, 0, (size_t)C * ldv * 2, st));   // the key columns behind P: finite (they meet probability 0)
        GemmProblem t;
        memset(&t, 0, sizeof t);
        t.A = W(w.v.w); t.lda = C; t.W = B; t.ldw = C; t.C = v->avt.ptr; t.ldc = ldv;
        t.M = C; t.N = P; t.K = C; t.epi = EPI_BF16; t.alpha = 1.0f;
        F2_LAUNCH(launch_gemm(st, &t, 1), "vae.v^T");
        prof_before(v->ctx, PROF_VAE_ATTN);
        const int r = launch_vae_attention(st, v->aq.as<uint16_t>(), v->ak.as<uint16_t>(), v->avt.as<uint16_t>(), ldv, F(w.v.b),
                                           v->av.as<uint16_t>(), P, C, 1.0f / sqrtf((float)C));
        prof_after(v->ctx, PROF_VAE_ATTN, 4.0 * (double)P * (double)P * C);
        F2_LAUNCH(r, "vae.attention(streaming)");
        F2_TRY(lin(w.o, v->av.as<uint16_t>(), A, P, C, A));
        return FLUX2MI_OK;
    }
};

}  // namespace

extern "C" int flux2mi_vae_decode(flux2mi_vae* v, const float* z, int h, int w, float* out_f32, uint8_t* out_u8) {
    F2_REQUIRE(v && z && h > 0 && w > 0, FLUX2MI_ERR_INVALID_ARG, "vae_decode: bad argument");
    F2_WORKSPACE_USE(v->ctx->device);
    F2_REQUIRE(v->finalized, FLUX2MI_ERR_NOT_LOADED, "vae_decode: weights not finalized");
    F2_REQUIRE(out_f32 || out_u8, FLUX2MI_ERR_INVALID_ARG, "vae_decode: no output requested");
    F2_CHECK_HIP(hipSetDevice(v->ctx->device));
    const flux2mi_vae_config& c = v->cfg;
    // largest activation: channels[3-i] at (h<<i); the upsample conv output keeps the channel count
    size_t max_elems = 0;
    for (int i = 0; i < 4; ++i) {
        const size_t hw = ((size_t)h << i) * ((size_t)w << i);
        const size_t ch = (size_t)(i == 0 ? c.channels[3] : (c.channels[3 - i] > c.channels[4 - i] ? c.channels[3 - i] : c.channels[4 - i]));
        if (hw * ch > max_elems) max_elems = hw * ch;
    }
    // zero-haloed operands of the GEMM convolutions: (H + 2)(W + 2) pixels instead of H W
    {
        size_t padded = 0;
        for (int i = 0; i < 4; ++i) {
            const size_t hw = (((size_t)h << i) + 2) * (((size_t)w << i) + 2);
            const size_t ch = (size_t)(i == 0 ? c.channels[3] : (c.channels[3 - i] > c.channels[4 - i] ? c.channels[3 - i] : c.channels[4 - i]));
            if (hw * ch > padded) padded = hw * ch;
        }
        if (padded > max_elems) max_elems = padded;
    }
    for (int i = 0; i < 4; ++i) F2_TRY(v->act[i].reserve(max_elems * 2 + 256));   // + the zeroed pad behind a tap-row operand (conv_gemm)
    F2_TRY(v->zin.reserve((size_t)h * w * 32 * 2));
    Dec d{v, v->ctx->stream};
    uint16_t* A = v->act[0].as<uint16_t>();
    uint16_t* B = v->act[1].as<uint16_t>();
    uint16_t* C = v->act[2].as<uint16_t>();
    uint16_t* D = v->act[3].as<uint16_t>();

    F2_LAUNCH(launch_nchw_f32_to_nhwc_bf16(d.st, z, v->zin.as<uint16_t>(), 32, h, w), "nchw->nhwc");
    F2_TRY(d.conv(v->post_quant, v->zin.as<uint16_t>(), h, w, 0, nullptr, B));
    F2_TRY(d.conv(v->conv_in, B, h, w, 0, nullptr, A));
    F2_TRY(d.resnet(v->mid0, A, B, C, D, h, w));
    F2_TRY(d.attention(v->attn, A, B, h, w));
    F2_TRY(d.resnet(v->mid1, A, B, C, D, h, w));
    int H = h, W = w;
    for (int i = 0; i < 4; ++i) {
        // the upsampling convolution on the GEMM kernel reads a zero-haloed source: the block's last resnet writes it that way (into D;
        // that resnet has no shortcut convolution, D is free)
        const bool up_gemm = i < 3 && v->up_phase[i].ptr && d.gemm_ok(v->upsample[i], H, W) && d.gemm_ok(v->up[i][c.layers_per_block].c2, H, W) &&
                             !v->up[i][c.layers_per_block].has_sc;
        for (int j = 0; j < c.layers_per_block + 1; ++j) {
            F2_TRY(d.resnet(v->up[i][j], A, B, C, D, H, W, (up_gemm && j == c.layers_per_block) ? D : nullptr));
        }
        if (i < 3) {
            if (up_gemm) F2_TRY(d.upsample_gemm(v->upsample[i], D, H, W, v->up_phase[i].as<uint16_t>(), B));
            else F2_TRY(d.conv(v->upsample[i], A, H, W, 1, nullptr, B, nullptr, nullptr, v->up_phase[i].ptr ? v->up_phase[i].as<uint16_t>() : nullptr));
            std::swap(A, B);
            H *= 2;
            W *= 2;
        }
    }
    // conv_norm_out -> SiLU -> conv_out: one kernel at the production widths (the normalised tensor is never written), two otherwise
    {
        static const bool tail_fuse = [] {
            const char* e = F2_DEV_ENV("FLUX2MI_VAE_TAIL_FUSE");
            return e ? atoi(e) != 0 : true;
        }();
        const ConvW& co = v->conv_out;
        if (tail_fuse && (co.cin == 128 || co.cin == 96) && co.cout <= 3 && co.ks == 3) {
            F2_TRY(d.gn_affine(v->norm_out, A, (int64_t)H * W));
            prof_before(v->ctx, PROF_VAE_CONV);
            const int r = launch_conv_out_fused(d.st, A, v->gn_ab.as<float>(), d.W(co.w), d.F(co.b), H, W, co.cin, co.cout, out_f32, out_u8);
            prof_after(v->ctx, PROF_VAE_CONV, 2.0 * H * W * (double)co.cout * co.cin * 9);
            F2_LAUNCH(r, "conv_out(fused)");
            return FLUX2MI_OK;
        }
    }
    F2_TRY(d.gn(v->norm_out, A, B, (int64_t)H * W, 1));
    F2_TRY(d.conv(v->conv_out, B, H, W, 0, nullptr, nullptr, out_f32, out_u8));
    return FLUX2MI_OK;
}

// ------------------------------------------------------------------ encode
extern "C" int flux2mi_vae_has_encoder(const flux2mi_vae* v) {
    if (!v || v->first_encoder_param < 0) return 0;
    for (size_t i = (size_t)v->first_encoder_param; i < v->params.size(); ++i)
        if (!v->params[i].got) return 0;
    return 1;
}

extern "C" int flux2mi_vae_encode(flux2mi_vae* v, const float* image, int H, int W, float* z_mean) {
    F2_REQUIRE(v && image && z_mean, FLUX2MI_ERR_INVALID_ARG, "vae_encode: NULL argument");
    F2_WORKSPACE_USE(v->ctx->device);
    F2_REQUIRE(H > 0 && W > 0 && H % 16 == 0 && W % 16 == 0, FLUX2MI_ERR_SHAPE,
               "vae_encode: image %dx%d must be a multiple of 16 (8x VAE stride, 2x2 patchify)", H, W);
    F2_REQUIRE(v->finalized, FLUX2MI_ERR_NOT_LOADED, "vae_encode: weights not finalized");
    if (!flux2mi_vae_has_encoder(v)) {
        for (size_t i = (size_t)v->first_encoder_param; i < v->params.size(); ++i)
            if (!v->params[i].got) {
                flux2mi_set_error("vae_encode: encoder tensor '%s' was not loaded", v->params[i].name.c_str());
                break;
            }
        return FLUX2MI_ERR_NOT_LOADED;
    }
    F2_CHECK_HIP(hipSetDevice(v->ctx->device));
    const flux2mi_vae_config& c = v->cfg;
    size_t max_elems = 0;
    for (int i = 0; i < 4; ++i) {
        const size_t hw = ((size_t)H >> i) * ((size_t)W >> i);
        const size_t ch = (size_t)c.channels[i];
        if (hw * ch > max_elems) max_elems = hw * ch;
    }
    for (int i = 0; i < 4; ++i) F2_TRY(v->act[i].reserve(max_elems * 2 + 256));   // + the zeroed pad behind a tap-row operand (conv_gemm)
    F2_TRY(v->img_in.reserve((size_t)H * W * 32 * 2));
    Dec d{v, v->ctx->stream};
    uint16_t* A = v->act[0].as<uint16_t>();
    uint16_t* B = v->act[1].as<uint16_t>();
    uint16_t* C = v->act[2].as<uint16_t>();
    uint16_t* D = v->act[3].as<uint16_t>();

    F2_LAUNCH(launch_nchw_f32_to_nhwc_bf16_padded(d.st, image, v->img_in.as<uint16_t>(), 3, 32, H, W), "image->nhwc");
    F2_TRY(d.conv(v->e_conv_in, v->img_in.as<uint16_t>(), H, W, 0, nullptr, A));
    int h = H, w = W;
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < c.layers_per_block; ++j) F2_TRY(d.resnet(v->e_res[i][j], A, B, C, D, h, w));
        if (i < 3) {
            F2_TRY(d.conv(v->e_down[i], A, h, w, -1, nullptr, B));
            std::swap(A, B);
            h /= 2;
            w /= 2;
        }
    }
    F2_TRY(d.resnet(v->e_mid0, A, B, C, D, h, w));
    F2_TRY(d.attention(v->e_attn, A, B, h, w));
    F2_TRY(d.resnet(v->e_mid1, A, B, C, D, h, w));
    F2_TRY(d.gn(v->e_norm_out, A, B, (int64_t)h * w, 1));
    F2_TRY(d.conv(v->e_conv_out, B, h, w, 0, nullptr, C));           // [h, w, 64]
    F2_TRY(d.conv(v->e_quant, C, h, w, 0, nullptr, B));              // quant_conv 1x1
    F2_LAUNCH(launch_nhwc_bf16_to_nchw_f32(d.st, B, z_mean, 2 * c.latent_channels, 0, c.latent_channels, (int64_t)h * w),
              "mean->nchw");
    return FLUX2MI_OK;
}

// The mid block's attention alone (tests / microbenchmarks): q, k, v bf16 [P, C] as the projections leave them (biases included).
// Same path as the decoder: V^T by a transpose pass here (the decoder gets it straight from its value GEMM), then the streaming kernel.
extern "C" int flux2mi_op_vae_attention(flux2mi_ctx* ctx, const void* q, const void* k, const void* vv, void* out, int P, int C) {
    F2_REQUIRE(ctx && q && k && vv && out && P > 0, FLUX2MI_ERR_INVALID_ARG, "op_vae_attention: bad argument");
    F2_REQUIRE(vae_attention_supported(C), FLUX2MI_ERR_UNSUPPORTED, "op_vae_attention: head dim %d (512 and 384 are built)", C);
    F2_WORKSPACE_USE(ctx->device);
    F2_CHECK_HIP(hipSetDevice(ctx->device));
    const int ldv = (P + 31) & ~31;
    F2_TRY(ctx->scratch.reserve((size_t)C * ldv * 2));
    uint16_t* vt = ctx->scratch.as<uint16_t>();
    if (ldv != P) F2_CHECK_HIP(hipMemsetAsync(vt, 0, (size_t)C * ldv * 2, ctx->stream));
    // [P, C] -> [C][ldv]: the transpose kernel writes pitch P; a 2-D copy re-pitches it when P is not a multiple of 32
    if (ldv == P) {
        F2_LAUNCH(launch_transpose_bf16(ctx->stream, reinterpret_cast<const uint16_t*>(vv), vt, P, C), "op_vae_attention: v^T");
    } else {
        DevBuf tmp;
        F2_TRY(tmp.reserve((size_t)C * P * 2));
        F2_LAUNCH(launch_transpose_bf16(ctx->stream, reinterpret_cast<const uint16_t*>(vv), tmp.as<uint16_t>(), P, C), "op_vae_attention: v^T");
        F2_CHECK_HIP(hipMemcpy2DAsync(vt, (size_t)ldv * 2, tmp.ptr, (size_t)P * 2, (size_t)P * 2, C, hipMemcpyDeviceToDevice, ctx->stream));
        F2_CHECK_HIP(hipStreamSynchronize(ctx->stream));   // tmp is released on return
    }
    // the kernel reads ceil32(P) key rows (the pad rows' scores are masked): a caller's [P, C] buffer ends at row P
    DevBuf kpad;
    const uint16_t* kk = reinterpret_cast<const uint16_t*>(k);
    if (ldv != P) {
        F2_TRY(kpad.reserve((size_t)ldv * C * 2));
        F2_CHECK_HIP(hipMemcpyAsync(kpad.ptr, k, (size_t)P * C * 2, hipMemcpyDeviceToDevice, ctx->stream));
        kk = kpad.as<uint16_t>();
    }
    struct SyncAtExit { hipStream_t s; bool on; ~SyncAtExit() { if (on) (void)hipStreamSynchronize(s); } } sync_at_exit{ctx->stream, ldv != P};   // kpad is released on return
    prof_before(ctx, PROF_VAE_ATTN);
    const int r = launch_vae_attention(ctx->stream, reinterpret_cast<const uint16_t*>(q), kk, vt, ldv, nullptr,
                                       reinterpret_cast<uint16_t*>(out), P, C, 1.0f / sqrtf((float)C));
    prof_after(ctx, PROF_VAE_ATTN, 4.0 * (double)P * (double)P * C);
    F2_LAUNCH(r, "op_vae_attention");
    return FLUX2MI_OK;
}
