// pipeline.cpp — the hot section of Flux2Pipeline.generateWithResult
// (Pipeline/Flux2Pipeline.swift:1442-2117 T2I, :1490-1808 I2I conditioning) as a host
// driver over the C ABI: scheduler tables on the host, everything else enqueued on the
// context's stream.  Text embeddings and noise are explicit inputs (the text encoder and
// MLX's threefry RNG are outside the path; SURVEY.md §8b, §7 "Hard parts").
//
// Differences from the reference that do not change results:
//   * no per-step device sync (`eval`, :1983) unless a host callback needs one;
//   * no clearCache / eval segmentation (Apple unified-memory policy only, SURVEY §8a note 12);
//   * RoPE tables are recomputed from the ids of each call instead of cached by length.
#include <math.h>
#include <string.h>

#include <vector>

#include "common.h"
#include "kernels.h"

namespace {

// events + KV cache of one generation: released on every exit path
struct GenGuard {
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr;
    flux2mi_kv_cache* kv = nullptr;
    flux2mi_dit* cached = nullptr;   // flux2mi_dit_step_cache is on for this model
    ~GenGuard() {
        if (cached) flux2mi_dit_step_cache(cached, false);
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        if (ev2) (void)hipEventDestroy(ev2);
        if (kv) (void)flux2mi_kv_cache_destroy(kv);
    }
};

int upload_ids(hipStream_t st, DevBuf& buf, const std::vector<int32_t>& host) {
    F2_TRY(buf.reserve(host.size() * 4));
    F2_CHECK_HIP(hipMemcpyAsync(buf.ptr, host.data(), host.size() * 4, hipMemcpyHostToDevice, st));
    F2_CHECK_HIP(hipStreamSynchronize(st));  // host vector goes out of scope in the caller
    return FLUX2MI_OK;
}

}  // namespace

extern "C" int flux2mi_generate(flux2mi_ctx* ctx, flux2mi_dit* dit, flux2mi_vae* vae, flux2mi_generate_params* p) {
    F2_REQUIRE(ctx && dit && p, FLUX2MI_ERR_INVALID_ARG, "generate: NULL argument");
    F2_WORKSPACE_USE(ctx->device);
    F2_REQUIRE(p->txt_emb && p->s_txt > 0 && p->noise && p->steps > 0, FLUX2MI_ERR_INVALID_ARG,
               "generate: txt_emb/noise/steps missing");
    F2_REQUIRE(p->skip_decode || vae, FLUX2MI_ERR_NOT_LOADED, "generate: VAE required unless skip_decode");
    // effectiveStrength = clamp(strength, 0.01, 1); < 1 without initLatents is an error (Flux2Pipeline.swift:1208-1213)
    const float strength = fmaxf(0.01f, fminf(1.0f, p->strength));
    F2_REQUIRE(!(strength < 1.0f && p->init_latents == nullptr), FLUX2MI_ERR_INVALID_ARG,
               "generate: strength %.3f < 1 requires init_latents", strength);
    F2_CHECK_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;

    int H = 0, W = 0;
    flux2mi_validate_dimensions(p->height, p->width, &H, &W);  // :1191
    p->out_height = H;
    p->out_width = W;
    const int h16 = H / 16, w16 = W / 16;
    const int S_img = h16 * w16;
    const int s_ref = (p->ref_latents && p->s_ref > 0) ? p->s_ref : 0;
    const int S_all = S_img + s_ref;
    const int64_t n_lat = (int64_t)S_img * 128;
    const bool custom = p->custom_sigmas && p->n_custom_sigmas > 0;
    // useImg2ImgInit (:1209); a custom sigma schedule (LoRA scheduler override) ignores strength / initLatents (:1879-1883)
    const bool use_init = p->init_latents != nullptr && strength < 1.0f && !custom;
    const bool use_cfg = p->neg_emb != nullptr && p->s_neg > 0 && p->cfg_scale > 1.0f;

    // ---- scheduler (:1873-1907)
    std::vector<float> sigmas((size_t)p->steps + 2 + (p->n_custom_sigmas > 0 ? p->n_custom_sigmas : 0));
    int n_sig = 0;
    if (custom) {
        F2_TRY(flux2mi_set_custom_sigmas(p->custom_sigmas, p->n_custom_sigmas, sigmas.data(), (int)sigmas.size(), &n_sig));
    } else {
        F2_TRY(flux2mi_set_timesteps(p->steps, S_img, use_init ? strength : 1.0f, sigmas.data(), (int)sigmas.size(),
                                     &n_sig, nullptr));
        F2_REQUIRE(!(use_init && n_sig < 2), FLUX2MI_ERR_INVALID_ARG,
                   "generate: strength %.3f with %d steps yields zero denoising steps", strength, p->steps);
    }
    const int n_steps = n_sig - 1;
    p->effective_steps = n_steps;

    flux2mi_ctx::GenScratch& g = ctx->gen;
    F2_TRY(g.packed.reserve((size_t)n_lat * 4));
    F2_TRY(g.v.reserve((size_t)n_lat * 4));
    if (use_cfg) F2_TRY(g.v2.reserve((size_t)n_lat * 4));
    float* packed = g.packed.as<float>();
    float* v = g.v.as<float>();

    // ---- position ids (:1865, LatentUtils.swift:256-298); I2I appends the reference ids (:1502)
    {
        std::vector<int32_t> t((size_t)p->s_txt * 4), im((size_t)S_img * 4);
        F2_TRY(flux2mi_text_position_ids(p->s_txt, t.data(), p->s_txt));
        F2_TRY(flux2mi_image_position_ids(H, W, im.data(), S_img));
        F2_TRY(upload_ids(st, g.ids_txt, t));
        F2_TRY(g.ids_img.reserve((size_t)S_all * 16));
        F2_CHECK_HIP(hipMemcpyAsync(g.ids_img.ptr, im.data(), im.size() * 4, hipMemcpyHostToDevice, st));
        if (s_ref) {
            F2_REQUIRE(p->ref_ids, FLUX2MI_ERR_INVALID_ARG, "generate: ref_latents without ref_ids");
            F2_CHECK_HIP(hipMemcpyAsync(g.ids_img.as<int32_t>() + (size_t)S_img * 4, p->ref_ids, (size_t)s_ref * 16,
                                        hipMemcpyDeviceToDevice, st));
        }
        F2_CHECK_HIP(hipStreamSynchronize(st));
        if (use_cfg) {
            std::vector<int32_t> tn((size_t)p->s_neg * 4);
            F2_TRY(flux2mi_text_position_ids(p->s_neg, tn.data(), p->s_neg));
            F2_TRY(upload_ids(st, g.ids_neg, tn));
        }
    }

    // ---- noise is drawn in patchified NCHW layout and packed to [S,128] (:1448, :1860)
    F2_LAUNCH(launch_transpose_2d(st, p->noise, packed, 128, S_img), "pack noise");
    if (use_init) {  // x0 = (1-s0)*init + s0*noise (:1897-1907); shape mismatch is the caller's contract: s_img rows
        F2_LAUNCH(launch_lerp(st, packed, p->init_latents, packed, sigmas[0], n_lat), "img2img init");
    }
    // transformer input buffer [output tokens ; reference tokens]
    const float* lat_in = packed;
    if (s_ref) {
        F2_TRY(g.lat_in.reserve((size_t)S_all * 128 * 4));
        F2_CHECK_HIP(hipMemcpyAsync(g.lat_in.as<float>() + n_lat, p->ref_latents, (size_t)s_ref * 128 * 4,
                                    hipMemcpyDeviceToDevice, st));
        lat_in = g.lat_in.as<float>();
    }

    GenGuard guard;
    F2_CHECK_HIP(hipEventCreate(&guard.ev0));
    F2_CHECK_HIP(hipEventCreate(&guard.ev1));
    F2_CHECK_HIP(hipEventCreate(&guard.ev2));
    hipEvent_t ev0 = guard.ev0, ev1 = guard.ev1, ev2 = guard.ev2;
    flux2mi_kv_cache*& kv = guard.kv;
    if (p->use_kv_cache) {
        F2_REQUIRE(s_ref > 0, FLUX2MI_ERR_INVALID_ARG, "generate: use_kv_cache needs reference latents");
        F2_REQUIRE(!use_cfg, FLUX2MI_ERR_UNSUPPORTED, "generate: use_kv_cache with classical CFG (the reference's KV path has none)");
        F2_TRY(flux2mi_kv_cache_create(dit, p->s_txt, s_ref, &kv));
    }
#define GEN_TRY(expr) F2_TRY(expr)

    // the text projection and the RoPE tables do not change from step to step: the embeddings and ids live in this call's buffers
    flux2mi_dit_step_cache(dit, true);
    guard.cached = dit;
    F2_TRY(flux2mi_dit_prepare_steps(dit, sigmas.data(), n_steps, p->guidance));   // timestep MLP + modulation vectors of every step at once
    std::vector<uint8_t> ckpt_host;
    F2_CHECK_HIP(hipEventRecord(ev0, st));
    // ---- "6. Denoising Loop" (:1933-2052)
    for (int k = 0; k < n_steps; ++k) {
        const float sigma = sigmas[k], sigma_next = sigmas[k + 1];
        if (s_ref && !kv)  // per-step concat [out ; ref] (:1703)
            F2_CHECK_HIP(hipMemcpyAsync(g.lat_in.ptr, packed, (size_t)n_lat * 4, hipMemcpyDeviceToDevice, st));
        if (kv) {
            // KV-cached reference conditioning (Flux2Pipeline.swift:1555-1682): step 0 runs
            // [txt | ref | out] and stores the reference K/V of every layer, later steps run
            // [txt | out] against the stored K/V
            if (k == 0)
                GEN_TRY(flux2mi_dit_forward_kv_extract(dit, kv, packed, S_img, p->ref_latents, s_ref, p->txt_emb, p->s_txt, sigma,
                                                       p->guidance, g.ids_img.as<int32_t>(), g.ids_txt.as<int32_t>(),
                                                       g.ids_img.as<int32_t>() + (size_t)S_img * 4, v));
            else
                GEN_TRY(flux2mi_dit_forward_kv_cached(dit, kv, packed, S_img, p->txt_emb, p->s_txt, sigma, p->guidance,
                                                      g.ids_img.as<int32_t>(), g.ids_txt.as<int32_t>(), v));
        } else
        GEN_TRY(flux2mi_dit_forward(dit, lat_in, S_all, p->txt_emb, p->s_txt, sigma, p->guidance,
                                    g.ids_img.as<int32_t>(), g.ids_txt.as<int32_t>(), v, S_img));
        if (use_cfg) {  // classical CFG: second forward + v_u + g (v_c - v_u) (:1955-1973)
            GEN_TRY(flux2mi_dit_forward(dit, lat_in, S_all, p->neg_emb, p->s_neg, sigma, p->guidance,
                                        g.ids_img.as<int32_t>(), g.ids_neg.as<int32_t>(), g.v2.as<float>(), S_img));
            GEN_TRY(flux2mi_cfg_combine(ctx, v, g.v2.as<float>(), p->cfg_scale, n_lat));
        }
        // Euler step x += (sigma_next - sigma) v (:1976, Scheduler:136-156)
        GEN_TRY(flux2mi_sched_step(ctx, packed, v, sigma_next - sigma, n_lat));
        // step hook runs after the update on every step incl. the last (:1985-2001)
        if (p->on_step) {
            flux2mi_step_context sc{k, n_steps, sigma, sigma_next, H, W, s_ref ? 1 : 0};
            if (p->on_step(&sc, packed, (void*)st, p->on_step_user) != 0) {
                flux2mi_set_error("generate: cancelled by step hook at step %d", k);
                return FLUX2MI_ERR_CANCELLED;
            }
        }
        if (p->on_progress) {  // progress after the hook (:2013); the reference has synchronised here
            F2_CHECK_HIP(hipStreamSynchronize(st));
            p->on_progress(k + 1, n_steps, p->on_progress_user);
        }
        // checkpoint image every N steps (:2018-2046)
        if (p->checkpoint_interval > 0 && p->on_checkpoint && vae && (k + 1) % p->checkpoint_interval == 0) {
            F2_TRY(g.z.reserve((size_t)32 * (H / 8) * (W / 8) * 4));
            F2_TRY(g.img_u8.reserve((size_t)H * W * 3));
            GEN_TRY(flux2mi_latents_to_vae_input(ctx, packed, flux2mi_vae_bn_mean(vae), flux2mi_vae_bn_var(vae),
                                                 g.z.as<float>(), h16, w16));
            GEN_TRY(flux2mi_vae_decode(vae, g.z.as<float>(), H / 8, W / 8, nullptr, g.img_u8.as<uint8_t>()));
            ckpt_host.resize((size_t)H * W * 3);
            F2_CHECK_HIP(hipMemcpyAsync(ckpt_host.data(), g.img_u8.ptr, ckpt_host.size(), hipMemcpyDeviceToHost, st));
            F2_CHECK_HIP(hipStreamSynchronize(st));
            p->on_checkpoint(k + 1, ckpt_host.data(), H, W, p->on_checkpoint_user);
        }
    }
    F2_CHECK_HIP(hipEventRecord(ev1, st));

    if (p->out_latents)
        F2_CHECK_HIP(hipMemcpyAsync(p->out_latents, packed, (size_t)n_lat * 4, hipMemcpyDeviceToDevice, st));

    // ---- "7. VAE Decode" + "8. Post-processing" (:2058-2098)
    if (!p->skip_decode) {
        GEN_TRY(g.z.reserve((size_t)32 * (H / 8) * (W / 8) * 4));
        GEN_TRY(g.img_u8.reserve((size_t)H * W * 3));
        GEN_TRY(flux2mi_latents_to_vae_input(ctx, packed, flux2mi_vae_bn_mean(vae), flux2mi_vae_bn_var(vae),
                                             g.z.as<float>(), h16, w16));
        GEN_TRY(flux2mi_vae_decode(vae, g.z.as<float>(), H / 8, W / 8, nullptr, g.img_u8.as<uint8_t>()));
        if (p->out_image)  // the path's only D2H copy (:2448)
            F2_CHECK_HIP(hipMemcpyAsync(p->out_image, g.img_u8.ptr, (size_t)H * W * 3, hipMemcpyDeviceToHost, st));
        if (p->out_image_device)
            F2_CHECK_HIP(hipMemcpyAsync(p->out_image_device, g.img_u8.ptr, (size_t)H * W * 3, hipMemcpyDeviceToDevice, st));
    }
    F2_CHECK_HIP(hipEventRecord(ev2, st));
    F2_CHECK_HIP(hipStreamSynchronize(st));
    (void)hipEventElapsedTime(&p->ms_denoise, ev0, ev1);
    (void)hipEventElapsedTime(&p->ms_decode, ev1, ev2);
#undef GEN_TRY
    return FLUX2MI_OK;
}
