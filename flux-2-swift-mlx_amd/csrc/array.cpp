// array.cpp — the minimal MLXArray façade of the boundary (SURVEY §8b row "MLXArray in the public API").
//
// `Flux2StepHook = (Flux2StepContext, MLXArray) -> MLXArray` (Pipeline/Flux2Pipeline.swift:64) hands MLX arrays to
// caller code that Flux2Core does not own: the chains build their RePaint blend from `MLXArray(Float)`, `+ - *` with
// broadcasting ([] x [1,S,128], [1,S,1] x [1,S,128]), `.shape`, `eval`, `MLXRandom.seed/normal`
// (Flux2Chains/Flux2MaskedInpaintingChain.swift:380-403).  For such closures to run UNCHANGED on top of this library
// the Swift shim's `MLXArray` is a handle to one of these: an f32 device tensor of <= 4 dims on the context's stream.
// Every op is eager (enqueued, not synchronised); `eval` is flux2mi_ctx_sync.  Values are immutable like MLX's: ops
// return new arrays, except that flux2mi_array_wrap gives a view over caller memory (the hook's latent buffer).
#include <stdint.h>
#include <string.h>

#include "common.h"
#include "kernels.h"

struct flux2mi_array {
    flux2mi_ctx* ctx = nullptr;
    float* data = nullptr;
    bool owned = false;
    int ndim = 0;
    int64_t shape[4] = {1, 1, 1, 1};
    int64_t numel() const {
        int64_t n = 1;
        for (int i = 0; i < ndim; ++i) n *= shape[i];
        return n;
    }
};

namespace {

int make(flux2mi_ctx* ctx, const int64_t* shape, int ndim, bool alloc, flux2mi_array** out) {
    F2_REQUIRE(ctx && out && ndim >= 0 && ndim <= 4 && (ndim == 0 || shape), FLUX2MI_ERR_INVALID_ARG, "array: bad argument (ndim %d)", ndim);
    flux2mi_array* a = new flux2mi_array();
    a->ctx = ctx;
    a->ndim = ndim;
    for (int i = 0; i < ndim; ++i) {
        if (shape[i] <= 0 || shape[i] > (int64_t)INT32_MAX) { /* binary() narrows dims to int */ delete a; flux2mi_set_error("array: dim %d is %lld", i, (long long)shape[i]); return FLUX2MI_ERR_SHAPE; }
        a->shape[i] = shape[i];
    }
    if (a->numel() > (int64_t)1 << 40) { delete a; flux2mi_set_error("array: too many elements"); return FLUX2MI_ERR_SHAPE; }
    if (alloc) {
        if (hipSetDevice(ctx->device) != hipSuccess || f2_malloc((void**)&a->data, (size_t)a->numel() * 4) != hipSuccess) {
            flux2mi_set_error("array: hipMalloc(%lld B) failed", (long long)a->numel() * 4);
            delete a;
            return FLUX2MI_ERR_OOM;
        }
        a->owned = true;
    }
    *out = a;
    return FLUX2MI_OK;
}

// numpy / MLX broadcasting: align trailing dims; each pair must match or one of them be 1
int binary(flux2mi_ctx* ctx, int op, const flux2mi_array* a, const flux2mi_array* b, flux2mi_array** out) {
    F2_REQUIRE(ctx && a && b && out, FLUX2MI_ERR_INVALID_ARG, "array op: NULL argument");
    F2_REQUIRE(a->ctx == ctx && b->ctx == ctx, FLUX2MI_ERR_INVALID_ARG, "array op: operands belong to another context");
    const int nd = a->ndim > b->ndim ? a->ndim : b->ndim;
    int64_t oshape[4] = {1, 1, 1, 1}, sa[4] = {0, 0, 0, 0}, sb[4] = {0, 0, 0, 0};
    int dims[4] = {1, 1, 1, 1};
    int64_t ra = 1, rb = 1;   // running contiguous strides, last dim first
    for (int k = 0; k < nd; ++k) {   // k-th dim from the end
        const int64_t da = k < a->ndim ? a->shape[a->ndim - 1 - k] : 1;
        const int64_t db = k < b->ndim ? b->shape[b->ndim - 1 - k] : 1;
        F2_REQUIRE(da == db || da == 1 || db == 1, FLUX2MI_ERR_SHAPE, "array op: shapes do not broadcast (dim -%d: %lld vs %lld)", k + 1,
                   (long long)da, (long long)db);
        const int64_t d = da > db ? da : db;
        oshape[nd - 1 - k] = d;
        dims[3 - k] = (int)d;
        sa[3 - k] = da == 1 ? 0 : ra;
        sb[3 - k] = db == 1 ? 0 : rb;
        ra *= da;
        rb *= db;
    }
    flux2mi_array* o = nullptr;
    F2_TRY(make(ctx, oshape, nd, true, &o));
    const int r = launch_bcast_binary(ctx->stream, op, a->data, sa, b->data, sb, o->data, dims);
    if (r != 0) {
        flux2mi_array_free(o);
        F2_LAUNCH(r, "array op");
    }
    *out = o;
    return FLUX2MI_OK;
}

}  // namespace

extern "C" int flux2mi_array_alloc(flux2mi_ctx* ctx, const int64_t* shape, int ndim, flux2mi_array** out) {
    return make(ctx, shape, ndim, true, out);
}

extern "C" int flux2mi_array_from_host(flux2mi_ctx* ctx, const float* host, const int64_t* shape, int ndim, flux2mi_array** out) {
    F2_REQUIRE(host, FLUX2MI_ERR_INVALID_ARG, "array_from_host: NULL data");
    flux2mi_array* a = nullptr;
    F2_TRY(make(ctx, shape, ndim, true, &a));
    // pageable host memory: the copy has left `host` when this returns
    if (hipMemcpyAsync(a->data, host, (size_t)a->numel() * 4, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) {
        flux2mi_array_free(a);
        flux2mi_set_error("array_from_host: copy failed");
        return FLUX2MI_ERR_HIP;
    }
    *out = a;
    return FLUX2MI_OK;
}

extern "C" int flux2mi_array_from_scalar(flux2mi_ctx* ctx, float value, flux2mi_array** out) {
    return flux2mi_array_from_host(ctx, &value, nullptr, 0, out);   // MLXArray(Float): shape []
}

extern "C" int flux2mi_array_wrap(flux2mi_ctx* ctx, float* device, const int64_t* shape, int ndim, flux2mi_array** out) {
    F2_REQUIRE(device, FLUX2MI_ERR_INVALID_ARG, "array_wrap: NULL data");
    flux2mi_array* a = nullptr;
    F2_TRY(make(ctx, shape, ndim, false, &a));
    a->data = device;
    *out = a;
    return FLUX2MI_OK;
}

extern "C" int flux2mi_array_free(flux2mi_array* a) {
    if (!a) return FLUX2MI_OK;
    if (a->owned && a->data) {
        (void)hipSetDevice(a->ctx->device);
        (void)hipStreamSynchronize(a->ctx->stream);   // work reading it may still be queued
        f2_free(a->data);
    }
    delete a;
    return FLUX2MI_OK;
}

extern "C" int flux2mi_array_ndim(const flux2mi_array* a) { return a ? a->ndim : -1; }
extern "C" int64_t flux2mi_array_dim(const flux2mi_array* a, int i) { return (a && i >= 0 && i < a->ndim) ? a->shape[i] : -1; }
extern "C" float* flux2mi_array_data(flux2mi_array* a) { return a ? a->data : nullptr; }

extern "C" int flux2mi_array_add(flux2mi_ctx* ctx, const flux2mi_array* a, const flux2mi_array* b, flux2mi_array** out) { return binary(ctx, 0, a, b, out); }
extern "C" int flux2mi_array_sub(flux2mi_ctx* ctx, const flux2mi_array* a, const flux2mi_array* b, flux2mi_array** out) { return binary(ctx, 1, a, b, out); }
extern "C" int flux2mi_array_mul(flux2mi_ctx* ctx, const flux2mi_array* a, const flux2mi_array* b, flux2mi_array** out) { return binary(ctx, 2, a, b, out); }

extern "C" int flux2mi_array_copy_to(flux2mi_ctx* ctx, const flux2mi_array* src, float* dst_device, int64_t n) {
    F2_REQUIRE(ctx && src && dst_device, FLUX2MI_ERR_INVALID_ARG, "array_copy_to: NULL argument");
    F2_REQUIRE(n == src->numel(), FLUX2MI_ERR_SHAPE, "array_copy_to: array has %lld elements, destination %lld", (long long)src->numel(), (long long)n);
    if (src->data != dst_device)
        F2_CHECK_HIP(hipMemcpyAsync(dst_device, src->data, (size_t)n * 4, hipMemcpyDeviceToDevice, ctx->stream));
    return FLUX2MI_OK;
}

extern "C" int flux2mi_array_to_host(flux2mi_ctx* ctx, const flux2mi_array* a, float* host) {
    F2_REQUIRE(ctx && a && host, FLUX2MI_ERR_INVALID_ARG, "array_to_host: NULL argument");
    F2_CHECK_HIP(hipMemcpyAsync(host, a->data, (size_t)a->numel() * 4, hipMemcpyDeviceToHost, ctx->stream));
    F2_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    return FLUX2MI_OK;
}

extern "C" int flux2mi_array_random_normal(flux2mi_ctx* ctx, const uint32_t key[2], const int64_t* shape, int ndim, flux2mi_array** out) {
    F2_REQUIRE(key, FLUX2MI_ERR_INVALID_ARG, "array_random_normal: NULL key");
    flux2mi_array* a = nullptr;
    F2_TRY(make(ctx, shape, ndim, true, &a));
    const int r = flux2mi_mlx_random_normal(ctx, key, a->data, a->numel());
    if (r != FLUX2MI_OK) { flux2mi_array_free(a); return r; }
    *out = a;
    return FLUX2MI_OK;
}

// `precomputedEmbeddings: MLXArray?` (f32 [1, S_txt, J]) -> the bf16 device buffer flux2mi_generate_params.txt_emb wants
// (Flux2Pipeline.swift:1166, :1381-1404: the reference feeds the array to the transformer as is)
extern "C" int flux2mi_array_to_bf16(flux2mi_ctx* ctx, const flux2mi_array* a, void* dst_bf16_device) {
    F2_REQUIRE(ctx && a && dst_bf16_device && a->ctx == ctx, FLUX2MI_ERR_INVALID_ARG, "array_to_bf16: bad argument");
    F2_CHECK_HIP(hipSetDevice(ctx->device));
    F2_LAUNCH(launch_convert_to_bf16(ctx->stream, a->data, FLUX2MI_F32, reinterpret_cast<uint16_t*>(dst_bf16_device), a->numel()), "array_to_bf16");
    return FLUX2MI_OK;
}
