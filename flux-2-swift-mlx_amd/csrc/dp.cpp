// dp.cpp — batch-level data parallelism across the GPUs of one node: the two exchanges of a batch, on RCCL over
// xGMI, behind the C ABI.
//
// The reference has no distributed code (single process, batch 1: SURVEY F2), so this interface is the build's own.  The path
// shards at image granularity — image i -> rank i mod world, a full model replica per GPU (Dev bf16: 64 GB of 288), NO collective
// inside the denoise loop.  Per batch: broadcast of the prompt embeddings from the rank that ran the text encoder, all-gather of
// the decoded u8 images.  Both run on the CONTEXT'S stream, in stream order with flux2mi_textenc_forward / flux2mi_generate:
// no host synchronisation, no second stream to hand over to.  One process per GPU; the 128-byte unique id travels from rank 0
// to the others through whatever side channel the host already has (a TCP store, a file, an environment variable).
//
// RCCL is bound at run time (dlopen on the first flux2mi_dp_unique_id / flux2mi_dp_init), not linked: single-GPU hosts and hosts whose
// ROCm lives elsewhere load libflux2mi.so without it; the dp entry points then return FLUX2MI_ERR_UNSUPPORTED naming what was tried.
#include <dlfcn.h>
#include <rccl/rccl.h>   // types and constants only
#include <stdio.h>
#include <stdlib.h>

#include <mutex>

#include "common.h"

namespace {
struct Rccl {
    void* handle = nullptr;
    std::string tried, bound;   // files that failed to load / the one that was bound (flux2mi_dp_rccl_library)
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
std::once_flag g_rccl_once;

void rccl_open() {
    // Order: an explicit file; then an RCCL the process has ALREADY mapped (a host that runs torch.distributed brought its own
    // librccl.so.1: binding a second copy by absolute path would mean a second set of proxy threads and possibly another version);
    // only then the ROCm installation and the loader's search path.
    std::vector<std::pair<std::string, int>> names;
    if (const char* e = getenv("FLUX2MI_RCCL_LIB")) names.push_back({e, 0});
    names.push_back({"librccl.so.1", RTLD_NOLOAD});
    names.push_back({"librccl.so", RTLD_NOLOAD});
    if (const char* e = getenv("ROCM_PATH")) names.push_back({std::string(e) + "/lib/librccl.so.1", 0});
    names.push_back({"/opt/rocm/lib/librccl.so.1", 0});
    names.push_back({"librccl.so.1", 0});
    names.push_back({"librccl.so", 0});
    for (const auto& n : names) {
        g_rccl.handle = dlopen(n.first.c_str(), RTLD_NOW | RTLD_LOCAL | n.second);
        if (g_rccl.handle) {
            g_rccl.bound = n.first + (n.second ? " (already loaded)" : "");
            break;
        }
        if (!n.second) g_rccl.tried += (g_rccl.tried.empty() ? "" : ", ") + n.first;
    }
    if (!g_rccl.handle) return;
    if (getenv("FLUX2MI_LOG")) fprintf(stderr, "[flux2mi] RCCL bound: %s\n", g_rccl.bound.c_str());
    bool ok = true;
    auto sym = [&](const char* name) { void* p = dlsym(g_rccl.handle, name); if (!p) { ok = false; g_rccl.tried = std::string("symbol ") + name; } return p; };
    g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(sym("ncclGetUniqueId"));
    g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(sym("ncclCommInitRank"));
    g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(sym("ncclCommDestroy"));
    g_rccl.Broadcast = reinterpret_cast<decltype(g_rccl.Broadcast)>(sym("ncclBroadcast"));
    g_rccl.AllGather = reinterpret_cast<decltype(g_rccl.AllGather)>(sym("ncclAllGather"));
    g_rccl.AllReduce = reinterpret_cast<decltype(g_rccl.AllReduce)>(sym("ncclAllReduce"));
    g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(sym("ncclGetErrorString"));
    if (!ok) { dlclose(g_rccl.handle); g_rccl.handle = nullptr; }
}
// 0 when RCCL is bound; FLUX2MI_ERR_UNSUPPORTED (with the error string set) otherwise
int rccl_require() {
    std::call_once(g_rccl_once, rccl_open);
    if (g_rccl.handle) return FLUX2MI_OK;
    flux2mi_set_error("data parallelism needs RCCL and none could be loaded (tried: %s); set FLUX2MI_RCCL_LIB or ROCM_PATH", g_rccl.tried.c_str());
    return FLUX2MI_ERR_UNSUPPORTED;
}
}  // namespace

struct flux2mi_dp {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    // flux2mi_dp_collective_timing: HIP event pairs on the context's stream around each exchange, by kind (0 broadcast, 1 all-gather)
    bool timing = false;
    struct Pair { hipEvent_t a, b; int kind; };
    std::vector<Pair> pending;
    std::vector<hipEvent_t> pool;
    double ms[2] = {0.0, 0.0};
    long long calls[2] = {0, 0};
    long long bytes[2] = {0, 0};
};

namespace {
hipEvent_t dp_event(flux2mi_dp* d) {
    if (!d->pool.empty()) { hipEvent_t e = d->pool.back(); d->pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    return hipEventCreate(&e) == hipSuccess ? e : nullptr;
}
// folds the finished pairs into the sums (synchronises the stream when there are any)
void dp_fold(flux2mi_ctx* ctx) {
    flux2mi_dp* d = ctx->dp;
    if (!d || d->pending.empty()) return;
    (void)hipStreamSynchronize(ctx->stream);
    for (const flux2mi_dp::Pair& p : d->pending) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, p.a, p.b) == hipSuccess) d->ms[p.kind] += (double)t;
        d->pool.push_back(p.a);
        d->pool.push_back(p.b);
    }
    d->pending.clear();
}
struct DpTimed {   // start event at construction, stop event at destruction, both on the context's stream
    flux2mi_ctx* ctx; int kind; hipEvent_t a = nullptr, b = nullptr;
    DpTimed(flux2mi_ctx* c, int k, long long nbytes) : ctx(c), kind(k) {
        flux2mi_dp* d = c->dp;
        d->calls[k] += 1;
        d->bytes[k] += nbytes;
        if (!d->timing) return;
        if (d->pending.size() >= 1024) dp_fold(c);
        a = dp_event(d);
        b = dp_event(d);
        if (a && b) (void)hipEventRecord(a, c->stream);
    }
    ~DpTimed() {
        if (!a || !b) return;
        (void)hipEventRecord(b, ctx->stream);
        ctx->dp->pending.push_back({a, b, kind});
    }
};
}  // namespace

static_assert(NCCL_UNIQUE_ID_BYTES == FLUX2MI_DP_UNIQUE_ID_BYTES, "unique id size");

#define F2_CHECK_NCCL(expr)                                                                                   \
    do {                                                                                                      \
        ncclResult_t _r = (expr);                                                                             \
        if (_r != ncclSuccess) {                                                                              \
            flux2mi_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, g_rccl.GetErrorString(_r));          \
            return FLUX2MI_ERR_HIP;                                                                           \
        }                                                                                                     \
    } while (0)

extern "C" const char* flux2mi_dp_rccl_library(void) {
    std::call_once(g_rccl_once, rccl_open);
    return g_rccl.handle ? g_rccl.bound.c_str() : nullptr;
}

extern "C" int flux2mi_dp_unique_id(uint8_t* id) {
    F2_REQUIRE(id, FLUX2MI_ERR_INVALID_ARG, "dp_unique_id: NULL");
    F2_TRY(rccl_require());
    ncclUniqueId u;
    F2_CHECK_NCCL(g_rccl.GetUniqueId(&u));
    memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return FLUX2MI_OK;
}

extern "C" int flux2mi_dp_init(flux2mi_ctx* ctx, int rank, int world, const uint8_t* id) {
    F2_REQUIRE(ctx && id && world >= 1 && rank >= 0 && rank < world, FLUX2MI_ERR_INVALID_ARG, "dp_init: rank %d of %d", rank, world);
    F2_REQUIRE(ctx->dp == nullptr, FLUX2MI_ERR_INVALID_ARG, "dp_init: the context already has a communicator (flux2mi_dp_finalize first)");
    F2_TRY(rccl_require());
    F2_CHECK_HIP(hipSetDevice(ctx->device));
    ncclUniqueId u;
    memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    flux2mi_dp* d = new flux2mi_dp();
    d->rank = rank;
    d->world = world;
    ncclResult_t r = g_rccl.CommInitRank(&d->comm, world, u, rank);
    if (r != ncclSuccess) {
        flux2mi_set_error("dp_init: ncclCommInitRank(rank %d of %d, device %d) -> %s", rank, world, ctx->device, g_rccl.GetErrorString(r));
        delete d;
        return FLUX2MI_ERR_HIP;
    }
    ctx->dp = d;
    return FLUX2MI_OK;
}

extern "C" int flux2mi_dp_finalize(flux2mi_ctx* ctx) {
    F2_REQUIRE(ctx, FLUX2MI_ERR_INVALID_ARG, "dp_finalize: NULL");
    if (!ctx->dp) return FLUX2MI_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    dp_fold(ctx);
    for (hipEvent_t e : ctx->dp->pool) (void)hipEventDestroy(e);
    ncclResult_t r = g_rccl.CommDestroy(ctx->dp->comm);
    delete ctx->dp;
    ctx->dp = nullptr;
    if (r != ncclSuccess) {
        flux2mi_set_error("dp_finalize: ncclCommDestroy -> %s", g_rccl.GetErrorString(r));
        return FLUX2MI_ERR_HIP;
    }
    return FLUX2MI_OK;
}

extern "C" int flux2mi_dp_rank(const flux2mi_ctx* ctx) { return ctx && ctx->dp ? ctx->dp->rank : 0; }
extern "C" int flux2mi_dp_world(const flux2mi_ctx* ctx) { return ctx && ctx->dp ? ctx->dp->world : 1; }

// Exchange 1: the prompt embeddings of the whole batch, bf16 [n_images, s_txt, joint_dim], in place, from `root`.
extern "C" int flux2mi_dp_broadcast_embeddings(flux2mi_ctx* ctx, void* embeddings, int64_t elems, int root) {
    F2_REQUIRE(ctx && embeddings && elems > 0, FLUX2MI_ERR_INVALID_ARG, "dp_broadcast_embeddings: bad argument");
    F2_REQUIRE(ctx->dp, FLUX2MI_ERR_NOT_LOADED, "dp_broadcast_embeddings: flux2mi_dp_init has not been called on this context");
    F2_REQUIRE(root >= 0 && root < ctx->dp->world, FLUX2MI_ERR_INVALID_ARG, "dp_broadcast_embeddings: root %d of %d", root, ctx->dp->world);
    F2_CHECK_HIP(hipSetDevice(ctx->device));
    DpTimed timed(ctx, 0, (long long)elems * 2);
    F2_CHECK_NCCL(g_rccl.Broadcast(embeddings, embeddings, (size_t)elems, ncclBfloat16, root, ctx->dp->comm, ctx->stream));
    return FLUX2MI_OK;
}

// Exchange 2: every rank's decoded images, u8 [n_local, H, W, 3] = bytes_per_rank bytes, into all_images [world][bytes_per_rank]
// (rank-major: the caller maps (rank, slot) back to the global image index slot * world + rank).
extern "C" int flux2mi_dp_all_gather_images(flux2mi_ctx* ctx, const uint8_t* local_images, uint8_t* all_images, int64_t bytes_per_rank) {
    F2_REQUIRE(ctx && local_images && all_images && bytes_per_rank > 0, FLUX2MI_ERR_INVALID_ARG, "dp_all_gather_images: bad argument");
    F2_REQUIRE(ctx->dp, FLUX2MI_ERR_NOT_LOADED, "dp_all_gather_images: flux2mi_dp_init has not been called on this context");
    F2_CHECK_HIP(hipSetDevice(ctx->device));
    DpTimed timed(ctx, 1, (long long)bytes_per_rank * ctx->dp->world);
    F2_CHECK_NCCL(g_rccl.AllGather(local_images, all_images, (size_t)bytes_per_rank, ncclUint8, ctx->dp->comm, ctx->stream));
    return FLUX2MI_OK;
}

// Per-exchange timings for the bench line (SURVEY §8e "Measurability"): with timing on, every broadcast / all-gather is bracketed by
// HIP events on the context's stream — the time between them is what the exchange costs the generating stream at that point,
// waiting for the slowest peer included.  flux2mi_dp_collective_stats folds what has finished (synchronises the stream) and
// returns, per kind (0 broadcast, 1 all-gather): summed milliseconds, number of calls, payload bytes; `reset` clears the sums.
extern "C" int flux2mi_dp_collective_timing(flux2mi_ctx* ctx, int on) {
    F2_REQUIRE(ctx, FLUX2MI_ERR_INVALID_ARG, "dp_collective_timing: NULL ctx");
    F2_REQUIRE(ctx->dp, FLUX2MI_ERR_NOT_LOADED, "dp_collective_timing: flux2mi_dp_init has not been called on this context");
    ctx->dp->timing = on != 0;
    return FLUX2MI_OK;
}
extern "C" int flux2mi_dp_collective_stats(flux2mi_ctx* ctx, double ms[2], int64_t calls[2], int64_t bytes[2], int reset) {
    F2_REQUIRE(ctx, FLUX2MI_ERR_INVALID_ARG, "dp_collective_stats: NULL ctx");
    F2_REQUIRE(ctx->dp, FLUX2MI_ERR_NOT_LOADED, "dp_collective_stats: flux2mi_dp_init has not been called on this context");
    F2_CHECK_HIP(hipSetDevice(ctx->device));
    dp_fold(ctx);
    flux2mi_dp* d = ctx->dp;
    for (int k = 0; k < 2; ++k) {
        if (ms) ms[k] = d->ms[k];
        if (calls) calls[k] = d->calls[k];
        if (bytes) bytes[k] = d->bytes[k];
        if (reset) { d->ms[k] = 0.0; d->calls[k] = 0; d->bytes[k] = 0; }
    }
    return FLUX2MI_OK;
}

// max over the ranks of one f64 (the bench's max-over-ranks step time; also a barrier).  Synchronises the stream.
extern "C" int flux2mi_dp_all_reduce_max(flux2mi_ctx* ctx, double* value) {
    F2_REQUIRE(ctx && value, FLUX2MI_ERR_INVALID_ARG, "dp_all_reduce_max: bad argument");
    F2_WORKSPACE_USE(ctx->device);
    F2_REQUIRE(ctx->dp, FLUX2MI_ERR_NOT_LOADED, "dp_all_reduce_max: flux2mi_dp_init has not been called on this context");
    F2_CHECK_HIP(hipSetDevice(ctx->device));
    F2_TRY(ctx->scratch.reserve(256));
    double* d = ctx->scratch.as<double>();
    F2_CHECK_HIP(hipMemcpyAsync(d, value, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    F2_CHECK_NCCL(g_rccl.AllReduce(d, d, 1, ncclDouble, ncclMax, ctx->dp->comm, ctx->stream));
    F2_CHECK_HIP(hipMemcpyAsync(value, d, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    F2_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    return FLUX2MI_OK;
}
