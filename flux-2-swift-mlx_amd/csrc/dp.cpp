// dp.cpp — batch-level data parallelism across the GPUs of one node: the two exchanges of a batch, on RCCL over
// xGMI, behind the C ABI.
//
// The reference has no distributed code (single process, batch 1: SURVEY F2), so this interface is the build's own.  The path
// shards at image granularity — image i -> rank i mod world, a full model replica per GPU (Dev bf16: 64 GB of 288), NO collective
// inside the denoise loop.  Per batch: broadcast of the prompt embeddings from the rank that ran the text encoder, all-gather of
// the decoded u8 images.  Both run on the CONTEXT'S stream, in stream order with flux2mi_textenc_forward / flux2mi_generate:
// no host synchronisation, no second stream to hand over to.  One process per GPU; the 128-byte unique id travels from rank 0
// to the others through whatever side channel the host already has (a TCP store, a file, an environment variable).
#include <rccl/rccl.h>

#include "common.h"

struct flux2mi_dp {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
};

static_assert(NCCL_UNIQUE_ID_BYTES == FLUX2MI_DP_UNIQUE_ID_BYTES, "unique id size");

#define F2_CHECK_NCCL(expr)                                                                                   \
    do {                                                                                                      \
        ncclResult_t _r = (expr);                                                                             \
        if (_r != ncclSuccess) {                                                                              \
            flux2mi_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, ncclGetErrorString(_r));          \
            return FLUX2MI_ERR_HIP;                                                                           \
        }                                                                                                     \
    } while (0)

extern "C" int flux2mi_dp_unique_id(uint8_t* id) {
    F2_REQUIRE(id, FLUX2MI_ERR_INVALID_ARG, "dp_unique_id: NULL");
    ncclUniqueId u;
    F2_CHECK_NCCL(ncclGetUniqueId(&u));
    memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return FLUX2MI_OK;
}

extern "C" int flux2mi_dp_init(flux2mi_ctx* ctx, int rank, int world, const uint8_t* id) {
    F2_REQUIRE(ctx && id && world >= 1 && rank >= 0 && rank < world, FLUX2MI_ERR_INVALID_ARG, "dp_init: rank %d of %d", rank, world);
    F2_REQUIRE(ctx->dp == nullptr, FLUX2MI_ERR_INVALID_ARG, "dp_init: the context already has a communicator (flux2mi_dp_finalize first)");
    F2_CHECK_HIP(hipSetDevice(ctx->device));
    ncclUniqueId u;
    memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    flux2mi_dp* d = new flux2mi_dp();
    d->rank = rank;
    d->world = world;
    ncclResult_t r = ncclCommInitRank(&d->comm, world, u, rank);
    if (r != ncclSuccess) {
        flux2mi_set_error("dp_init: ncclCommInitRank(rank %d of %d, device %d) -> %s", rank, world, ctx->device, ncclGetErrorString(r));
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
    ncclResult_t r = ncclCommDestroy(ctx->dp->comm);
    delete ctx->dp;
    ctx->dp = nullptr;
    if (r != ncclSuccess) {
        flux2mi_set_error("dp_finalize: ncclCommDestroy -> %s", ncclGetErrorString(r));
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
    F2_CHECK_NCCL(ncclBroadcast(embeddings, embeddings, (size_t)elems, ncclBfloat16, root, ctx->dp->comm, ctx->stream));
    return FLUX2MI_OK;
}

// Exchange 2: every rank's decoded images, u8 [n_local, H, W, 3] = bytes_per_rank bytes, into all_images [world][bytes_per_rank]
// (rank-major: the caller maps (rank, slot) back to the global image index slot * world + rank).
extern "C" int flux2mi_dp_all_gather_images(flux2mi_ctx* ctx, const uint8_t* local_images, uint8_t* all_images, int64_t bytes_per_rank) {
    F2_REQUIRE(ctx && local_images && all_images && bytes_per_rank > 0, FLUX2MI_ERR_INVALID_ARG, "dp_all_gather_images: bad argument");
    F2_REQUIRE(ctx->dp, FLUX2MI_ERR_NOT_LOADED, "dp_all_gather_images: flux2mi_dp_init has not been called on this context");
    F2_CHECK_HIP(hipSetDevice(ctx->device));
    F2_CHECK_NCCL(ncclAllGather(local_images, all_images, (size_t)bytes_per_rank, ncclUint8, ctx->dp->comm, ctx->stream));
    return FLUX2MI_OK;
}

// max over the ranks of one f64 (the bench's max-over-ranks step time; also a barrier).  Synchronises the stream.
extern "C" int flux2mi_dp_all_reduce_max(flux2mi_ctx* ctx, double* value) {
    F2_REQUIRE(ctx && value, FLUX2MI_ERR_INVALID_ARG, "dp_all_reduce_max: bad argument");
    F2_REQUIRE(ctx->dp, FLUX2MI_ERR_NOT_LOADED, "dp_all_reduce_max: flux2mi_dp_init has not been called on this context");
    F2_CHECK_HIP(hipSetDevice(ctx->device));
    F2_TRY(ctx->scratch.reserve(256));
    double* d = ctx->scratch.as<double>();
    F2_CHECK_HIP(hipMemcpyAsync(d, value, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    F2_CHECK_NCCL(ncclAllReduce(d, d, 1, ncclDouble, ncclMax, ctx->dp->comm, ctx->stream));
    F2_CHECK_HIP(hipMemcpyAsync(value, d, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    F2_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    return FLUX2MI_OK;
}
