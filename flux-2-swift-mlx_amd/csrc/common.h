// common.h — host-side shared declarations of libflux2mi (context, errors, buffers).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <string>
#include <vector>
#include "../../include/flux2mi.h"

typedef uint16_t bf16_t;  // raw bfloat16 bits on the host side

void flux2mi_set_error(const char* fmt, ...);

#define F2_CHECK_HIP(expr)                                                                  \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            flux2mi_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return FLUX2MI_ERR_HIP;                                                         \
        }                                                                                   \
    } while (0)

#define F2_REQUIRE(cond, code, ...)            \
    do {                                       \
        if (!(cond)) {                         \
            flux2mi_set_error(__VA_ARGS__);    \
            return (code);                     \
        }                                      \
    } while (0)

#define F2_TRY(expr)                 \
    do {                             \
        int _s = (expr);             \
        if (_s != FLUX2MI_OK) return _s; \
    } while (0)

#define F2_LAUNCH(expr, what)                                                                   \
    do {                                                                                        \
        int _r = (expr);                                                                        \
        if (_r == 1) { flux2mi_set_error("%s: invalid argument", what); return FLUX2MI_ERR_INVALID_ARG; } \
        if (_r != 0) { flux2mi_set_error("%s: kernel launch failed: %s", what, hipGetErrorString(hipGetLastError())); return FLUX2MI_ERR_HIP; } \
    } while (0)

// kernel classes for flux2mi_prof_begin (bit masks passed to prof_before/after)
// classes 5 and 6 are HBM-bound: their prof_after "flops" argument carries the ALGORITHMIC BYTES of the launch
enum { PROF_LINEAR1 = 1 << 0, PROF_DIT_GEMM = 1 << 1, PROF_ATTENTION = 1 << 2, PROF_VAE_CONV = 1 << 3, PROF_LINEAR2 = 1 << 4,
       PROF_LN_MODULATE = 1 << 5, PROF_GN_APPLY = 1 << 6, PROF_PROJ_MLPOUT = 1 << 7, PROF_VAE_ATTN = 1 << 8 };

// Accounted device memory: every hipMalloc / hipFree of the library goes through these two, so that
// flux2mi_ctx_memory_stats (the MLX.Memory.activeMemory / peakMemory / cacheMemory of the shim) reports what the library holds.
// workspace = true: bytes of a trimmable DevBuf ("cache" in MLX's terms: held for reuse, given back by flux2mi_ctx_trim).
#define FLUX2MI_MAX_DEVICES 16
hipError_t f2_malloc(void** p, size_t bytes, bool workspace = false);
void f2_free(void* p);
// Threading contract of the workspaces (MLX.Memory.clearCache is callable from any thread, and so is flux2mi_ctx_trim): every entry
// point that reserves or hands out DevBuf memory holds a WorkspaceUse of its device for the duration of the call (nestable: generate ->
// forward).  flux2mi_ctx_trim gives nothing back while a use is open on the device (also: called from a hook of a running
// generation), keeps new uses and DevBuf destructors waiting while it releases, and drains the device's queued work first.
struct WorkspaceUse {
    int device;
    explicit WorkspaceUse(int device);
    ~WorkspaceUse();
    WorkspaceUse(const WorkspaceUse&) = delete;
    WorkspaceUse& operator=(const WorkspaceUse&) = delete;
};
#define F2_WORKSPACE_USE(dev) WorkspaceUse _f2_ws_use(dev)

// Grow-only device buffer (no allocation inside the steady-state loop).  Registered process-wide so that flux2mi_ctx_trim can
// release the ones of its device; `keep` marks buffers whose contents are built once (derived weights) and must survive a trim.
// reserve() is only called under a WorkspaceUse of the buffer's device; the destructor waits for a running trim.
struct DevBuf {
    void* ptr = nullptr;
    size_t bytes = 0;
    int device = -1;
    bool keep = false;
    DevBuf();
    ~DevBuf();
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    int reserve(size_t n);
    void release();
    template <class T> T* as() { return reinterpret_cast<T*>(ptr); }
};

struct ProfState {
    int active_class = -1;
    std::vector<hipEvent_t> pool;   // pairs (start, stop)
    size_t used = 0;
    double flops = 0.0;
};

struct flux2mi_dp;   // dp.cpp: the RCCL communicator of a data-parallel batch
struct flux2mi_dit;
// dit.cpp, for flux2mi_generate: between on and off, plain forwards that get the SAME text embedding pointer / length and the SAME id
// pointers reuse the text rows of the input projection (context_embedder: it does not depend on the step) and the RoPE tables of the
// previous call instead of recomputing them — the caller guarantees that the contents behind those pointers do not change in between.
// Results are bit-identical to the uncached forward.  Off (the default) outside a generation.
void flux2mi_dit_step_cache(flux2mi_dit* m, bool on);
// With the cache on: the timestep embeddings and modulation vectors of the coming steps (timesteps ts[0..n), one guidance value) in one
// pass over the weights; a plain forward whose (timestep, guidance) is among them takes its vectors from there.
int flux2mi_dit_prepare_steps(flux2mi_dit* m, const float* ts, int n, float guidance);

struct flux2mi_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    ProfState prof;
    DevBuf scratch;  // small misc scratch
    int n_cu = 0;    // compute units of the device (workgroup slots for the attention's KV-split)
    DevBuf attn_ws;  // raw accumulators of the attention's KV-split pieces (attention64b_split_ws_bytes: n_cu x 135 KB)
    flux2mi_dp* dp = nullptr;   // flux2mi_dp_init
    // flux2mi_generate's own buffers (packed latents, prediction(s), position ids, VAE input, image bytes): grow-only workspaces of the
    // context like every other — no hipMalloc / hipFree (a device-wide synchronisation) per image
    struct GenScratch { DevBuf packed, lat_in, v, v2, ids_txt, ids_neg, ids_img, z, img_u8; } gen;
};

// profiling helpers: bracket a launch of class `cls` with events when active.
void prof_before(flux2mi_ctx* ctx, int cls_mask);
void prof_after(flux2mi_ctx* ctx, int cls_mask, double flops);
