// capi.cpp — context, errors, host-side scheduler / latent utilities and the thin
// wrappers of the elementwise device ops.  Host logic restates, in C++, the Swift host
// code of the reference (no Swift toolchain exists on the target image).
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <unordered_map>
#include <unordered_set>

#include "common.h"
#include "kernels.h"

// ------------------------------------------------------------------ errors
static thread_local char g_err[1024] = "";

void flux2mi_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* flux2mi_last_error(void) { return g_err; }
extern "C" int flux2mi_version(void) { return FLUX2MI_VERSION; }

// ------------------------------------------------------------------ accounted device memory + DevBuf
// MLX.Memory.{activeMemory, peakMemory, cacheMemory, clearCache} (mlx-swift 0.31.6; called at Flux2CLI/InpaintCommand.swift:253-271,
// CompareEncoders.swift:112,142) report MLX's allocator.  Here: one table of the library's live device allocations per process.
namespace {
struct MemTable {
    std::mutex mu;
    struct Rec { size_t bytes; int device; bool workspace; };
    std::unordered_map<void*, Rec> live;
    std::unordered_set<DevBuf*> bufs;
    size_t active[FLUX2MI_MAX_DEVICES] = {0}, workspace[FLUX2MI_MAX_DEVICES] = {0}, peak[FLUX2MI_MAX_DEVICES] = {0};
    // per device: open WorkspaceUse count, and whether a trim is releasing right now (uses and DevBuf destructors wait on cv)
    int users[FLUX2MI_MAX_DEVICES] = {0};
    bool trimming[FLUX2MI_MAX_DEVICES] = {false};
    std::condition_variable cv;
};
int dev_slot(int device) { return device >= 0 && device < FLUX2MI_MAX_DEVICES ? device : 0; }
MemTable& mem() {
    static MemTable* t = new MemTable();   // never destroyed: DevBufs of static objects may outlive any static table
    return *t;
}
}  // namespace

hipError_t f2_malloc(void** p, size_t bytes, bool workspace) {
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) return e;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= FLUX2MI_MAX_DEVICES) dev = 0;
    MemTable& t = mem();
    std::lock_guard<std::mutex> g(t.mu);
    t.live[*p] = MemTable::Rec{bytes, dev, workspace};
    (workspace ? t.workspace : t.active)[dev] += bytes;
    const size_t total = t.active[dev] + t.workspace[dev];
    if (total > t.peak[dev]) t.peak[dev] = total;
    return hipSuccess;
}
void f2_free(void* p) {
    if (!p) return;
    {
        MemTable& t = mem();
        std::lock_guard<std::mutex> g(t.mu);
        auto it = t.live.find(p);
        if (it != t.live.end()) {
            (it->second.workspace ? t.workspace : t.active)[it->second.device] -= it->second.bytes;
            t.live.erase(it);
        }
    }
    (void)hipFree(p);
}
WorkspaceUse::WorkspaceUse(int dev) : device(dev_slot(dev)) {
    MemTable& t = mem();
    std::unique_lock<std::mutex> g(t.mu);
    // a nested use (users > 0) never waits: a trim only starts with no use open
    t.cv.wait(g, [&] { return t.users[device] > 0 || !t.trimming[device]; });
    ++t.users[device];
}
WorkspaceUse::~WorkspaceUse() {
    MemTable& t = mem();
    std::lock_guard<std::mutex> g(t.mu);
    --t.users[device];
}

DevBuf::DevBuf() { std::lock_guard<std::mutex> g(mem().mu); mem().bufs.insert(this); }
DevBuf::~DevBuf() {
    {   // leave the table first (a trim collects its victims from it), and not under a trim that may hold this buffer as a victim.
        // The wait is bounded: a trim sits in hipDeviceSynchronize() meanwhile, and a wedged device must not turn teardown (static
        // objects at process exit included) into a hang — after the bound the buffer leaves the table anyway (the trim re-checks
        // membership under the lock before it touches a victim, so this is safe) and the HIP error surfaces from the trim's caller.
        MemTable& t = mem();
        std::unique_lock<std::mutex> g(t.mu);
        if (!t.cv.wait_for(g, std::chrono::seconds(30), [&] { return device < 0 || !t.trimming[dev_slot(device)]; }))
            fprintf(stderr, "[flux2mi] a workspace trim on device %d has not finished within 30 s (device wedged?): releasing a buffer beside it\n", device);
        t.bufs.erase(this);
    }
    release();
}
// ptr / bytes / device are published under the table's mutex: flux2mi_ctx_trim reads them for EVERY registered buffer (other devices'
// included) while other threads reserve on their own devices
int DevBuf::reserve(size_t n) {
    if (n <= bytes) return FLUX2MI_OK;
    release();
    const size_t rounded = (n + 255) & ~(size_t)255;
    void* p = nullptr;
    hipError_t e = f2_malloc(&p, rounded, !keep);
    if (e != hipSuccess) {
        flux2mi_set_error("hipMalloc(%zu bytes) failed: %s", rounded, hipGetErrorString(e));
        return FLUX2MI_ERR_OOM;
    }
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> g(mem().mu);
    ptr = p;
    device = dev;
    bytes = rounded;
    return FLUX2MI_OK;
}
void DevBuf::release() {
    void* p = nullptr;
    {
        std::lock_guard<std::mutex> g(mem().mu);
        p = ptr;
        ptr = nullptr;
        bytes = 0;
    }
    if (p) f2_free(p);
}

extern "C" int flux2mi_ctx_memory_stats(flux2mi_ctx* ctx, size_t* active_bytes, size_t* peak_bytes, size_t* workspace_bytes) {
    F2_REQUIRE(ctx, FLUX2MI_ERR_INVALID_ARG, "ctx_memory_stats: NULL ctx");
    MemTable& t = mem();
    std::lock_guard<std::mutex> g(t.mu);
    const int d = ctx->device < FLUX2MI_MAX_DEVICES ? ctx->device : 0;
    if (active_bytes) *active_bytes = t.active[d];
    if (peak_bytes) *peak_bytes = t.peak[d];
    if (workspace_bytes) *workspace_bytes = t.workspace[d];
    return FLUX2MI_OK;
}
extern "C" int flux2mi_ctx_reset_peak(flux2mi_ctx* ctx) {
    F2_REQUIRE(ctx, FLUX2MI_ERR_INVALID_ARG, "ctx_reset_peak: NULL ctx");
    MemTable& t = mem();
    std::lock_guard<std::mutex> g(t.mu);
    const int d = ctx->device < FLUX2MI_MAX_DEVICES ? ctx->device : 0;
    t.peak[d] = t.active[d] + t.workspace[d];
    return FLUX2MI_OK;
}
extern "C" int flux2mi_ctx_trim(flux2mi_ctx* ctx, size_t* freed_bytes) {
    F2_REQUIRE(ctx, FLUX2MI_ERR_INVALID_ARG, "ctx_trim: NULL ctx");
    if (freed_bytes) *freed_bytes = 0;
    F2_CHECK_HIP(hipSetDevice(ctx->device));
    std::vector<DevBuf*> victims;
    MemTable& t = mem();
    const int d = dev_slot(ctx->device);
    {
        std::lock_guard<std::mutex> g(t.mu);
        // a call that uses the workspaces is running on this device (another thread's forward / decode / merge, or the generation whose
        // hook called us), or another trim is: nothing to give back now
        if (t.users[d] > 0 || t.trimming[d]) return FLUX2MI_OK;
        for (DevBuf* b : t.bufs)
            if (b->ptr && !b->keep && b->device == ctx->device) victims.push_back(b);
        if (victims.empty()) return FLUX2MI_OK;
        t.trimming[d] = true;   // from here to the end: no new use starts on the device, no victim is destroyed
    }
    const hipError_t se = hipDeviceSynchronize();   // work queued on any context of this device may still use them
    size_t freed = 0;
    std::vector<void*> stolen;
    {
        std::lock_guard<std::mutex> g(t.mu);
        if (se == hipSuccess)
            for (DevBuf* b : victims) {
                if (!t.bufs.count(b) || !b->ptr) continue;   // destroyed beside a slow trim (its destructor's bounded wait ran out)
                freed += b->bytes;
                stolen.push_back(b->ptr);
                b->ptr = nullptr;
                b->bytes = 0;
            }
        t.trimming[d] = false;
    }
    for (void* p : stolen) f2_free(p);
    t.cv.notify_all();
    F2_CHECK_HIP(se);
    if (freed_bytes) *freed_bytes = freed;
    return FLUX2MI_OK;
}

// ------------------------------------------------------------------ context
extern "C" int flux2mi_ctx_create(int device, void* hip_stream, flux2mi_ctx** out) {
    F2_REQUIRE(out != nullptr, FLUX2MI_ERR_INVALID_ARG, "ctx_create: out is NULL");
    int ndev = 0;
    F2_CHECK_HIP(hipGetDeviceCount(&ndev));
    F2_REQUIRE(device >= 0 && device < ndev, FLUX2MI_ERR_INVALID_ARG, "ctx_create: device %d of %d", device, ndev);
    F2_CHECK_HIP(hipSetDevice(device));
    flux2mi_ctx* c = new flux2mi_ctx();
    c->device = device;
    if (hipDeviceGetAttribute(&c->n_cu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) c->n_cu = 0;
    if (hip_stream) {
        c->stream = reinterpret_cast<hipStream_t>(hip_stream);
        c->own_stream = false;
    } else {
        hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete c;
            flux2mi_set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
            return FLUX2MI_ERR_HIP;
        }
        c->own_stream = true;
    }
    *out = c;
    return FLUX2MI_OK;
}

extern "C" int flux2mi_ctx_destroy(flux2mi_ctx* ctx) {
    if (!ctx) return FLUX2MI_OK;
    F2_WORKSPACE_USE(ctx->device);   // explicit releases below: not beside a trim
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)flux2mi_dp_finalize(ctx);
    for (hipEvent_t e : ctx->prof.pool) (void)hipEventDestroy(e);
    ctx->scratch.release();
    ctx->attn_ws.release();
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return FLUX2MI_OK;
}

// ---- plain device buffers for a host that has no HIP binding of its own (the Swift shim: prompt embeddings, token ids, image bytes)
extern "C" int flux2mi_buffer_alloc(flux2mi_ctx* ctx, size_t bytes, void** out) {
    F2_REQUIRE(ctx && out && bytes > 0, FLUX2MI_ERR_INVALID_ARG, "buffer_alloc: bad argument");
    F2_CHECK_HIP(hipSetDevice(ctx->device));
    hipError_t e = f2_malloc(out, bytes);
    if (e != hipSuccess) {
        flux2mi_set_error("buffer_alloc: hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
        return FLUX2MI_ERR_OOM;
    }
    return FLUX2MI_OK;
}
extern "C" int flux2mi_buffer_free(flux2mi_ctx* ctx, void* p) {
    F2_REQUIRE(ctx, FLUX2MI_ERR_INVALID_ARG, "buffer_free: NULL ctx");
    if (!p) return FLUX2MI_OK;
    F2_CHECK_HIP(hipSetDevice(ctx->device));
    F2_CHECK_HIP(hipStreamSynchronize(ctx->stream));   // queued work may still read it
    f2_free(p);
    return FLUX2MI_OK;
}
// host -> device in stream order; the host bytes may be reused on return
extern "C" int flux2mi_buffer_upload(flux2mi_ctx* ctx, void* dst_device, const void* src_host, size_t bytes) {
    F2_REQUIRE(ctx && dst_device && src_host && bytes > 0, FLUX2MI_ERR_INVALID_ARG, "buffer_upload: bad argument");
    F2_CHECK_HIP(hipSetDevice(ctx->device));
    F2_CHECK_HIP(hipMemcpyAsync(dst_device, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    F2_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    return FLUX2MI_OK;
}
// device -> host behind everything queued on the stream; synchronises
extern "C" int flux2mi_buffer_download(flux2mi_ctx* ctx, void* dst_host, const void* src_device, size_t bytes) {
    F2_REQUIRE(ctx && dst_host && src_device && bytes > 0, FLUX2MI_ERR_INVALID_ARG, "buffer_download: bad argument");
    F2_CHECK_HIP(hipSetDevice(ctx->device));
    F2_CHECK_HIP(hipMemcpyAsync(dst_host, src_device, bytes, hipMemcpyDeviceToHost, ctx->stream));
    F2_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    return FLUX2MI_OK;
}

extern "C" int flux2mi_ctx_sync(flux2mi_ctx* ctx) {
    F2_REQUIRE(ctx, FLUX2MI_ERR_INVALID_ARG, "ctx_sync: NULL ctx");
    F2_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    return FLUX2MI_OK;
}
extern "C" void* flux2mi_ctx_stream(flux2mi_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

// ------------------------------------------------------------------ profiling (HIP events around one kernel class)
void prof_before(flux2mi_ctx* ctx, int cls_mask) {
    ProfState& p = ctx->prof;
    if (p.active_class < 0 || !(cls_mask & (1 << p.active_class))) return;
    if (p.used + 2 > p.pool.size()) {
        for (int i = 0; i < 512; ++i) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return;
            p.pool.push_back(e);
        }
    }
    (void)hipEventRecord(p.pool[p.used], ctx->stream);
}
void prof_after(flux2mi_ctx* ctx, int cls_mask, double flops) {
    ProfState& p = ctx->prof;
    if (p.active_class < 0 || !(cls_mask & (1 << p.active_class))) return;
    if (p.used + 2 > p.pool.size()) return;
    (void)hipEventRecord(p.pool[p.used + 1], ctx->stream);
    p.used += 2;
    p.flops += flops;
}

extern "C" int flux2mi_debug_mfma_stream(flux2mi_ctx* ctx, int fp8, double seconds, double* tflops) {
    F2_REQUIRE(ctx && tflops && seconds > 0.0 && seconds <= 10.0 && (fp8 == 0 || fp8 == 1), FLUX2MI_ERR_INVALID_ARG, "debug_mfma_stream: bad args");
    F2_WORKSPACE_USE(ctx->device);
    int n_cu = ctx->n_cu;
    if (n_cu <= 0) {
        hipDeviceProp_t pr;
        F2_CHECK_HIP(hipGetDeviceProperties(&pr, ctx->device));
        n_cu = pr.multiProcessorCount;
    }
    F2_TRY(ctx->scratch.reserve((size_t)n_cu * 256 * sizeof(float)));
    const int r = launch_mfma_stream(ctx->stream, ctx->scratch.as<float>(), n_cu, fp8 != 0, seconds, tflops);
    F2_REQUIRE(r == 0, FLUX2MI_ERR_HIP, "debug_mfma_stream: launch failed");
    return FLUX2MI_OK;
}

extern "C" int flux2mi_prof_begin(flux2mi_ctx* ctx, int kernel_class) {
    F2_REQUIRE(ctx && kernel_class >= 0 && kernel_class < 9, FLUX2MI_ERR_INVALID_ARG, "prof_begin: bad args");
    ProfState& p = ctx->prof;
    p.used = 0;
    p.flops = 0.0;
    while (p.pool.size() < 8192) {
        hipEvent_t e;
        F2_CHECK_HIP(hipEventCreate(&e));
        p.pool.push_back(e);
    }
    p.active_class = kernel_class;
    return FLUX2MI_OK;
}
extern "C" int flux2mi_prof_end(flux2mi_ctx* ctx, int* n_launches, double* total_ms, double* flops) {
    F2_REQUIRE(ctx, FLUX2MI_ERR_INVALID_ARG, "prof_end: NULL ctx");
    ProfState& p = ctx->prof;
    p.active_class = -1;
    F2_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    double ms = 0.0;
    for (size_t i = 0; i + 1 < p.used; i += 2) {
        float t = 0.f;
        F2_CHECK_HIP(hipEventElapsedTime(&t, p.pool[i], p.pool[i + 1]));
        ms += t;
    }
    if (n_launches) *n_launches = (int)(p.used / 2);
    if (total_ms) *total_ms = ms;
    if (flops) *flops = p.flops;
    p.used = 0;
    return FLUX2MI_OK;
}

// ------------------------------------------------------------------ scheduler (host)
// Flux2Core/Scheduler/FlowMatchEulerScheduler.swift — Swift `Float` arithmetic == C float.
extern "C" float flux2mi_compute_empirical_mu(int image_seq_len, int num_steps) {  // :9-28
    const float a1 = 8.73809524e-05f, b1 = 1.89833333f;
    const float a2 = 0.00016927f, b2 = 0.45666666f;
    if (image_seq_len > 4300) return a2 * (float)image_seq_len + b2;
    const float m_200 = a2 * (float)image_seq_len + b2;
    const float m_10 = a1 * (float)image_seq_len + b1;
    const float a = (m_200 - m_10) / 190.0f;
    const float b = m_200 - 200.0f * a;
    return a * (float)num_steps + b;
}

static float time_shift(float mu, float sigma, float t) {  // :123-128
    const float exp_mu = expf(mu);
    const float den = exp_mu + powf(1.0f / t - 1.0f, sigma);
    return exp_mu / den;
}

extern "C" int flux2mi_set_timesteps(int num_inference_steps, int image_seq_len, float strength,
                                     float* sigmas_out, int capacity, int* n_out, int* t_start_out) {  // :65-115
    F2_REQUIRE(num_inference_steps > 0 && sigmas_out && n_out, FLUX2MI_ERR_INVALID_ARG,
               "set_timesteps: steps=%d", num_inference_steps);
    const int seq = image_seq_len > 0 ? image_seq_len : 4096;
    const float mu = flux2mi_compute_empirical_mu(seq, num_inference_steps);
    std::vector<float> all((size_t)num_inference_steps + 1);
    for (int i = 0; i < num_inference_steps; ++i) {
        const float s = 1.0f - (float)i / (float)num_inference_steps;
        all[i] = time_shift(mu, 1.0f, s);
    }
    all[num_inference_steps] = 0.0f;
    const float clamped = fmaxf(0.01f, fminf(1.0f, strength));
    const int init_idx = num_inference_steps - (int)((float)num_inference_steps * clamped);
    const int t_start = init_idx > 0 ? init_idx : 0;
    const int n = num_inference_steps + 1 - t_start;
    F2_REQUIRE(n <= capacity, FLUX2MI_ERR_INVALID_ARG, "set_timesteps: capacity %d < %d", capacity, n);
    for (int i = 0; i < n; ++i) sigmas_out[i] = all[t_start + i];
    *n_out = n;
    if (t_start_out) *t_start_out = t_start;
    return FLUX2MI_OK;
}

extern "C" int flux2mi_set_custom_sigmas(const float* sigmas_in, int n_in, float* sigmas_out, int capacity,
                                         int* n_out) {  // :236-260
    F2_REQUIRE(sigmas_in && n_in > 0 && sigmas_out && n_out, FLUX2MI_ERR_INVALID_ARG,
               "set_custom_sigmas: empty sigma list is ignored by the reference");
    const bool append = sigmas_in[n_in - 1] != 0.0f;
    const int n = n_in + (append ? 1 : 0);
    F2_REQUIRE(n <= capacity, FLUX2MI_ERR_INVALID_ARG, "set_custom_sigmas: capacity %d < %d", capacity, n);
    for (int i = 0; i < n_in; ++i) sigmas_out[i] = sigmas_in[i];
    if (append) sigmas_out[n_in] = 0.0f;
    *n_out = n;
    return FLUX2MI_OK;
}

// ------------------------------------------------------------------ latent utils (host) — Pipeline/LatentUtils.swift
extern "C" void flux2mi_validate_dimensions(int height, int width, int* out_height, int* out_width) {  // :382-394
    const int f = 16;
    if (out_height) *out_height = ((height + f - 1) / f) * f;
    if (out_width) *out_width = ((width + f - 1) / f) * f;
}

extern "C" void flux2mi_resolve_chain_dimensions(int width, int height, int max_pixels, int* out_height,
                                                 int* out_width) {  // +ChainHelpers.swift:112-129
    long w = width, h = height;
    const long px = w * h;
    if (max_pixels > 0 && px > max_pixels) {
        const double scale = sqrt((double)max_pixels / (double)px);
        w = (long)((double)w * scale);
        h = (long)((double)h * scale);
    }
    w = (w / 32) * 32;
    h = (h / 32) * 32;
    if (w < 32) w = 32;
    if (h < 32) h = 32;
    if (out_height) *out_height = (int)h;
    if (out_width) *out_width = (int)w;
}

extern "C" int flux2mi_image_position_ids(int height, int width, int32_t* out_host, int capacity_rows) {  // :256-286
    const int nh = height / 8 / 2, nw = width / 8 / 2;
    F2_REQUIRE(out_host && nh > 0 && nw > 0 && nh * nw <= capacity_rows, FLUX2MI_ERR_INVALID_ARG,
               "image_position_ids: %dx%d needs %d rows", height, width, nh * nw);
    for (int y = 0; y < nh; ++y)
        for (int x = 0; x < nw; ++x) {
            int32_t* r = out_host + ((size_t)y * nw + x) * 4;
            r[0] = 0; r[1] = y; r[2] = x; r[3] = 0;
        }
    return FLUX2MI_OK;
}

extern "C" int flux2mi_text_position_ids(int length, int32_t* out_host, int capacity_rows) {  // :291-298
    F2_REQUIRE(out_host && length > 0 && length <= capacity_rows, FLUX2MI_ERR_INVALID_ARG, "text_position_ids: %d",
               length);
    for (int l = 0; l < length; ++l) {
        int32_t* r = out_host + (size_t)l * 4;
        r[0] = 0; r[1] = 0; r[2] = 0; r[3] = l;
    }
    return FLUX2MI_OK;
}

extern "C" int flux2mi_reference_position_ids(const int* latent_heights, const int* latent_widths, int n_images,
                                              int scale, int32_t* out_host, int capacity_rows) {  // :324-346
    F2_REQUIRE(latent_heights && latent_widths && out_host && n_images > 0, FLUX2MI_ERR_INVALID_ARG,
               "reference_position_ids: bad args");
    long total = 0;
    for (int i = 0; i < n_images; ++i) total += (long)latent_heights[i] * latent_widths[i];
    F2_REQUIRE(total <= capacity_rows, FLUX2MI_ERR_INVALID_ARG, "reference_position_ids: need %ld rows", total);
    size_t row = 0;
    for (int i = 0; i < n_images; ++i) {
        const int32_t t = scale + scale * i;
        for (int y = 0; y < latent_heights[i]; ++y)
            for (int x = 0; x < latent_widths[i]; ++x) {
                int32_t* r = out_host + row * 4;
                r[0] = t; r[1] = y; r[2] = x; r[3] = 0;
                ++row;
            }
    }
    return FLUX2MI_OK;
}

// ------------------------------------------------------------------ MLX-compatible random stream
// (MLXRandom.seed / MLXRandom.normal as the reference calls them: LatentUtils.swift:32-41, Flux2Pipeline.swift:1203).
// Key bookkeeping is a handful of integer hashes and stays on the host; the draws run on the device.
namespace {
void host_threefry2x32(uint32_t k0, uint32_t k1, uint32_t& x0, uint32_t& x1) {
    static const int rot[2][4] = {{13, 15, 26, 6}, {17, 29, 16, 24}};
    const uint32_t ks[3] = {k0, k1, 0x1BD11BDAu ^ k0 ^ k1};
    x0 += ks[0];
    x1 += ks[1];
    for (int r = 0; r < 5; ++r) {
        for (int q = 0; q < 4; ++q) {
            x0 += x1;
            x1 = (x1 << rot[r & 1][q]) | (x1 >> (32 - rot[r & 1][q]));
            x1 ^= x0;
        }
        x0 += ks[(r + 1) % 3];
        x1 += ks[(r + 2) % 3] + (uint32_t)(r + 1);
    }
}
}  // namespace
extern "C" void flux2mi_mlx_random_key(uint64_t seed, uint32_t key[2]) {
    key[0] = (uint32_t)(seed >> 32);
    key[1] = (uint32_t)(seed & 0xFFFFFFFFull);
}
extern "C" void flux2mi_mlx_random_split(const uint32_t key[2], uint32_t first[2], uint32_t second[2]) {
    // bits([2,2]) with one key: words (0,2) from counter (0,2), words (1,3) from counter (1,3)
    uint32_t a0 = 0, a1 = 2, b0 = 1, b1 = 3;
    host_threefry2x32(key[0], key[1], a0, a1);
    host_threefry2x32(key[0], key[1], b0, b1);
    first[0] = a0, first[1] = b0, second[0] = a1, second[1] = b1;
}
extern "C" int flux2mi_mlx_random_bits(flux2mi_ctx* ctx, const uint32_t key[2], uint32_t* out, int64_t n) {
    F2_REQUIRE(ctx && key && out && n > 0 && n <= 0xFFFFFFFFll, FLUX2MI_ERR_INVALID_ARG, "mlx_random_bits: bad args");
    F2_LAUNCH(launch_mlx_random(ctx->stream, key[0], key[1], reinterpret_cast<float*>(out), n, 0, 0.f, 1.f), "mlx_random_bits");
    return FLUX2MI_OK;
}
extern "C" int flux2mi_mlx_random_uniform(flux2mi_ctx* ctx, const uint32_t key[2], float* out, int64_t n, float lo, float hi) {
    F2_REQUIRE(ctx && key && out && n > 0 && n <= 0xFFFFFFFFll && hi > lo, FLUX2MI_ERR_INVALID_ARG, "mlx_random_uniform: bad args");
    F2_LAUNCH(launch_mlx_random(ctx->stream, key[0], key[1], out, n, 1, lo, hi), "mlx_random_uniform");
    return FLUX2MI_OK;
}
extern "C" int flux2mi_mlx_random_normal(flux2mi_ctx* ctx, const uint32_t key[2], float* out, int64_t n) {
    F2_REQUIRE(ctx && key && out && n > 0 && n <= 0xFFFFFFFFll, FLUX2MI_ERR_INVALID_ARG, "mlx_random_normal: bad args");
    F2_LAUNCH(launch_mlx_random(ctx->stream, key[0], key[1], out, n, 2, -0.99999994f, 1.f), "mlx_random_normal");
    return FLUX2MI_OK;
}

// ------------------------------------------------------------------ elementwise device ops
extern "C" int flux2mi_sched_step(flux2mi_ctx* ctx, float* x, const float* v, float dt, int64_t n) {
    F2_REQUIRE(ctx && x && v && n > 0, FLUX2MI_ERR_INVALID_ARG, "sched_step: bad args");
    F2_LAUNCH(launch_axpy(ctx->stream, x, v, dt, n), "sched_step");
    return FLUX2MI_OK;
}
extern "C" int flux2mi_scale_noise(flux2mi_ctx* ctx, float* out, const float* sample, const float* noise,
                                   float sigma, int64_t n) {
    F2_REQUIRE(ctx && out && sample && noise && n > 0, FLUX2MI_ERR_INVALID_ARG, "scale_noise: bad args");
    F2_LAUNCH(launch_lerp(ctx->stream, out, sample, noise, sigma, n), "scale_noise");
    return FLUX2MI_OK;
}
extern "C" int flux2mi_cfg_combine(flux2mi_ctx* ctx, float* v_cond, const float* v_uncond, float guidance,
                                   int64_t n) {
    F2_REQUIRE(ctx && v_cond && v_uncond && n > 0, FLUX2MI_ERR_INVALID_ARG, "cfg_combine: bad args");
    F2_LAUNCH(launch_cfg_combine(ctx->stream, v_cond, v_uncond, guidance, n), "cfg_combine");
    return FLUX2MI_OK;
}
extern "C" int flux2mi_repaint_blend(flux2mi_ctx* ctx, float* x, const float* image_latents,
                                     const float* blend_noise, const float* mask, float sigma_next, int seq,
                                     int channels) {
    F2_REQUIRE(ctx && x && image_latents && blend_noise && mask, FLUX2MI_ERR_INVALID_ARG, "repaint_blend: bad args");
    F2_LAUNCH(launch_repaint_blend(ctx->stream, x, image_latents, blend_noise, mask, sigma_next, seq, channels),
              "repaint_blend");
    return FLUX2MI_OK;
}
extern "C" int flux2mi_pack_patchified_to_sequence(flux2mi_ctx* ctx, const float* patchified, float* seq, int c,
                                                   int h, int w) {
    F2_REQUIRE(ctx && patchified && seq, FLUX2MI_ERR_INVALID_ARG, "pack_patchified: bad args");
    F2_LAUNCH(launch_transpose_2d(ctx->stream, patchified, seq, c, h * w), "pack_patchified_to_sequence");
    return FLUX2MI_OK;
}
extern "C" int flux2mi_unpack_sequence_to_patchified(flux2mi_ctx* ctx, const float* seq, float* patchified, int c,
                                                     int h, int w) {
    F2_REQUIRE(ctx && patchified && seq, FLUX2MI_ERR_INVALID_ARG, "unpack_sequence: bad args");
    F2_LAUNCH(launch_transpose_2d(ctx->stream, seq, patchified, h * w, c), "unpack_sequence_to_patchified");
    return FLUX2MI_OK;
}
extern "C" int flux2mi_latents_to_vae_input(flux2mi_ctx* ctx, const float* packed, const float* bn_mean,
                                            const float* bn_var, float* z, int h16, int w16) {
    F2_REQUIRE(ctx && packed && bn_mean && bn_var && z, FLUX2MI_ERR_INVALID_ARG, "latents_to_vae_input: bad args");
    F2_LAUNCH(launch_latents_to_vae_input(ctx->stream, packed, bn_mean, bn_var, z, h16, w16), "latents_to_vae_input");
    return FLUX2MI_OK;
}
extern "C" int flux2mi_vae_latents_to_packed(flux2mi_ctx* ctx, const float* z, const float* bn_mean,
                                             const float* bn_var, float* packed, int h16, int w16) {
    F2_REQUIRE(ctx && packed && bn_mean && bn_var && z, FLUX2MI_ERR_INVALID_ARG, "vae_latents_to_packed: bad args");
    F2_LAUNCH(launch_vae_latents_to_packed(ctx->stream, z, bn_mean, bn_var, packed, h16, w16), "vae_latents_to_packed");
    return FLUX2MI_OK;
}
