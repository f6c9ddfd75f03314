// elementwise.hip — small / HBM-bound kernels: sampler arithmetic, layout permutes,
// timestep embedding, M=1 projections (modulation GEMVs), dtype conversion.
#include "dev.h"
#include "kernels.h"
#include "../../include/flux2mi.h"

namespace {

constexpr int EW_BLOCK = 256;
inline unsigned ew_grid(int64_t n_items) {
    int64_t b = (n_items + EW_BLOCK - 1) / EW_BLOCK;
    if (b > 256 * 16) b = 256 * 16;  // grid-stride the rest (guide G11)
    return (unsigned)(b < 1 ? 1 : b);
}

// ---- GEMV: out[n] (+)= sum_k W[n,k] * f(x[k]).  One wave per output row, 16 B/lane loads.
// M=1 projections are weight-bandwidth bound (time MLP, AdaLN modulations:
// Flux2Embeddings.swift:59-79, Flux2Modulation.swift:49-75).
__global__ __launch_bounds__(256) void gemv_kernel(const uint16_t* __restrict__ W, const float* __restrict__ x,
                                                   float* __restrict__ out, int N, int K, int silu_in,
                                                   int accumulate) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const uint16_t* wr = W + (size_t)n * K;
    float acc = 0.f;
    for (int k = lane * 8; k < K; k += 64 * 8) {
        const uint4 w = *reinterpret_cast<const uint4*>(wr + k);
        const float4 xa = *reinterpret_cast<const float4*>(x + k);
        const float4 xb = *reinterpret_cast<const float4*>(x + k + 4);
        float xv[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
        if (silu_in) {
#pragma unroll
            for (int j = 0; j < 8; ++j) xv[j] = silu_f(xv[j]);
        }
        acc += lo_bf16(w.x) * xv[0] + hi_bf16(w.x) * xv[1] + lo_bf16(w.y) * xv[2] + hi_bf16(w.y) * xv[3] +
               lo_bf16(w.z) * xv[4] + hi_bf16(w.z) * xv[5] + lo_bf16(w.w) * xv[6] + hi_bf16(w.w) * xv[7];
    }
    acc = wave_sum(acc);
    if (lane == 0) out[n] = accumulate ? out[n] + acc : acc;
}

// The same for up to GEMV_MULTI vectors at once (x_v = x + v xs, out_v = out + v os): the weights are read once, every vector goes through
// the single-vector kernel's arithmetic in its order (same per-lane chunks, same expression, same wave reduction): the results are the
// single-vector kernel's.  For flux2mi_generate: the timestep MLP and the modulation projections of ALL steps in one pass.
constexpr int GEMV_MULTI = 32;
__global__ __launch_bounds__(256) void gemv_multi_kernel(const uint16_t* __restrict__ W, const float* __restrict__ x, long xs,
                                                         float* __restrict__ out, long os, int N, int K, int V, int silu_in, int accumulate) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const uint16_t* wr = W + (size_t)n * K;
    float acc[GEMV_MULTI];
#pragma unroll
    for (int v = 0; v < GEMV_MULTI; ++v) acc[v] = 0.f;
    for (int k = lane * 8; k < K; k += 64 * 8) {
        const uint4 w = *reinterpret_cast<const uint4*>(wr + k);
#pragma unroll
        for (int v = 0; v < GEMV_MULTI; ++v) {
            if (v < V) {
                const float4 xa = *reinterpret_cast<const float4*>(x + v * xs + k);
                const float4 xb = *reinterpret_cast<const float4*>(x + v * xs + k + 4);
                float xv[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
                if (silu_in) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) xv[j] = silu_f(xv[j]);
                }
                acc[v] += lo_bf16(w.x) * xv[0] + hi_bf16(w.x) * xv[1] + lo_bf16(w.y) * xv[2] + hi_bf16(w.y) * xv[3] +
                          lo_bf16(w.z) * xv[4] + hi_bf16(w.z) * xv[5] + lo_bf16(w.w) * xv[6] + hi_bf16(w.w) * xv[7];
            }
        }
    }
#pragma unroll
    for (int v = 0; v < GEMV_MULTI; ++v) {
        if (v < V) {
            const float a = wave_sum(acc[v]);
            if (lane == 0) out[v * os + n] = accumulate ? out[v * os + n] + a : a;
        }
    }
}

// ---- sinusoidal embedding: [cos(v f_i) | sin(v f_i)], f_i = exp(-ln(1e4) i/128), i < 128 (Flux2Embeddings.swift:29-40).
// Evaluated in f64 and rounded once: the argument reaches 1000 rad, where an f32 evaluation carries ~4e-5 of argument error (one ulp of
// f_i times the timestep) whatever the libm — the exact value is what every f32 evaluation, the reference's included, approximates, and
// a timestep MLP amplifies the difference (synthetic weights: 3e-3 of the output).  128 threads, once per step: no cost.
__global__ void sinusoid_kernel(float value, float* __restrict__ emb) {
    const int i = threadIdx.x;  // 128 threads
    const double f = exp(-9.210340371976184 * (double)i / 128.0);
    const double a = (double)value * f;
    emb[i] = (float)cos(a);
    emb[128 + i] = (float)sin(a);
}

__global__ void silu_inplace_kernel(float* x, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) x[i] = silu_f(x[i]);
}

// ---- dtype conversion (weights arrive as f32 / bf16 / f16)
__global__ void to_bf16_kernel(const void* __restrict__ in, int dtype, uint16_t* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float v;
        if (dtype == FLUX2MI_F32) v = reinterpret_cast<const float*>(in)[i];
        else if (dtype == FLUX2MI_F16) v = (float)reinterpret_cast<const _Float16*>(in)[i];
        else { out[i] = reinterpret_cast<const uint16_t*>(in)[i]; continue; }
        out[i] = f32_to_bf16_bits(v);
    }
}
__global__ void to_f32_kernel(const void* __restrict__ in, int dtype, float* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float v;
        if (dtype == FLUX2MI_F32) v = reinterpret_cast<const float*>(in)[i];
        else if (dtype == FLUX2MI_F16) v = (float)reinterpret_cast<const _Float16*>(in)[i];
        else v = bf16_bits_to_f32(reinterpret_cast<const uint16_t*>(in)[i]);
        out[i] = v;
    }
}

// ---- SwiGLU row interleave: out[64b + j] = in[32b + j], out[64b + 32 + j] = in[hidden + 32b + j]
__global__ void swiglu_interleave_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, int hidden,
                                         int K) {
    const int r = blockIdx.x;  // output row
    const int b = r >> 6, j = r & 63;
    const int src = (j < 32) ? (b * 32 + j) : (hidden + b * 32 + (j - 32));
    const uint4* s = reinterpret_cast<const uint4*>(in + (size_t)src * K);
    uint4* d = reinterpret_cast<uint4*>(out + (size_t)r * K);
    for (int c = threadIdx.x; c < K / 8; c += blockDim.x) d[c] = s[c];
}

// ---- MLX-compatible random stream (MLXRandom.seed + MLXRandom.normal, LatentUtils.swift:32-41).  The generator is
// MLX's: Threefry-2x32 (20 rounds) keyed by a uint32 pair; word i and word half+i of a draw are the two outputs of
// the counter (i, half+i), an odd count takes its middle word from counter (half, 0).  HBM-bound: 4 B written per
// element, ~100 integer ops — one counter per thread, grid-stride.
__device__ __forceinline__ void threefry2x32(uint32_t k0, uint32_t k1, uint32_t& x0, uint32_t& x1) {
    const uint32_t ks[3] = {k0, k1, 0x1BD11BDAu ^ k0 ^ k1};
    x0 += ks[0];
    x1 += ks[1];
#pragma unroll
    for (int r = 0; r < 5; ++r) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int rot = (r & 1) ? (q == 0 ? 17 : q == 1 ? 29 : q == 2 ? 16 : 24) : (q == 0 ? 13 : q == 1 ? 15 : q == 2 ? 26 : 6);
            x0 += x1;
            x1 = (x1 << rot) | (x1 >> (32 - rot));
            x1 ^= x0;
        }
        x0 += ks[(r + 1) % 3];
        x1 += ks[(r + 2) % 3] + (uint32_t)(r + 1);
    }
}
// mode 0: raw words (uint32 bit pattern), 1: uniform [lo, hi), 2: standard normal
__device__ __forceinline__ float mlx_transform(uint32_t w, int mode, float lo, float range) {
#pragma clang fp contract(off)  // multiply and add are two MLX ops with two roundings (plain operators: the __fmul_rn /
    // __fadd_rn header inlines still get fused); the explicit fmaf calls below stay fused
    if (mode == 0) return __uint_as_float(w);
    float u = (float)w * 2.3283064365386963e-10f;  // / 2^32, exact
    u = fminf(u, 0.99999994f);
    const float ru = range * u;
    u = ru + lo;  // multiply, then add: two roundings as two MLX ops
    if (mode == 1) return u;
    // erfinv, single-precision polynomial pair in t = log(1 - u^2)
    float t = logf(fmaf(u, -u, 1.0f));
    float p;
    if (fabsf(t) > 6.125f) {
        p = 3.03697567e-10f;
        p = fmaf(p, t, 2.93243101e-8f);
        p = fmaf(p, t, 1.22150334e-6f);
        p = fmaf(p, t, 2.84108955e-5f);
        p = fmaf(p, t, 3.93552968e-4f);
        p = fmaf(p, t, 3.02698812e-3f);
        p = fmaf(p, t, 4.83185798e-3f);
        p = fmaf(p, t, -2.64646143e-1f);
        p = fmaf(p, t, 8.40016484e-1f);
    } else {
        p = 5.43877832e-9f;
        p = fmaf(p, t, 1.43285448e-7f);
        p = fmaf(p, t, 1.22774793e-6f);
        p = fmaf(p, t, 1.12963626e-7f);
        p = fmaf(p, t, -5.61530760e-5f);
        p = fmaf(p, t, -1.47697632e-4f);
        p = fmaf(p, t, 2.31468678e-3f);
        p = fmaf(p, t, 1.15392581e-2f);
        p = fmaf(p, t, -2.32015476e-1f);
        p = fmaf(p, t, 8.86226892e-1f);
    }
    return 1.41421356f * (u * p);
}
__global__ void mlx_random_kernel(uint32_t k0, uint32_t k1, float* __restrict__ out, int64_t n, int mode, float lo, float range) {
    const int64_t half = n >> 1;
    const int64_t odd = n & 1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < half + odd; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t x0 = (uint32_t)i, x1 = (i < half) ? (uint32_t)(i + half + odd) : 0u;
        threefry2x32(k0, k1, x0, x1);
        if (i < half) {
            out[i] = mlx_transform(x0, mode, lo, range);
            out[i + half + odd] = mlx_transform(x1, mode, lo, range);
        } else {
            out[half] = mlx_transform(x0, mode, lo, range);
        }
    }
}

__global__ void bf16_add_kernel(uint16_t* __restrict__ w, const uint16_t* __restrict__ dl, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        w[i] = f32_to_bf16_bits(bf16_bits_to_f32(w[i]) + bf16_bits_to_f32(dl[i]));
}
__global__ void dequant_i8_rows_kernel(const int8_t* __restrict__ data, const float* __restrict__ scale,
                                       uint16_t* __restrict__ out, int rows, int cols) {
    const int64_t n = (int64_t)rows * cols;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        out[i] = f32_to_bf16_bits((float)data[i] * scale[i / cols]);
}

__global__ void embed_rows_kernel(const int32_t* __restrict__ ids, const uint16_t* __restrict__ table,
                                  float* __restrict__ hs, int dim, int vocab) {
    const int row = blockIdx.x;
    int id = ids[row];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);   // ids are validated on the host side of the C ABI
    const uint16_t* src = table + (size_t)id * dim;
    for (int c = threadIdx.x; c < dim; c += blockDim.x) hs[(size_t)row * dim + c] = bf16_bits_to_f32(src[c]);
}
__global__ void f32_rows_to_bf16_cols_kernel(const float* __restrict__ hs, uint16_t* __restrict__ out, int dim, int ldo,
                                             int c0) {
    const int row = blockIdx.x;
    for (int c = threadIdx.x; c < dim; c += blockDim.x)
        out[(size_t)row * ldo + c0 + c] = f32_to_bf16_bits(hs[(size_t)row * dim + c]);
}

// ---- sampler arithmetic (f32, 16 B/lane)
__global__ void axpy_kernel(float* __restrict__ x, const float* __restrict__ v, float a, int64_t n4, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 xv = reinterpret_cast<float4*>(x)[i];
        const float4 vv = reinterpret_cast<const float4*>(v)[i];
        xv.x += a * vv.x; xv.y += a * vv.y; xv.z += a * vv.z; xv.w += a * vv.w;
        reinterpret_cast<float4*>(x)[i] = xv;
    }
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) x[i] += a * v[i];
}
__global__ void lerp_kernel(float* __restrict__ out, const float* __restrict__ a, const float* __restrict__ b,
                            float t, int64_t n) {
    const float u = 1.0f - t;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = u * a[i] + t * b[i];
}
__global__ void cfg_kernel(float* __restrict__ vc, const float* __restrict__ vu, float g, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        vc[i] = vu[i] + g * (vc[i] - vu[i]);
}
// ---- MLXArray façade: out = a (op) b with numpy-style broadcasting over <= 4 dims (array.cpp).  HBM-bound: one thread
// per 4 consecutive output elements of the last dim when both operands are contiguous or broadcast there (16-byte
// accesses), else one per element.  Each op is its own kernel, so every product and sum is rounded separately to f32
// exactly like the un-fused MLX graph (no FMA contraction across ops).
struct BcastArgs {
    int64_t n;            // output elements
    int d[4];             // output dims (padded with leading 1s)
    int64_t sa[4], sb[4]; // element strides of a / b per output dim (0 = broadcast)
};
template <int OP> __device__ __forceinline__ float bc_op(float x, float y) {
    return OP == 0 ? x + y : OP == 1 ? x - y : x * y;
}
template <int OP, bool VEC>
__global__ void bcast_binary_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, BcastArgs g) {
    const int64_t step = VEC ? 4 : 1;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * step; i < g.n; i += (int64_t)gridDim.x * blockDim.x * step) {
        int64_t r = i;
        const int i3 = (int)(r % g.d[3]); r /= g.d[3];
        const int i2 = (int)(r % g.d[2]); r /= g.d[2];
        const int i1 = (int)(r % g.d[1]); r /= g.d[1];
        const int64_t oa = r * g.sa[0] + i1 * g.sa[1] + i2 * g.sa[2] + i3 * g.sa[3];
        const int64_t ob = r * g.sb[0] + i1 * g.sb[1] + i2 * g.sb[2] + i3 * g.sb[3];
        if (VEC) {
            const f32x4 x = g.sa[3] ? *reinterpret_cast<const f32x4*>(a + oa) : f32x4{a[oa], a[oa], a[oa], a[oa]};
            const f32x4 y = g.sb[3] ? *reinterpret_cast<const f32x4*>(b + ob) : f32x4{b[ob], b[ob], b[ob], b[ob]};
            f32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = bc_op<OP>(x[k], y[k]);
            *reinterpret_cast<f32x4*>(out + i) = o;
        } else {
            out[i] = bc_op<OP>(a[oa], b[ob]);
        }
    }
}

// RePaint: x = (1-m)*((1-s)*img + s*noise) + m*x, m [S,1].  No FMA contraction: every product and sum is rounded on its
// own, so that the fused kernel is bit-identical to the reference's closure executed op by op
// (Flux2MaskedInpaintingChain.swift:399-403 through the flux2mi_array_* façade).
#pragma clang fp contract(off)
__global__ void repaint_kernel(float* __restrict__ x, const float* __restrict__ img, const float* __restrict__ noise,
                               const float* __restrict__ mask, float s, int64_t n, int c) {
    const float u = 1.0f - s;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float m = mask[i / c];
        const float orig = u * img[i] + s * noise[i];
        x[i] = (1.0f - m) * orig + m * x[i];
    }
}
#pragma clang fp contract(fast)

// ---- [rows, cols] f32 -> [cols, rows] (pack / unpack of patchified latents)
__global__ void transpose_f32_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int cols) {
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const int r = by + j, c = bx + threadIdx.x;
        if (r < rows && c < cols) tile[j][threadIdx.x] = in[(size_t)r * cols + c];
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const int c = bx + j, r = by + threadIdx.x;
        if (r < rows && c < cols) out[(size_t)c * rows + r] = tile[threadIdx.x][j];
    }
}

// ---- packed [h16*w16,128] -> BN-denormalise -> unpatchify -> z [32, 2*h16, 2*w16]
// channel index of the patchified tensor = c*4 + py*2 + px (LatentUtils.swift:119-142);
// x*sqrt(var+1e-4)+mean (LatentUtils.swift:483-496).
__global__ void latents_to_vae_kernel(const float* __restrict__ packed, const float* __restrict__ mean,
                                      const float* __restrict__ var, float* __restrict__ z, int h16, int w16) {
    const int total = h16 * w16 * 128;
    const int H = 2 * h16, W = 2 * w16;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int pc = i & 127, tok = i >> 7;
        const int ty = tok / w16, tx = tok % w16;
        const int c = pc >> 2, py = (pc >> 1) & 1, px = pc & 1;
        const float v = packed[i] * sqrtf(var[pc] + 1e-4f) + mean[pc];
        z[((size_t)c * H + (2 * ty + py)) * W + 2 * tx + px] = v;
    }
}
__global__ void vae_to_packed_kernel(const float* __restrict__ z, const float* __restrict__ mean,
                                     const float* __restrict__ var, float* __restrict__ packed, int h16, int w16) {
    const int total = h16 * w16 * 128;
    const int H = 2 * h16, W = 2 * w16;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int pc = i & 127, tok = i >> 7;
        const int ty = tok / w16, tx = tok % w16;
        const int c = pc >> 2, py = (pc >> 1) & 1, px = pc & 1;
        const float v = z[((size_t)c * H + (2 * ty + py)) * W + 2 * tx + px];
        packed[i] = (v - mean[pc]) / sqrtf(var[pc] + 1e-4f);
    }
}

}  // namespace

#define EW_DONE() return hipGetLastError() == hipSuccess ? 0 : 2

int launch_gemv(hipStream_t stream, const uint16_t* W, const float* x, float* out, int N, int K, int silu_in,
                int accumulate) {
    if (N <= 0 || K <= 0 || (K & 7)) return 1;
    hipLaunchKernelGGL(gemv_kernel, dim3((N + 3) / 4), dim3(256), 0, stream, W, x, out, N, K, silu_in, accumulate);
    EW_DONE();
}
int launch_gemv_multi(hipStream_t stream, const uint16_t* W, const float* x, long xs, float* out, long os, int N, int K, int V, int silu_in,
                      int accumulate) {
    if (N <= 0 || K <= 0 || (K & 7) || V <= 0 || (xs & 3) || V > GEMV_MULTI) return 1;
    hipLaunchKernelGGL(gemv_multi_kernel, dim3((N + 3) / 4), dim3(256), 0, stream, W, x, xs, out, os, N, K, V, silu_in, accumulate);
    EW_DONE();
}
int launch_sinusoid(hipStream_t stream, float value, float* emb) {
    hipLaunchKernelGGL(sinusoid_kernel, dim3(1), dim3(128), 0, stream, value, emb);
    EW_DONE();
}
int launch_silu_inplace(hipStream_t stream, float* x, int n) {
    hipLaunchKernelGGL(silu_inplace_kernel, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, stream, x, n);
    EW_DONE();
}
int launch_f32_to_bf16(hipStream_t stream, const float* in, uint16_t* out, int64_t n) {
    return launch_convert_to_bf16(stream, in, FLUX2MI_F32, out, n);
}
int launch_convert_to_bf16(hipStream_t stream, const void* in, int dtype, uint16_t* out, int64_t n) {
    if (n <= 0) return 1;
    hipLaunchKernelGGL(to_bf16_kernel, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, stream, in, dtype, out, n);
    EW_DONE();
}
int launch_convert_to_f32(hipStream_t stream, const void* in, int dtype, float* out, int64_t n) {
    if (n <= 0) return 1;
    hipLaunchKernelGGL(to_f32_kernel, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, stream, in, dtype, out, n);
    EW_DONE();
}
int launch_swiglu_interleave_rows(hipStream_t stream, const uint16_t* in, uint16_t* out, int hidden, int K) {
    if (hidden <= 0 || (hidden & 31) || (K & 7)) return 1;
    hipLaunchKernelGGL(swiglu_interleave_kernel, dim3(2 * hidden), dim3(256), 0, stream, in, out, hidden, K);
    EW_DONE();
}
int launch_embed_rows(hipStream_t stream, const int32_t* ids, const uint16_t* table, float* hs, int S, int dim, int vocab) {
    if (S <= 0 || dim <= 0 || vocab <= 0) return 1;
    hipLaunchKernelGGL(embed_rows_kernel, dim3(S), dim3(256), 0, stream, ids, table, hs, dim, vocab);
    EW_DONE();
}
int launch_f32_rows_to_bf16_cols(hipStream_t stream, const float* hs, uint16_t* out, int S, int dim, int ldo, int c0) {
    if (S <= 0 || dim <= 0 || ldo < c0 + dim) return 1;
    hipLaunchKernelGGL(f32_rows_to_bf16_cols_kernel, dim3(S), dim3(256), 0, stream, hs, out, dim, ldo, c0);
    EW_DONE();
}
int launch_mlx_random(hipStream_t stream, uint32_t k0, uint32_t k1, float* out, int64_t n, int mode, float lo, float hi) {
    if (n <= 0 || n > 0xFFFFFFFFll || mode < 0 || mode > 2) return 1;
    const float range = hi - lo;
    hipLaunchKernelGGL(mlx_random_kernel, dim3(ew_grid((n + 1) / 2)), dim3(EW_BLOCK), 0, stream, k0, k1, out, n, mode, lo, range);
    EW_DONE();
}
int launch_bf16_add_inplace(hipStream_t stream, uint16_t* w, const uint16_t* delta, int64_t n) {
    if (n <= 0) return 1;
    hipLaunchKernelGGL(bf16_add_kernel, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, stream, w, delta, n);
    EW_DONE();
}
int launch_dequant_i8_rows(hipStream_t stream, const int8_t* data, const float* scale, uint16_t* out, int rows, int cols) {
    if (rows <= 0 || cols <= 0) return 1;
    hipLaunchKernelGGL(dequant_i8_rows_kernel, dim3(ew_grid((int64_t)rows * cols)), dim3(EW_BLOCK), 0, stream, data, scale, out,
                       rows, cols);
    EW_DONE();
}
int launch_axpy(hipStream_t stream, float* x, const float* v, float a, int64_t n) {
    if (n <= 0) return 1;
    const bool al = (((uintptr_t)x | (uintptr_t)v) & 15) == 0;
    const int64_t n4 = al ? n / 4 : 0;
    hipLaunchKernelGGL(axpy_kernel, dim3(ew_grid(n4 > 0 ? n4 : n)), dim3(EW_BLOCK), 0, stream, x, v, a, n4, n);
    EW_DONE();
}
int launch_lerp(hipStream_t stream, float* out, const float* a, const float* b, float t, int64_t n) {
    if (n <= 0) return 1;
    hipLaunchKernelGGL(lerp_kernel, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, stream, out, a, b, t, n);
    EW_DONE();
}
int launch_cfg_combine(hipStream_t stream, float* vc, const float* vu, float g, int64_t n) {
    if (n <= 0) return 1;
    hipLaunchKernelGGL(cfg_kernel, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, stream, vc, vu, g, n);
    EW_DONE();
}
int launch_repaint_blend(hipStream_t stream, float* x, const float* img, const float* noise, const float* mask,
                         float sigma_next, int seq, int c) {
    if (seq <= 0 || c <= 0) return 1;
    const int64_t n = (int64_t)seq * c;
    hipLaunchKernelGGL(repaint_kernel, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, stream, x, img, noise, mask, sigma_next,
                       n, c);
    EW_DONE();
}
int launch_bcast_binary(hipStream_t stream, int op, const float* a, const int64_t* sa, const float* b, const int64_t* sb, float* out,
                        const int* dims) {
    BcastArgs g;
    g.n = 1;
    for (int i = 0; i < 4; ++i) { g.d[i] = dims[i]; g.sa[i] = sa[i]; g.sb[i] = sb[i]; g.n *= dims[i]; }
    if (g.n <= 0 || op < 0 || op > 2) return 1;
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    // 16-byte form: last dim a multiple of 4, each operand contiguous (stride 1, every outer stride a multiple of 4) or broadcast there
    bool vec = dims[3] % 4 == 0 && al16(a) && al16(b) && al16(out) && (sa[3] == 1 || sa[3] == 0) && (sb[3] == 1 || sb[3] == 0);
    for (int i = 0; i < 3; ++i) vec = vec && (sa[3] == 0 || sa[i] % 4 == 0) && (sb[3] == 0 || sb[i] % 4 == 0);
    const unsigned grid = ew_grid(vec ? (g.n + 3) / 4 : g.n);
#define BC_GO(OP_) if (vec) hipLaunchKernelGGL((bcast_binary_kernel<OP_, true>), dim3(grid), dim3(EW_BLOCK), 0, stream, a, b, out, g); \
                   else hipLaunchKernelGGL((bcast_binary_kernel<OP_, false>), dim3(grid), dim3(EW_BLOCK), 0, stream, a, b, out, g);
    if (op == 0) { BC_GO(0) } else if (op == 1) { BC_GO(1) } else { BC_GO(2) }
#undef BC_GO
    EW_DONE();
}
int launch_transpose_2d(hipStream_t stream, const float* in, float* out, int rows, int cols) {
    if (rows <= 0 || cols <= 0) return 1;
    hipLaunchKernelGGL(transpose_f32_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(32, 8), 0, stream, in,
                       out, rows, cols);
    EW_DONE();
}
int launch_latents_to_vae_input(hipStream_t stream, const float* packed, const float* mean, const float* var,
                                float* z, int h16, int w16) {
    if (h16 <= 0 || w16 <= 0) return 1;
    hipLaunchKernelGGL(latents_to_vae_kernel, dim3(ew_grid((int64_t)h16 * w16 * 128)), dim3(EW_BLOCK), 0, stream,
                       packed, mean, var, z, h16, w16);
    EW_DONE();
}
int launch_vae_latents_to_packed(hipStream_t stream, const float* z, const float* mean, const float* var,
                                 float* packed, int h16, int w16) {
    if (h16 <= 0 || w16 <= 0) return 1;
    hipLaunchKernelGGL(vae_to_packed_kernel, dim3(ew_grid((int64_t)h16 * w16 * 128)), dim3(EW_BLOCK), 0, stream, z,
                       mean, var, packed, h16, w16);
    EW_DONE();
}
