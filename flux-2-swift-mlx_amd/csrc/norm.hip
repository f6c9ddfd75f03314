// norm.hip — HBM-bound normalisation kernels of the DiT forward (gfx950).
//   * LayerNorm(no affine, eps 1e-6) + AdaLN modulate -> bf16 GEMM operand
//     (Flux2TransformerBlock.swift:95-106, Flux2Modulation.swift:96-112)
//   * per-head RMSNorm + RoPE on Q,K in place; V -> transposed, key-permuted V^T
//     (Flux2Attention.swift:137-157, 212-229, 442-461)
//   * RoPE cos/sin tables from position ids (Flux2RoPE.swift:129-169)
// All loads/stores are 8-16 B per lane on contiguous rows.
#include "dev.h"
#include "kernels.h"

namespace {

// ------------------------------------------------------------------ LayerNorm + modulate
// One 256-thread workgroup per row; the row (dim <= 8192 f32) lives in registers.
constexpr int LN_MAX_VEC = 8;  // 8 float4 per thread -> dim <= 8192

__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void layernorm_modulate_kernel(const float* __restrict__ x,
                                                                 uint16_t* __restrict__ y, int dim, int split,
                                                                 const float* __restrict__ shift0,
                                                                 const float* __restrict__ scale0,
                                                                 const float* __restrict__ shift1,
                                                                 const float* __restrict__ scale1) {
    __shared__ float red[4];
    const int row = blockIdx.x;
    const float* xr = x + (size_t)row * dim;
    const int nvec = dim >> 2;
    float4 v[LN_MAX_VEC];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_VEC; ++i) {
        const int c = threadIdx.x + 256 * i;
        if (c < nvec) {
            v[i] = reinterpret_cast<const float4*>(xr)[c];
            s += v[i].x + v[i].y + v[i].z + v[i].w;
        }
    }
    const float mean = block_sum_256(s, red) / (float)dim;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_VEC; ++i) {
        const int c = threadIdx.x + 256 * i;
        if (c < nvec) {
            const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
            sq += a * a + b * b + cc * cc + d * d;
        }
    }
    const float var = block_sum_256(sq, red) / (float)dim;
    const float rstd = rsqrtf(var + 1e-6f);
    const float* sh = (row < split) ? shift0 : shift1;
    const float* sc = (row < split) ? scale0 : scale1;
    uint16_t* yr = y + (size_t)row * dim;
#pragma unroll
    for (int i = 0; i < LN_MAX_VEC; ++i) {
        const int c = threadIdx.x + 256 * i;
        if (c < nvec) {
            const float4 h = reinterpret_cast<const float4*>(sh)[c];
            const float4 g = reinterpret_cast<const float4*>(sc)[c];
            const float o0 = (v[i].x - mean) * rstd * (1.f + g.x) + h.x;
            const float o1 = (v[i].y - mean) * rstd * (1.f + g.y) + h.y;
            const float o2 = (v[i].z - mean) * rstd * (1.f + g.z) + h.z;
            const float o3 = (v[i].w - mean) * rstd * (1.f + g.w) + h.w;
            uint2 pk;
            pk.x = pack_bf16x2(o0, o1);
            pk.y = pack_bf16x2(o2, o3);
            reinterpret_cast<uint2*>(yr)[c] = pk;
        }
    }
}

// ------------------------------------------------------------------ LayerNorm + modulate -> fp8 + row scale
__device__ __forceinline__ float block_max_256(float v, float* red) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

__global__ __launch_bounds__(256) void layernorm_modulate_fp8_kernel(const float* __restrict__ x,
                                                                     uint8_t* __restrict__ y, float* __restrict__ ys,
                                                                     int dim, int split,
                                                                     const float* __restrict__ shift0,
                                                                     const float* __restrict__ scale0,
                                                                     const float* __restrict__ shift1,
                                                                     const float* __restrict__ scale1) {
    __shared__ float red[4];
    const int row = blockIdx.x;
    const float* xr = x + (size_t)row * dim;
    const int nvec = dim >> 2;
    float4 v[LN_MAX_VEC];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_VEC; ++i) {
        const int c = threadIdx.x + 256 * i;
        if (c < nvec) {
            v[i] = reinterpret_cast<const float4*>(xr)[c];
            s += v[i].x + v[i].y + v[i].z + v[i].w;
        }
    }
    const float mean = block_sum_256(s, red) / (float)dim;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_VEC; ++i) {
        const int c = threadIdx.x + 256 * i;
        if (c < nvec) {
            const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
            sq += a * a + b * b + cc * cc + d * d;
        }
    }
    const float var = block_sum_256(sq, red) / (float)dim;
    const float rstd = rsqrtf(var + 1e-6f);
    const float* sh = (row < split) ? shift0 : shift1;
    const float* sc = (row < split) ? scale0 : scale1;
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_VEC; ++i) {
        const int c = threadIdx.x + 256 * i;
        if (c < nvec) {
            const float4 h = reinterpret_cast<const float4*>(sh)[c];
            const float4 g = reinterpret_cast<const float4*>(sc)[c];
            // the bf16 path rounds these values to bf16; quantise the same bf16-rounded values
            v[i].x = bf16_bits_to_f32(f32_to_bf16_bits((v[i].x - mean) * rstd * (1.f + g.x) + h.x));
            v[i].y = bf16_bits_to_f32(f32_to_bf16_bits((v[i].y - mean) * rstd * (1.f + g.y) + h.y));
            v[i].z = bf16_bits_to_f32(f32_to_bf16_bits((v[i].z - mean) * rstd * (1.f + g.z) + h.z));
            v[i].w = bf16_bits_to_f32(f32_to_bf16_bits((v[i].w - mean) * rstd * (1.f + g.w) + h.w));
            amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[i].x), fabsf(v[i].y)), fmaxf(fabsf(v[i].z), fabsf(v[i].w))));
        }
    }
    amax = block_max_256(amax, red);
    const float qs = fmaxf(amax, 1e-12f) * (1.0f / 448.0f);
    const float inv = 1.0f / qs;
    if (threadIdx.x == 0) ys[row] = qs;
    uint32_t* yr = reinterpret_cast<uint32_t*>(y + (size_t)row * dim);
#pragma unroll
    for (int i = 0; i < LN_MAX_VEC; ++i) {
        const int c = threadIdx.x + 256 * i;
        if (c < nvec) yr[c] = pack_fp8x4(v[i].x * inv, v[i].y * inv, v[i].z * inv, v[i].w * inv);
    }
}

// Row-wise dynamic quantisation of a bf16 matrix (weights at load time, attention / SwiGLU outputs
// in front of the next fp8 GEMM): one workgroup per row, two passes over the row (second from L2).
__global__ __launch_bounds__(256) void quantize_rows_fp8_kernel(const uint16_t* __restrict__ x, int ld,
                                                                uint8_t* __restrict__ q, int ldq,
                                                                float* __restrict__ scale, int cols) {
    __shared__ float red[4];
    const int row = blockIdx.x;
    const uint4* xr = reinterpret_cast<const uint4*>(x + (size_t)row * ld);
    const int nvec = cols >> 3;
    float amax = 0.f;
    for (int c = threadIdx.x; c < nvec; c += 256) {
        const uint4 u = xr[c];
        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(lo_bf16(u.x)), fabsf(hi_bf16(u.x))), fmaxf(fabsf(lo_bf16(u.y)), fabsf(hi_bf16(u.y)))));
        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(lo_bf16(u.z)), fabsf(hi_bf16(u.z))), fmaxf(fabsf(lo_bf16(u.w)), fabsf(hi_bf16(u.w)))));
    }
    amax = block_max_256(amax, red);
    const float qs = fmaxf(amax, 1e-12f) * (1.0f / 448.0f);
    const float inv = 1.0f / qs;
    if (threadIdx.x == 0) scale[row] = qs;
    uint2* qr = reinterpret_cast<uint2*>(q + (size_t)row * ldq);
    for (int c = threadIdx.x; c < nvec; c += 256) {
        const uint4 u = xr[c];
        uint2 o;
        o.x = pack_fp8x4(lo_bf16(u.x) * inv, hi_bf16(u.x) * inv, lo_bf16(u.y) * inv, hi_bf16(u.y) * inv);
        o.y = pack_fp8x4(lo_bf16(u.z) * inv, hi_bf16(u.z) * inv, lo_bf16(u.w) * inv, hi_bf16(u.w) * inv);
        qr[c] = o;
    }
}

// Block quantisation of a bf16 matrix (GemmProblem::a_bscale): one workgroup per 16 rows x 128 columns, thread = (row, 8-column chunk).
// The stand-alone form of what the SwiGLU and attention epilogues do in place; rows beyond `rows` count as zeros.
__global__ __launch_bounds__(256) void quantize_blocks_fp8_kernel(const uint16_t* __restrict__ x, int ld, uint8_t* __restrict__ q, int ldq,
                                                                  uint8_t* __restrict__ bscale, int lds, int rows) {
    __shared__ float red[4];
    const int kb = blockIdx.x, rb = blockIdx.y;
    const int row = rb * 16 + (threadIdx.x >> 4), c0 = kb * 128 + (threadIdx.x & 15) * 8;
    uint4 u = make_uint4(0u, 0u, 0u, 0u);
    if (row < rows) u = *reinterpret_cast<const uint4*>(x + (size_t)row * ld + c0);
    const float f[8] = {lo_bf16(u.x), hi_bf16(u.x), lo_bf16(u.y), hi_bf16(u.y), lo_bf16(u.z), hi_bf16(u.z), lo_bf16(u.w), hi_bf16(u.w)};
    float amax = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(f[j]));
    amax = block_max_256(amax, red);
    const int e = fp8_block_exp(amax);
    const float m = ldexpf(1.0f, e);
    if (threadIdx.x == 0) bscale[(size_t)kb * lds + rb] = (uint8_t)(127 - e);
    if (row < rows) {
        uint2 o;
        o.x = pack_fp8x4(f[0] * m, f[1] * m, f[2] * m, f[3] * m);
        o.y = pack_fp8x4(f[4] * m, f[5] * m, f[6] * m, f[7] * m);
        *reinterpret_cast<uint2*>(q + (size_t)row * ldq + c0) = o;
    }
}

// ------------------------------------------------------------------ QK RMSNorm + RoPE (in place)
// 16 lanes per (row, head, q|k) vector of 128: a lane owns 8 consecutive elements = 4 RoPE
// pairs (16-B accesses); a wave handles 4 vectors, a workgroup 16.  Sum of squares by a
// 4-step xor-shuffle inside each 16-lane group.
__global__ __launch_bounds__(256) void qk_norm_rope_kernel(uint16_t* __restrict__ qkv, int ldqkv, int S, int heads,
                                                           int split, const uint16_t* __restrict__ qs0,
                                                           const uint16_t* __restrict__ ks0,
                                                           const uint16_t* __restrict__ qs1,
                                                           const uint16_t* __restrict__ ks1,
                                                           const float* __restrict__ cos_tab,
                                                           const float* __restrict__ sin_tab, float q_mul,
                                                           uint8_t* __restrict__ qk8, float q8_mul, float k8_mul) {
    const int sub = threadIdx.x & 15;
    const long vec = (long)blockIdx.x * 16 + (threadIdx.x >> 4);  // ((row*2 + which)*heads + head)
    const long total = (long)S * 2 * heads;
    const bool live = vec < total;
    const long v = live ? vec : total - 1;
    const int head = (int)(v % heads);
    const int which = (int)((v / heads) & 1);
    const int row = (int)(v / (2 * heads));
    uint16_t* p = qkv + (size_t)row * ldqkv + which * heads * 128 + head * 128 + sub * 8;
    const uint4 u = *reinterpret_cast<const uint4*>(p);
    float x[8] = {lo_bf16(u.x), hi_bf16(u.x), lo_bf16(u.y), hi_bf16(u.y),
                  lo_bf16(u.z), hi_bf16(u.z), lo_bf16(u.w), hi_bf16(u.w)};
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) ss += x[j] * x[j];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    // q_mul: the attention's score scale folded into Q before its one bf16 rounding (see launch_attention)
    const float r = rsqrtf(ss * (1.0f / 128.0f) + 1e-6f) * (which ? 1.0f : q_mul);
    const uint16_t* sc = (row < split) ? (which ? ks0 : qs0) : (which ? ks1 : qs1);
    const uint4 su = *reinterpret_cast<const uint4*>(sc + sub * 8);
    const float w[8] = {lo_bf16(su.x), hi_bf16(su.x), lo_bf16(su.y), hi_bf16(su.y),
                        lo_bf16(su.z), hi_bf16(su.z), lo_bf16(su.w), hi_bf16(su.w)};
    const float4 c0 = *reinterpret_cast<const float4*>(cos_tab + (size_t)row * 128 + sub * 8);
    const float4 c1 = *reinterpret_cast<const float4*>(cos_tab + (size_t)row * 128 + sub * 8 + 4);
    const float4 s0 = *reinterpret_cast<const float4*>(sin_tab + (size_t)row * 128 + sub * 8);
    const float4 s1 = *reinterpret_cast<const float4*>(sin_tab + (size_t)row * 128 + sub * 8 + 4);
    const float cs[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
    const float sn[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    float o[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float n0 = x[2 * j] * r * w[2 * j], n1 = x[2 * j + 1] * r * w[2 * j + 1];
        // x*cos + rotate_half(x)*sin with rotate_half = (-x1, x0)
        o[2 * j] = n0 * cs[2 * j] - n1 * sn[2 * j];
        o[2 * j + 1] = n1 * cs[2 * j + 1] + n0 * sn[2 * j + 1];
    }
    if (live) {
        uint4 pk;
        pk.x = pack_bf16x2(o[0], o[1]); pk.y = pack_bf16x2(o[2], o[3]);
        pk.z = pack_bf16x2(o[4], o[5]); pk.w = pack_bf16x2(o[6], o[7]);
        *reinterpret_cast<uint4*>(p) = pk;
        if (qk8) {   // the e4m3 copy the fp8 models' attention takes for Q K^T (attention64b.hip, QK8): [S][q | k] bytes, power-of-two scales
            const float m8 = which ? k8_mul : q8_mul;
            uint2 e;
            e.x = pack_fp8x4(o[0] * m8, o[1] * m8, o[2] * m8, o[3] * m8);
            e.y = pack_fp8x4(o[4] * m8, o[5] * m8, o[6] * m8, o[7] * m8);
            *reinterpret_cast<uint2*>(qk8 + (size_t)row * (2 * heads * 128) + which * heads * 128 + head * 128 + sub * 8) = e;
        }
    }
}

// ------------------------------------------------------------------ V -> V^T (key-permuted)
// vt[head][d][pos(key)] with, inside every group of 16 keys, key offset 8a+4h+b stored at
// 8h+4a+b (the k order of an MFMA 32x32 accumulator reused as operand; see attention.hip).
// One workgroup = 64 keys x 128 d of one head, transposed through LDS.
__global__ __launch_bounds__(256) void v_transpose_kernel(const uint16_t* __restrict__ qkv, int ldqkv, int S,
                                                          int v_col0, uint16_t* __restrict__ vt, int s_pad) {
    __shared__ uint16_t tile[64][128 + 2];
    const int head = blockIdx.y;
    const int k0 = blockIdx.x * 64;
    const uint16_t* vsrc = qkv + v_col0 + head * 128;
    // load 64 rows x 256 B: 1024 chunks of 16 B, 4 per thread
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = threadIdx.x + 256 * i;
        const int r = idx >> 4, c = idx & 15;
        uint4 val = make_uint4(0, 0, 0, 0);
        if (k0 + r < S) val = *reinterpret_cast<const uint4*>(vsrc + (size_t)(k0 + r) * ldqkv + c * 8);
        uint32_t* dst = reinterpret_cast<uint32_t*>(&tile[r][c * 8]);
        dst[0] = val.x; dst[1] = val.y; dst[2] = val.z; dst[3] = val.w;
    }
    __syncthreads();
    // write 128 d-rows x 64 keys: 1024 chunks of 8 keys (16 B), 4 per thread
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = threadIdx.x + 256 * i;
        const int d = idx >> 3, c = idx & 7;   // chunk c holds stored positions 8c .. 8c+7
        uint16_t vals[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int pos = c * 8 + j;                 // stored position within the 64-key tile
            const int grp = pos & ~15, o = pos & 15;   // o = 8h + 4a + b
            const int h = o >> 3, a = (o >> 2) & 1, b = o & 3;
            const int key = grp + 8 * a + 4 * h + b;
            vals[j] = tile[key][d];
        }
        uint4 pk;
        pk.x = vals[0] | ((uint32_t)vals[1] << 16);
        pk.y = vals[2] | ((uint32_t)vals[3] << 16);
        pk.z = vals[4] | ((uint32_t)vals[5] << 16);
        pk.w = vals[6] | ((uint32_t)vals[7] << 16);
        *reinterpret_cast<uint4*>(vt + ((size_t)head * 128 + d) * s_pad + k0 + c * 8) = pk;
    }
}

// Reference-token V^T columns <-> compact cache (see kernels.h).  Thread = one (row, key).
__global__ __launch_bounds__(256) void vt_keys_copy_kernel(uint16_t* __restrict__ vt, int s_pad, uint16_t* __restrict__ cache,
                                                           long rows, int key0, int nkeys, int to_cache) {
    const long total = rows * nkeys;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int kk = (int)(i % nkeys);
        const long row = i / nkeys;
        const int key = key0 + kk;
        const int o = key & 15;                                  // key offset 8a + 4h + b is stored at 8h + 4a + b
        const int col = (key & ~15) | (((o >> 2) & 1) << 3) | ((o >> 3) << 2) | (o & 3);
        if (to_cache) cache[i] = vt[row * s_pad + col];
        else vt[row * s_pad + col] = cache[i];
    }
}

// ------------------------------------------------------------------ text encoder (Qwen3)
// RMSNorm of an f32 row with a bf16 weight -> bf16 (Qwen3DecoderLayer.swift:25-26, MLXNN RMSNorm)
__global__ __launch_bounds__(256) void rmsnorm_rows_kernel(const float* __restrict__ x, const uint16_t* __restrict__ w,
                                                           uint16_t* __restrict__ y, int dim, float eps) {
    __shared__ float red[4];
    const int row = blockIdx.x;
    const float* xr = x + (size_t)row * dim;
    const int nvec = dim >> 2;
    float4 v[LN_MAX_VEC];
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_VEC; ++i) {
        const int c = threadIdx.x + 256 * i;
        if (c < nvec) {
            v[i] = reinterpret_cast<const float4*>(xr)[c];
            sq += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
        }
    }
    const float rstd = rsqrtf(block_sum_256(sq, red) / (float)dim + eps);
    uint16_t* yr = y + (size_t)row * dim;
#pragma unroll
    for (int i = 0; i < LN_MAX_VEC; ++i) {
        const int c = threadIdx.x + 256 * i;
        if (c < nvec) {
            const uint2 wv = reinterpret_cast<const uint2*>(w)[c];
            uint2 pk;
            pk.x = pack_bf16x2(v[i].x * rstd * lo_bf16(wv.x), v[i].y * rstd * hi_bf16(wv.x));
            pk.y = pack_bf16x2(v[i].z * rstd * lo_bf16(wv.y), v[i].w * rstd * hi_bf16(wv.y));
            reinterpret_cast<uint2*>(yr)[c] = pk;
        }
    }
}

// Per (row, head) vector of 128: 64 lanes, lane i holds elements i and i + 64 = one RoPE pair of the
// split-half convention (Qwen3Attention.swift:110-123: q_norm / k_norm before RoPE, traditional: false).
__global__ __launch_bounds__(256) void qk_norm_rope_llm_kernel(uint16_t* __restrict__ qkv, int ld, int S, int nvec_heads,
                                                               int hq, const uint16_t* __restrict__ q_w,
                                                               const uint16_t* __restrict__ k_w, float eps,
                                                               float log2_theta) {
    const int lane = threadIdx.x & 63;
    const long vec = (long)blockIdx.x * 4 + (threadIdx.x >> 6);   // (row, head) over the hq + hkv q/k heads
    if (vec >= (long)S * nvec_heads) return;
    const int row = (int)(vec / nvec_heads), head = (int)(vec % nvec_heads);   // heads 0..hq-1 = q, then k
    uint16_t* p = qkv + (size_t)row * ld + head * 128;
    const uint16_t* w = head < hq ? q_w : k_w;
    const float a = bf16_bits_to_f32(p[lane]), b = bf16_bits_to_f32(p[lane + 64]);
    float an = a, bn = b;
    if (w != nullptr) {   // wave-uniform
        const float ss = wave_sum(a * a + b * b);
        const float rstd = rsqrtf(ss * (1.0f / 128.0f) + eps);
        // the reference normalises in the activation dtype: round the normalised value to bf16 before RoPE
        an = bf16_bits_to_f32(f32_to_bf16_bits(a * rstd * bf16_bits_to_f32(w[lane])));
        bn = bf16_bits_to_f32(f32_to_bf16_bits(b * rstd * bf16_bits_to_f32(w[lane + 64])));
    }
    const float inv_freq = exp2f(-log2_theta * (float)(2 * lane) * (1.0f / 128.0f));
    float sn, cs;
    sincosf((float)row * inv_freq, &sn, &cs);
    p[lane] = f32_to_bf16_bits(an * cs - bn * sn);
    p[lane + 64] = f32_to_bf16_bits(bn * cs + an * sn);
}

// ------------------------------------------------------------------ RoPE tables
// cos/sin [S,128]: axis a (0..3) covers columns 32a..32a+31; pair j (0..15) of an axis uses
// omega = theta^(-2j/32), theta = 2000, value repeated for columns 2j, 2j+1.
__global__ void rope_tables_kernel(const int32_t* __restrict__ txt_ids, int s_txt,
                                   const int32_t* __restrict__ img_ids, int s_img, float* __restrict__ cos_tab,
                                   float* __restrict__ sin_tab, float* __restrict__ pos_tab) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // (row, axis, j)
    const int S = s_txt + s_img;
    if (idx >= S * 64) return;
    const int row = idx >> 6, aj = idx & 63, axis = aj >> 4, j = aj & 15;
    const int32_t* ids = (row < s_txt) ? (txt_ids + (size_t)row * 4) : (img_ids + (size_t)(row - s_txt) * 4);
    const float pos = (float)ids[axis];
    const float inv_freq = 1.0f / powf(2000.0f, (float)(2 * j) / 32.0f);
    const float ang = pos * inv_freq;
    const float c = cosf(ang), s = sinf(ang);
    const size_t o = (size_t)row * 128 + axis * 32 + j * 2;
    cos_tab[o] = c; cos_tab[o + 1] = c;
    sin_tab[o] = s; sin_tab[o + 1] = s;
    if (pos_tab != nullptr && j == 0) pos_tab[(size_t)row * 4 + axis] = pos;
}

}  // namespace

int launch_layernorm_modulate(hipStream_t stream, const float* x, uint16_t* y, int rows, int dim, int split,
                              const float* shift0, const float* scale0, const float* shift1,
                              const float* scale1) {
    if (rows <= 0 || dim <= 0 || (dim & 3) || dim > LN_MAX_VEC * 1024) return 1;
    hipLaunchKernelGGL(layernorm_modulate_kernel, dim3(rows), dim3(256), 0, stream, x, y, dim, split, shift0,
                       scale0, shift1, scale1);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

int launch_layernorm_modulate_fp8(hipStream_t stream, const float* x, uint8_t* y, float* y_scale, int rows, int dim,
                                  int split, const float* shift0, const float* scale0, const float* shift1,
                                  const float* scale1) {
    if (rows <= 0 || dim <= 0 || (dim & 3) || dim > LN_MAX_VEC * 1024) return 1;
    hipLaunchKernelGGL(layernorm_modulate_fp8_kernel, dim3(rows), dim3(256), 0, stream, x, y, y_scale, dim, split,
                       shift0, scale0, shift1, scale1);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

int launch_quantize_rows_fp8(hipStream_t stream, const uint16_t* x, int ld, uint8_t* q, int ldq, float* scale,
                             int rows, int cols) {
    if (rows <= 0 || cols <= 0 || (cols & 7) || (ld & 7) || (ldq & 7)) return 1;
    hipLaunchKernelGGL(quantize_rows_fp8_kernel, dim3(rows), dim3(256), 0, stream, x, ld, q, ldq, scale, cols);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

int launch_quantize_blocks_fp8(hipStream_t stream, const uint16_t* x, int ld, uint8_t* q, int ldq, uint8_t* bscale, int lds, int rows, int cols) {
    if (rows <= 0 || cols <= 0 || (cols & 127) || (ld & 7) || (ldq & 7) || lds < (rows + 15) / 16) return 1;
    hipLaunchKernelGGL(quantize_blocks_fp8_kernel, dim3(cols / 128, (rows + 15) / 16), dim3(256), 0, stream, x, ld, q, ldq, bscale, lds, rows);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

int launch_qk_norm_rope_vt(hipStream_t stream, uint16_t* qkv, int ldqkv, int S, int heads, int split,
                           const uint16_t* qscale0, const uint16_t* kscale0, const uint16_t* qscale1,
                           const uint16_t* kscale1, const float* cos_tab, const float* sin_tab,
                           uint16_t* vt, int s_pad, float q_mul, uint8_t* qk8, int q8_exp, int k8_exp) {
    if (S <= 0 || heads <= 0 || (ldqkv & 7) || s_pad < attn_s_pad(S)) return 1;
    const long total = (long)S * 2 * heads;
    hipLaunchKernelGGL(qk_norm_rope_kernel, dim3((unsigned)((total + 15) / 16)), dim3(256), 0, stream, qkv, ldqkv, S,
                       heads, split, qscale0, kscale0, qscale1, kscale1, cos_tab, sin_tab, q_mul, qk8, ldexpf(1.0f, q8_exp), ldexpf(1.0f, k8_exp));
    hipLaunchKernelGGL(v_transpose_kernel, dim3((S + 63) / 64, heads), dim3(256), 0, stream, qkv, ldqkv, S,
                       2 * heads * 128, vt, s_pad);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

int launch_v_transpose_only(hipStream_t stream, const uint16_t* qkv, int ldqkv, int S, int heads, uint16_t* vt,
                            int s_pad, int v_col0) {
    if (v_col0 < 0) v_col0 = 2 * heads * 128;   // the fused [q | k | v] row
    if (S <= 0 || heads <= 0 || (ldqkv & 7) || (v_col0 & 7) || v_col0 + heads * 128 > ldqkv || s_pad < attn_s_pad(S)) return 1;
    hipLaunchKernelGGL(v_transpose_kernel, dim3((S + 63) / 64, heads), dim3(256), 0, stream, qkv, ldqkv, S,
                       v_col0, vt, s_pad);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

int launch_vt_keys_copy(hipStream_t stream, uint16_t* vt, int s_pad, uint16_t* cache, int heads, int key0, int nkeys,
                        int to_cache) {
    if (heads <= 0 || key0 < 0 || nkeys <= 0 || key0 + nkeys > s_pad) return 1;
    const long total = (long)heads * 128 * nkeys;
    const int blocks = (int)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
    hipLaunchKernelGGL(vt_keys_copy_kernel, dim3(blocks), dim3(256), 0, stream, vt, s_pad, cache, (long)heads * 128, key0,
                       nkeys, to_cache);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

int launch_rmsnorm_rows(hipStream_t stream, const float* x, const uint16_t* w, uint16_t* y, int rows, int dim, float eps) {
    if (rows <= 0 || dim <= 0 || (dim & 3) || dim > LN_MAX_VEC * 1024) return 1;
    hipLaunchKernelGGL(rmsnorm_rows_kernel, dim3(rows), dim3(256), 0, stream, x, w, y, dim, eps);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

int launch_qk_norm_rope_llm_vt(hipStream_t stream, uint16_t* qkv, int S, int hq, int hkv, const uint16_t* q_w,
                               const uint16_t* k_w, float eps, float theta, uint16_t* vt, int s_pad) {
    if (S <= 0 || hq <= 0 || hkv <= 0 || theta <= 1.0f || s_pad < attn_s_pad(S)) return 1;
    const int ld = (hq + 2 * hkv) * 128;
    const long total = (long)S * (hq + hkv);
    hipLaunchKernelGGL(qk_norm_rope_llm_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, stream, qkv, ld, S, hq + hkv,
                       hq, q_w, k_w, eps, log2f(theta));
    hipLaunchKernelGGL(v_transpose_kernel, dim3((S + 63) / 64, hkv), dim3(256), 0, stream, qkv, ld, S, (hq + hkv) * 128,
                       vt, s_pad);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

int launch_rope_tables(hipStream_t stream, const int32_t* txt_ids, int s_txt, const int32_t* img_ids, int s_img,
                       float* cos_tab, float* sin_tab, float* pos_tab) {
    const int total = (s_txt + s_img) * 64;
    if (total <= 0) return 1;
    hipLaunchKernelGGL(rope_tables_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, txt_ids, s_txt, img_ids,
                       s_img, cos_tab, sin_tab, pos_tab);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
