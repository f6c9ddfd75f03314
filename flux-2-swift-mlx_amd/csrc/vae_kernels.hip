// vae_kernels.hip — gfx950 kernels of the VAE decoder (VAE/VAEDecoder.swift:91-121,
// VAE/ResnetBlock.swift): NHWC bf16 activations (the reference is NHWC too), f32
// accumulation and f32 GroupNorm statistics.
//
//   * conv2d: implicit GEMM, M = output pixels, N = Cout, K = taps x Cin, MFMA 16x16x32.
//     Workgroup = 4 waves x 32 pixels, BN in {32,64,96,128} output channels, K-step = one
//     tap x 32 input channels.  Nearest-2x upsample (ResnetBlock.swift:217-254) is folded
//     into the gather (input coordinate >> 1), bias / residual / final u8 conversion into
//     the epilogue.
//   * GroupNorm: two-stage statistics (per-workgroup f32 partials, f64 finalize) producing a
//     per-channel affine (a_c, b_c); apply (+SiLU) is one bf16 pass.
#include "dev.h"
#include "kernels.h"
#include <stdlib.h>
#include <algorithm>

namespace {

constexpr int CBM = 128, CBK = 32;

template <int BN>
__global__ __launch_bounds__(256) void conv2d_kernel(ConvArgs a, int Ho, int Wo, int cout_pad) {
    constexpr int NT = BN / 16;
    constexpr int A_BYTES = CBM * CBK * 2;  // 8 KB
    constexpr int B_BYTES = BN * CBK * 2;
    __shared__ __attribute__((aligned(16))) uint8_t smem[2 * (A_BYTES + B_BYTES)];

    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const int P = Ho * Wo;
    const int tiles_n = (a.Cout + BN - 1) / BN;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    const int m0 = tm * CBM, n0 = tn * BN;
    const int taps = a.ks * a.ks, pad = a.ks >> 1;
    const int cblocks = a.Cin / CBK;
    const int nsteps = taps * cblocks;
    const int up = a.upsample;
    // input-view extent and the output -> input coordinate map: iy = oy*stride + ky - pad
    const int stride = a.downsample ? 2 : 1;
    const int pad_tl = a.downsample ? 0 : pad;
    const int Hv = a.Hin << up, Wv = a.Win << up;

    // ---- A staging: 2 chunks/thread: row = idx>>2 (pixel), chunk = idx&3 (8 channels)
    int a_y[2], a_x[2], a_lds[2], a_chunk[2];
    bool a_ok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = t + 256 * i;
        const int row = idx >> 2, chunk = idx & 3;
        const int p = m0 + row;
        a_ok[i] = p < P;
        const int pp = a_ok[i] ? p : 0;
        a_y[i] = pp / Wo;
        a_x[i] = pp % Wo;
        a_chunk[i] = chunk;
        a_lds[i] = row * 64 + ((chunk ^ ((-(row >> 2)) & 3)) << 4);
    }
    // ---- B staging: BN rows x 4 chunks
    constexpr int B_CHUNKS = BN * 4;
    constexpr int B_PER_THREAD = (B_CHUNKS + 255) / 256;
    int b_lds[B_PER_THREAD], b_row[B_PER_THREAD], b_chunk[B_PER_THREAD];
    bool b_act[B_PER_THREAD];
#pragma unroll
    for (int i = 0; i < B_PER_THREAD; ++i) {
        const int idx = t + 256 * i;
        b_act[i] = idx < B_CHUNKS;
        const int row = (idx >> 2) % BN, chunk = idx & 3;
        b_row[i] = min(n0 + row, cout_pad - 1);
        b_chunk[i] = chunk;
        b_lds[i] = A_BYTES + row * 64 + ((chunk ^ ((-(row >> 2)) & 3)) << 4);
    }

    u32x4 a_reg[2], b_reg[B_PER_THREAD];
#define CONV_LOAD_STEP(s_)                                                                                     \
    {                                                                                                          \
        const int tap_ = (s_) / cblocks, cb_ = (s_) - tap_ * cblocks;                                          \
        const int ky_ = tap_ / a.ks, kx_ = tap_ - ky_ * a.ks;                                                  \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                     \
            const int iy_ = a_y[i_] * stride + ky_ - pad_tl, ix_ = a_x[i_] * stride + kx_ - pad_tl;            \
            const bool ok_ = a_ok[i_] && iy_ >= 0 && iy_ < Hv && ix_ >= 0 && ix_ < Wv;                         \
            u32x4 v_ = {0u, 0u, 0u, 0u};                                                                 \
            if (ok_) {                                                                                         \
                const size_t off_ = ((size_t)(iy_ >> up) * a.Win + (ix_ >> up)) * a.Cin + cb_ * CBK + a_chunk[i_] * 8; \
                v_ = *reinterpret_cast<const u32x4*>(a.x + off_);                                              \
            }                                                                                                  \
            a_reg[i_] = v_;                                                                                    \
        }                                                                                                      \
        _Pragma("unroll") for (int i_ = 0; i_ < B_PER_THREAD; ++i_) {                                          \
            if (b_act[i_]) {                                                                                   \
                const size_t off_ = ((size_t)b_row[i_] * taps + tap_) * a.Cin + cb_ * CBK + b_chunk[i_] * 8;   \
                b_reg[i_] = *reinterpret_cast<const u32x4*>(a.w + off_);                                       \
            }                                                                                                  \
        }                                                                                                      \
    }
#define CONV_STORE_STEP(buf_)                                                                                  \
    {                                                                                                          \
        uint8_t* base_ = smem + (buf_) * (A_BYTES + B_BYTES);                                                  \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) *reinterpret_cast<u32x4*>(base_ + a_lds[i_]) = a_reg[i_]; \
        _Pragma("unroll") for (int i_ = 0; i_ < B_PER_THREAD; ++i_)                                            \
            if (b_act[i_]) *reinterpret_cast<u32x4*>(base_ + b_lds[i_]) = b_reg[i_];                           \
    }

    f32x4 acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fq = lane >> 4;
    const int fslot = ((fq ^ ((-(frow >> 2)) & 3)) << 4);

    CONV_LOAD_STEP(0)
    CONV_STORE_STEP(0)
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        if (s + 1 < nsteps) CONV_LOAD_STEP(s + 1)
        const uint8_t* sa = smem + buf * (A_BYTES + B_BYTES);
        const uint8_t* sb = sa + A_BYTES;
        bf16x8 af[2], bfr[NT];
#pragma unroll
        for (int i = 0; i < 2; ++i)
            af[i] = *reinterpret_cast<const bf16x8*>(sa + (wid * 32 + i * 16 + frow) * 64 + fslot);
#pragma unroll
        for (int j = 0; j < NT; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(sb + (j * 16 + frow) * 64 + fslot);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        if (s + 1 < nsteps) CONV_STORE_STEP(buf ^ 1)
        __syncthreads();
    }

    // ---- epilogue
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int col = n0 + j * 16 + frow;
        if (col >= a.Cout) continue;
        const float bias = a.bias ? a.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int p = m0 + wid * 32 + i * 16 + fq * 4 + r;
                if (p >= P) continue;
                float v = acc[i][j][r] + bias;
                if (a.resid) v += bf16_bits_to_f32(a.resid[(size_t)p * a.Cout + col]);
                if (a.y) a.y[(size_t)p * a.Cout + col] = f32_to_bf16_bits(v);
                if (a.y_f32_nchw) a.y_f32_nchw[(size_t)col * P + p] = v;
                if (a.y_u8_hwc) {
                    // postprocessVAEOutput (Flux2Pipeline.swift:2438-2444): clip then truncating cast
                    const float u = fminf(fmaxf((v + 1.0f) * 127.5f, 0.0f), 255.0f);
                    a.y_u8_hwc[(size_t)p * a.Cout + col] = (uint8_t)u;
                }
            }
    }
}

// ---------------------------------------------------------------- conv2d, 128x128x64 tile
// The decoder's heavy convolutions (Cin % 64 == 0, bf16 NHWC output): implicit GEMM with a 128-pixel x
// 128-channel tile and a K-step of one tap x 64 input channels.  4 waves as 2 x 2, wave tile 64 x 64 =
// 4 x 4 MFMA 16x16x32 per 32 k (the 32-pixel x 128-channel wave tile of conv2d_kernel reads 160 B of LDS per
// CU-clock at full MFMA rate; this shape 128), LDS 2 x [A 128x64 | W 128x64] bf16 = 64 KB XOR-swizzled, two
// workgroups per CU, operands staged through registers (issued before the MFMAs of a K-step, written to
// LDS after them), one barrier per K-step.  The weight fragment is the MFMA's A operand, so a lane holds 4
// CONSECUTIVE output channels of one pixel: bias, residual and the store are 16 / 8 / 8-byte accesses.
// TK = 64 (Cin % 64 == 0) or 32 (the 96-channel level of the small decoder: rows of 64 B, which ds_read_b128 reads
// conflict-free without a swizzle; one MFMA k-step per barrier).
// PH (a.w_phase): the upsampling 3x3 convolution as four 2x2 convolutions on the source image.  The tile runs over SOURCE pixels
// (Ho, Wo = Hin, Win here), a workgroup takes one phase (py, px) of its pixel block and stores to output pixel (2y + py, 2x + px).
template <int TK, bool PH = false>
__global__ __launch_bounds__(256, 2) void conv2d_g128_kernel(ConvArgs a, int Ho, int Wo, int cout_pad) {
    constexpr int TM = 128, TN = 128;
    constexpr int RB = TK * 2;            // bytes per LDS row
    constexpr int CPR = TK / 8;           // 16-byte chunks per row
    constexpr int NP = (TM * CPR) / 256;  // chunks per operand per thread
    __shared__ __attribute__((aligned(16))) uint8_t smem[2 * 2 * TM * TK * 2];   // 64 KB

    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int P = Ho * Wo;
    const int tiles_m = (P + TM - 1) / TM, tiles_n = (a.Cout + TN - 1) / TN;
    // XCD remap: workgroups with equal (blockIdx % 8) share an XCD; give each a contiguous run of tiles, the
    // N-tiles of one pixel block adjacent (they re-read the same activation rows from that XCD's L2)
    int tm, tn, ph = 0;
    {
        const int ntiles = tiles_m * tiles_n * (PH ? 4 : 1), q = ntiles >> 3, r = ntiles & 7;
        const int xcd = blockIdx.x & 7, within = blockIdx.x >> 3;
        int idx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
        if (PH) { ph = idx & 3; idx >>= 2; }   // the four phases of a pixel block are neighbours: they read the same source rows
        tm = idx / tiles_n;
        tn = idx - tm * tiles_n;
    }
    const int py = ph >> 1, px = ph & 1;
    const int m0 = tm * TM, n0 = tn * TN;
    const int ks = PH ? 2 : a.ks;
    const int taps = ks * ks, pad = PH ? 0 : (a.ks >> 1);
    const int cblocks = a.Cin / TK;
    const int nk = taps * cblocks;
    const int up = PH ? 0 : a.upsample;
    const int stride = a.downsample ? 2 : 1;
    const int pad_tl = a.downsample ? 0 : pad;
    const int Hv = a.Hin << up, Wv = a.Win << up;
    const uint16_t* wbase = PH ? a.w_phase + (size_t)ph * cout_pad * taps * a.Cin : a.w;

    // ---- staging: NP x 16 B per operand per thread: rows t / CPR + (256 / CPR) i, chunk t % CPR
    const int chunk = t % CPR;
    int a_y[NP], a_x[NP], lds_off[NP];
    const uint16_t* w_src[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int row = t / CPR + (256 / CPR) * i;
        const int p = m0 + row;
        const bool ok = p < P;
        const int pp = ok ? p : 0;
        a_y[i] = ok ? (PH ? pp / Wo + py - 1 : (pp / Wo) * stride - pad_tl) : -100000;   // a dead row is out of bounds for every tap
        a_x[i] = PH ? pp % Wo + px - 1 : (pp % Wo) * stride - pad_tl;
        w_src[i] = wbase + (size_t)min(n0 + row, cout_pad - 1) * taps * a.Cin + chunk * 8;
        lds_off[i] = TK == 64 ? row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4) : row * 64 + (chunk << 4);
    }
    u32x4 a_reg[NP], w_reg[NP];
#define CG_LOAD(kt_)                                                                                           \
    {                                                                                                          \
        const int tap_ = (kt_) / cblocks, cb_ = (kt_) - tap_ * cblocks;                                        \
        const int ky_ = tap_ / ks, kx_ = tap_ - ky_ * ks;                                                      \
        _Pragma("unroll") for (int i_ = 0; i_ < NP; ++i_) {                                                    \
            const int iy_ = a_y[i_] + ky_, ix_ = a_x[i_] + kx_;                                                \
            u32x4 v_ = {0u, 0u, 0u, 0u};                                                                       \
            if (iy_ >= 0 && iy_ < Hv && ix_ >= 0 && ix_ < Wv)                                                  \
                v_ = *reinterpret_cast<const u32x4*>(a.x + ((size_t)(iy_ >> up) * a.Win + (ix_ >> up)) * a.Cin + cb_ * TK + chunk * 8); \
            a_reg[i_] = v_;                                                                                    \
            w_reg[i_] = *reinterpret_cast<const u32x4*>(w_src[i_] + (size_t)(kt_) * TK);                       \
        }                                                                                                      \
    }
#define CG_STORE(buf_)                                                                                         \
    {                                                                                                          \
        uint8_t* base_ = smem + (buf_) * (2 * TM * TK * 2);                                                    \
        _Pragma("unroll") for (int i_ = 0; i_ < NP; ++i_) {                                                    \
            *reinterpret_cast<u32x4*>(base_ + lds_off[i_]) = a_reg[i_];                                        \
            *reinterpret_cast<u32x4*>(base_ + TM * TK * 2 + lds_off[i_]) = w_reg[i_];                          \
        }                                                                                                      \
    }

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fq = lane >> 4;
    const int sw = (frow >> 1) & 7;
    CG_LOAD(0)
    CG_STORE(0)
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) CG_LOAD(kt + 1)
        const uint8_t* sa = smem + buf * (2 * TM * TK * 2);
        const uint8_t* sb = sa + TM * TK * 2;
#pragma unroll
        for (int kk = 0; kk < TK / 32; ++kk) {
            const int coff = TK == 64 ? (((kk * 4 + fq) ^ sw) << 4) : (fq << 4);
            bf16x8 af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i] = *reinterpret_cast<const bf16x8*>(sa + (wm * 64 + i * 16 + frow) * RB + coff);
                bf[i] = *reinterpret_cast<const bf16x8*>(sb + (wn * 64 + i * 16 + frow) * RB + coff);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) CG_STORE(buf ^ 1)
        __syncthreads();
    }
#undef CG_LOAD
#undef CG_STORE

    // ---- epilogue: acc[i][j][r] = pixel m0 + wm*64 + i*16 + frow, channel n0 + wn*64 + j*16 + fq*4 + r
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = n0 + wn * 64 + j * 16 + fq * 4;
        if (col >= a.Cout) continue;                      // Cout % 4 == 0 (launcher)
        const float4 bs = a.bias ? *reinterpret_cast<const float4*>(a.bias + col) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ps = m0 + wm * 64 + i * 16 + frow;
            if (ps >= P) continue;
            const size_t p = PH ? (size_t)(2 * (ps / Wo) + py) * (size_t)(2 * Wo) + (size_t)(2 * (ps % Wo) + px) : (size_t)ps;
            float v0 = acc[i][j][0] + bs.x, v1 = acc[i][j][1] + bs.y, v2 = acc[i][j][2] + bs.z, v3 = acc[i][j][3] + bs.w;
            if (a.resid) {
                const uint2 rv = *reinterpret_cast<const uint2*>(a.resid + p * a.Cout + col);
                v0 += lo_bf16(rv.x); v1 += hi_bf16(rv.x); v2 += lo_bf16(rv.y); v3 += hi_bf16(rv.y);
            }
            *reinterpret_cast<uint2*>(a.y + p * a.Cout + col) = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
        }
    }
}

// ---------------------------------------------------------------- GroupNorm
// stage 1: per-workgroup partial (sum, sumsq) per group.  Thread owns a fixed 8-channel
// chunk and strides over pixels.  The per-thread sums are combined per group in a FIXED order (no
// atomics): the decode is bit-reproducible run to run.
__global__ __launch_bounds__(256) void gn_partial_kernel(const uint16_t* __restrict__ x, int64_t pixels, int C,
                                                         int groups, int64_t px_per_block,
                                                         float* __restrict__ partial) {
    __shared__ float sh_s[256][9], sh_q[256][9];   // [thread][channel of its chunk], padded
    const int cpb = C >> 3;             // 8-channel chunks per pixel
    const int ppi = 256 / cpb;          // pixels per iteration
    const int cc = threadIdx.x % cpb, pl = threadIdx.x / cpb;
    const int64_t p0 = (int64_t)blockIdx.x * px_per_block;
    const int64_t p1 = min(pixels, p0 + px_per_block);
    float s[8], q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s[j] = 0.f; q[j] = 0.f; }
    if (pl < ppi) {
        for (int64_t p = p0 + pl; p < p1; p += ppi) {
            const uint4 v = *reinterpret_cast<const uint4*>(x + p * C + cc * 8);
            const float f[8] = {lo_bf16(v.x), hi_bf16(v.x), lo_bf16(v.y), hi_bf16(v.y),
                                lo_bf16(v.z), hi_bf16(v.z), lo_bf16(v.w), hi_bf16(v.w)};
#pragma unroll
            for (int j = 0; j < 8; ++j) { s[j] += f[j]; q[j] += f[j] * f[j]; }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { sh_s[threadIdx.x][j] = s[j]; sh_q[threadIdx.x][j] = q[j]; }
    __syncthreads();
    if (threadIdx.x < groups) {
        const int g = threadIdx.x, cpg = C / groups;
        float gs = 0.f, gq = 0.f;
        for (int l = 0; l < ppi; ++l)                       // pixel lanes, then the group's channels: fixed order
            for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
                const int tt = l * cpb + (c >> 3);
                gs += sh_s[tt][c & 7];
                gq += sh_q[tt][c & 7];
            }
        partial[((size_t)blockIdx.x * groups + g) * 2 + 0] = gs;
        partial[((size_t)blockIdx.x * groups + g) * 2 + 1] = gq;
    }
}

// stage 2: one workgroup; f64 combine -> per-channel a_c = rstd*gamma, b_c = beta - mean*a_c
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ partial, int nblocks, int64_t pixels,
                                                          int C, int groups, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps,
                                                          float* __restrict__ ab) {
    __shared__ double dsum[64], dsq[64];
    __shared__ float gmean[64], grstd[64];
    __shared__ double ps[256], pq[256];
    const int t = threadIdx.x;
    // thread = (block lane bl, group g): consecutive threads read consecutive float2 of one block's row of
    // partials; 8 independent loads in flight per thread (the first form's one dependent load per iteration
    // made this single-workgroup kernel 50 us long), lanes combined in a fixed order through LDS
    const int nbl = 256 / groups;          // groups <= 64 (launcher)
    const int g = t % groups, bl = t / groups;
    double s = 0.0, q = 0.0;
    if (bl < nbl) {
        const float2* pp = reinterpret_cast<const float2*>(partial);
        int b = bl;
        for (; b + 7 * nbl < nblocks; b += 8 * nbl) {
            float2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = pp[(size_t)(b + u * nbl) * groups + g];
#pragma unroll
            for (int u = 0; u < 8; ++u) { s += (double)v[u].x; q += (double)v[u].y; }
        }
        for (; b < nblocks; b += nbl) {
            const float2 v = pp[(size_t)b * groups + g];
            s += (double)v.x; q += (double)v.y;
        }
    }
    ps[t] = s; pq[t] = q;
    __syncthreads();
    if (t < groups) {
        double ss = 0.0, qq = 0.0;
        for (int l = 0; l < nbl; ++l) { ss += ps[l * groups + t]; qq += pq[l * groups + t]; }
        dsum[t] = ss; dsq[t] = qq;
    }
    __syncthreads();
    if (t < groups) {
        const double n = (double)pixels * (double)(C / groups);
        const double mean = dsum[t] / n;
        double var = dsq[t] / n - mean * mean;
        if (var < 0.0) var = 0.0;
        gmean[t] = (float)mean;
        grstd[t] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    const int cpg = C / groups;
    for (int c = t; c < C; c += 256) {
        const int gg = c / cpg;
        const float a = grstd[gg] * gamma[c];
        ab[c] = a;
        ab[C + c] = beta[c] - gmean[gg] * a;
    }
}

// Statistics from the producing convolution's epilogue (GemmProblem::gn_part): part f32 [rows][C][2].  One workgroup per group;
// thread t adds rows t, t + 256, ... of the group's channels in f64, the 256 threads are combined in a fixed order.
// CPG = channels per group when it is one of the decoders' (3, 4, 6, 8, 12, 16), else 0 = the run-time loop.  With CPG known the values of
// four of the thread's rows are REQUESTED before the first is added (the additions keep their order: identical sums): row by row the
// kernel was a chain of dependent memory latencies — 6-14 us per launch, 44 us behind an upsampling convolution, 55 launches per decode.
template <int CPG>
__global__ __launch_bounds__(256) void gn_finalize_channels_kernel(const float* __restrict__ part, int rows, int64_t pixels, int C, int groups,
                                                                   const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                                   float* __restrict__ ab) {
    __shared__ double ps[256], pq[256];
    __shared__ float s_mean, s_rstd;
    const int g = blockIdx.x, cpg = CPG ? CPG : C / groups, t = threadIdx.x;
    double s = 0.0, q = 0.0;
    if constexpr (CPG != 0) {
        constexpr int RU = CPG <= 4 ? 8 : (CPG <= 8 ? 4 : 2);   // rows in flight per thread
        for (int r0 = t; r0 < rows; r0 += 256 * RU) {
            float2 v[RU][CPG];
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                const int r = r0 + 256 * u;
                const float2* pp = reinterpret_cast<const float2*>(part + ((size_t)(r < rows ? r : t) * C + (size_t)g * CPG) * 2);
#pragma unroll
                for (int c = 0; c < CPG; ++c) v[u][c] = pp[c];
            }
#pragma unroll
            for (int u = 0; u < RU; ++u)
                if (r0 + 256 * u < rows) {
#pragma unroll
                    for (int c = 0; c < CPG; ++c) { s += (double)v[u][c].x; q += (double)v[u][c].y; }
                }
        }
    } else
    for (int r = t; r < rows; r += 256) {
        const float2* pp = reinterpret_cast<const float2*>(part + ((size_t)r * C + (size_t)g * cpg) * 2);
        for (int c = 0; c < cpg; ++c) {
            const float2 v = pp[c];
            s += (double)v.x; q += (double)v.y;
        }
    }
    ps[t] = s; pq[t] = q;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (t < w) { ps[t] += ps[t + w]; pq[t] += pq[t + w]; }
        __syncthreads();
    }
    if (t == 0) {
        const double n = (double)pixels * (double)cpg;
        const double mean = ps[0] / n;
        double var = pq[0] / n - mean * mean;
        if (var < 0.0) var = 0.0;
        s_mean = (float)mean;
        s_rstd = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    if (t < cpg) {
        const int c = g * cpg + t;
        const float a = s_rstd * gamma[c];
        ab[c] = a;
        ab[C + c] = beta[c] - s_mean * a;
    }
}

// The grid is sized so that (total threads) % cpb == 0: a thread then sees the same 8 channels in every
// iteration, its 16 affine coefficients stay in registers and no per-chunk modulo is needed (the 64-bit
// `i % cpb` + 16 dependent loads of the first form ran at a quarter of the HBM rate).
__global__ __launch_bounds__(256) void gn_apply_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y,
                                                       int64_t nchunks, int cpb, const float* __restrict__ ab, int C,
                                                       int silu) {
    const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
    const int c0 = (int)(tid % (uint32_t)cpb) * 8;
    float ca[8], cb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { ca[j] = ab[c0 + j]; cb[j] = ab[C + c0 + j]; }
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = tid; i < nchunks; i += stride) {
        const uint4 v = reinterpret_cast<const uint4*>(x)[i];
        float f[8] = {lo_bf16(v.x), hi_bf16(v.x), lo_bf16(v.y), hi_bf16(v.y),
                      lo_bf16(v.z), hi_bf16(v.z), lo_bf16(v.w), hi_bf16(v.w)};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float o = ca[j] * f[j] + cb[j];
            if (silu) o = silu_f(o);
            f[j] = o;
        }
        uint4 o;
        o.x = pack_bf16x2(f[0], f[1]); o.y = pack_bf16x2(f[2], f[3]);
        o.z = pack_bf16x2(f[4], f[5]); o.w = pack_bf16x2(f[6], f[7]);
        reinterpret_cast<uint4*>(y)[i] = o;
    }
}

// The same affine (+SiLU), written as a ZERO-HALOED image [(H + 2)][Wp = W + 2][C] — the operand layout of the implicit-GEMM
// convolution on gemm256s (kernels.h, GemmProblem::conv_*): every pixel of the padded extent is written, the halo with zeros, so the
// buffer needs no clearing whatever it held.  blockDim = ppi * cpb threads (<= 256): a thread keeps one 8-channel chunk and its 16
// coefficients, and walks the padded pixels of its block's range.
__global__ __launch_bounds__(256) void gn_apply_padded_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, int H, int W, int cpb,
                                                              int ppi, const float* __restrict__ ab, int C, int silu, uint32_t px_per_block) {
    const int cc = threadIdx.x % cpb, pl = threadIdx.x / cpb;
    float ca[8], cb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { ca[j] = ab[cc * 8 + j]; cb[j] = ab[C + cc * 8 + j]; }
    const uint32_t wp = (uint32_t)W + 2u, total = ((uint32_t)H + 2u) * wp;
    const uint32_t q0 = blockIdx.x * px_per_block, q1 = min(total, q0 + px_per_block);
    for (uint32_t q = q0 + pl; q < q1; q += ppi) {
        const uint32_t yy = q / wp, xx = q - yy * wp;
        uint4 o = make_uint4(0u, 0u, 0u, 0u);
        if (yy >= 1u && yy <= (uint32_t)H && xx >= 1u && xx <= (uint32_t)W) {
            const uint4 v = *reinterpret_cast<const uint4*>(x + ((size_t)(yy - 1u) * W + (xx - 1u)) * C + cc * 8);
            float f[8] = {lo_bf16(v.x), hi_bf16(v.x), lo_bf16(v.y), hi_bf16(v.y), lo_bf16(v.z), hi_bf16(v.z), lo_bf16(v.w), hi_bf16(v.w)};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float t = ca[j] * f[j] + cb[j];
                if (silu) t = silu_f(t);
                f[j] = t;
            }
            o.x = pack_bf16x2(f[0], f[1]); o.y = pack_bf16x2(f[2], f[3]);
            o.z = pack_bf16x2(f[4], f[5]); o.w = pack_bf16x2(f[6], f[7]);
        }
        *reinterpret_cast<uint4*>(y + (size_t)q * C + cc * 8) = o;
    }
}

// ---------------------------------------------------------------- conv_out with the final GroupNorm folded in
// decoder tail (VAEDecoder.swift:112-121): conv_norm_out -> SiLU -> conv_out 3x3 C -> 3, at full resolution.  As two kernels that is a
// pass writing the activated [H][W][C] tensor (268 MB at 1024^2 x 128) and a convolution whose 3 output channels leave 29 of the 32
// columns of its tile idle (352 us).  Here: one workgroup per 8 x 32 output pixels stages the 10 x 34 halo tile of the RAW input through
// registers — affine of the GroupNorm (a_c x + b_c), SiLU, bf16 — into LDS (zero outside the image: the convolution pads the ACTIVATED
// tensor), keeps the 9 x C/32 weight fragments of the 16 (3 live) output channels in registers, and runs 36 MFMA 16x16x32 per 16 pixels
// with the weights as the A operand: lanes fq = 0 end up with channels 0..2 of pixel frow.  Epilogue as conv2d_kernel's (f32 NCHW and /
// or clip((v + 1) 127.5) truncated to u8 HWC, Flux2Pipeline.swift:2438-2444).
template <int C>
__global__ __launch_bounds__(256) void conv_out_fused_kernel(const uint16_t* __restrict__ x, const float* __restrict__ ab,
                                                             const uint16_t* __restrict__ w, const float* __restrict__ bias, int H, int W,
                                                             int cout, float* __restrict__ y_f32_nchw, uint8_t* __restrict__ y_u8_hwc) {
    constexpr int TY = 8, TX = 32, HY = TY + 2, HX = TX + 2, CP = C / 8, PITCH = (C + 8) * 2, NCB = C / 32;
    __shared__ __attribute__((aligned(16))) uint8_t tile[HY * HX * PITCH];
    __shared__ float s_ab[2 * C];
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6, frow = lane & 15, fq = lane >> 4;
    const int x0 = blockIdx.x * TX, y0 = blockIdx.y * TY;
    for (int i = t; i < 2 * C; i += 256) s_ab[i] = ab[i];
    // weight fragments: lane (frow = output channel, fq) holds w[frow][tap][cb*32 + fq*8 .. +8]  (OHWI, O zero-padded to 32 rows)
    bf16x8 wf[9][NCB];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
            wf[tap][cb] = *reinterpret_cast<const bf16x8*>(w + ((size_t)frow * 9 + tap) * C + cb * 32 + fq * 8);
    __syncthreads();
    // staging in batches of UNR chunks per thread: the batch's global loads are all in flight before the first is consumed (one by one
    // the 22 loads of a thread are 22 exposed latencies: 20 us per tile)
    constexpr int NCH = HY * HX * CP, UNR = 6;
    for (int base = 0; base < NCH; base += 256 * UNR) {
        uint4 v[UNR];
        bool inb[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int idx = base + u * 256 + t;
            const int px = idx / CP, ch = idx - px * CP;
            const int ty = px / HX, tx = px - ty * HX;
            const int gy = y0 + ty - 1, gx = x0 + tx - 1;
            inb[u] = idx < NCH && gy >= 0 && gy < H && gx >= 0 && gx < W;
            v[u] = make_uint4(0u, 0u, 0u, 0u);
            if (inb[u]) v[u] = *reinterpret_cast<const uint4*>(x + ((size_t)gy * W + gx) * C + ch * 8);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int idx = base + u * 256 + t;
            if (idx >= NCH) break;
            const int px = idx / CP, ch = idx - px * CP;
            uint4 o = make_uint4(0u, 0u, 0u, 0u);
            if (inb[u]) {
                float f[8] = {lo_bf16(v[u].x), hi_bf16(v[u].x), lo_bf16(v[u].y), hi_bf16(v[u].y), lo_bf16(v[u].z), hi_bf16(v[u].z), lo_bf16(v[u].w), hi_bf16(v[u].w)};
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] = silu_f(s_ab[ch * 8 + j] * f[j] + s_ab[C + ch * 8 + j]);
                o.x = pack_bf16x2(f[0], f[1]); o.y = pack_bf16x2(f[2], f[3]);
                o.z = pack_bf16x2(f[4], f[5]); o.w = pack_bf16x2(f[6], f[7]);
            }
            *reinterpret_cast<uint4*>(tile + px * PITCH + ch * 16) = o;
        }
    }
    __syncthreads();
    // wave `wid`: tile rows 2 wid, 2 wid + 1; two groups of 16 pixels per row
    f32x4 acc[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int g = 0; g < 2; ++g) acc[r][g] = f32x4{0.f, 0.f, 0.f, 0.f};
    // step s = (tap, cb): the four pixel fragments of step s + 1 are read while the MFMAs of step s run
    const int lane_off = frow * PITCH + fq * 16;
    auto frag = [&](int s_, int r, int g) -> bf16x8 {
        const int tap = s_ / NCB, cb = s_ - tap * NCB, ky = tap / 3, kx = tap - ky * 3;
        return *reinterpret_cast<const bf16x8*>(tile + ((2 * wid + r + ky) * HX + g * 16 + kx) * PITCH + cb * 64 + lane_off);
    };
    bf16x8 pv[2][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) pv[0][k] = frag(0, k >> 1, k & 1);
#pragma unroll
    for (int s_ = 0; s_ < 9 * NCB; ++s_) {
        if (s_ + 1 < 9 * NCB) {
#pragma unroll
            for (int k = 0; k < 4; ++k) pv[(s_ + 1) & 1][k] = frag(s_ + 1, k >> 1, k & 1);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            acc[k >> 1][k & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s_ / NCB][s_ % NCB], pv[s_ & 1][k], acc[k >> 1][k & 1], 0, 0, 0);
    }
    if (fq == 0) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int gy = y0 + 2 * wid + r, gx = x0 + g * 16 + frow;
                if (gy >= H || gx >= W) continue;
                const size_t p = (size_t)gy * W + gx, P = (size_t)H * W;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    if (c >= cout) break;
                    const float v = acc[r][g][c] + bias[c];
                    if (y_f32_nchw) y_f32_nchw[(size_t)c * P + p] = v;
                    if (y_u8_hwc) y_u8_hwc[p * cout + c] = (uint8_t)fminf(fmaxf((v + 1.0f) * 127.5f, 0.0f), 255.0f);
                }
            }
    }
}

// ---------------------------------------------------------------- layout helpers
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ in, uint16_t* __restrict__ out, int C, int HW) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < C * HW; i += gridDim.x * blockDim.x) {
        const int c = i % C, p = i / C;
        out[i] = f32_to_bf16_bits(in[(size_t)c * HW + p]);
    }
}

__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ in, uint16_t* __restrict__ out,
                                                           int cols) {
    __shared__ float red[4];
    const float* r = in + (size_t)blockIdx.x * cols;
    uint16_t* o = out + (size_t)blockIdx.x * cols;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float mx = -INFINITY;
    for (int c = threadIdx.x; c < cols; c += 256) mx = fmaxf(mx, r[c]);
    mx = wave_max(mx);
    if (lane == 0) red[wid] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int c = threadIdx.x; c < cols; c += 256) s += __expf(r[c] - mx);
    s = wave_sum(s);
    if (lane == 0) red[wid] = s;
    __syncthreads();
    const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
    for (int c = threadIdx.x; c < cols; c += 256) o[c] = f32_to_bf16_bits(__expf(r[c] - mx) * inv);
}

// The same softmax with the row held in registers (cols <= 256 threads x 16 float4, cols % 4 == 0: the VAE mid-block attention's
// 16384 keys at 1024^2): one 16-byte read per 4 scores, one 8-byte write per 4 probabilities, every exponential computed once.
__global__ __launch_bounds__(256) void softmax_rows_reg_kernel(const float* __restrict__ in, uint16_t* __restrict__ out, int cols) {
    __shared__ float red[4];
    const float4* r = reinterpret_cast<const float4*>(in + (size_t)blockIdx.x * cols);
    uint2* o = reinterpret_cast<uint2*>(out + (size_t)blockIdx.x * cols);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int nvec = cols >> 2;
    float4 v[16];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = threadIdx.x + 256 * i;
        v[i] = c < nvec ? r[c] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        mx = fmaxf(mx, fmaxf(fmaxf(v[i].x, v[i].y), fmaxf(v[i].z, v[i].w)));
    }
    mx = wave_max(mx);
    if (lane == 0) red[wid] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        v[i].x = __expf(v[i].x - mx); v[i].y = __expf(v[i].y - mx); v[i].z = __expf(v[i].z - mx); v[i].w = __expf(v[i].w - mx);
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    s = wave_sum(s);
    if (lane == 0) red[wid] = s;
    __syncthreads();
    const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = threadIdx.x + 256 * i;
        if (c < nvec) o[c] = make_uint2(pack_bf16x2(v[i].x * inv, v[i].y * inv), pack_bf16x2(v[i].z * inv, v[i].w * inv));
    }
}

__global__ void transpose_bf16_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, int rows, int cols) {
    __shared__ uint16_t tile[32][34];
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

// OIHW -> O(HW)I with O zero-padded to O_pad rows (WeightLoader.swift:495-497 does OIHW->OHWI)
__global__ void oihw_to_ohwi_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, int O, int I, int KK,
                                    int O_pad, int I_pad) {
    const int total = O_pad * KK * I_pad;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int ci = i % I_pad, tap = (i / I_pad) % KK, o = i / (I_pad * KK);
        out[i] = (o < O && ci < I) ? in[((size_t)o * I + ci) * KK + tap] : (uint16_t)0;
    }
}

__global__ void nchw_to_nhwc_padded_kernel(const float* __restrict__ in, uint16_t* __restrict__ out, int C, int C_pad,
                                           int64_t P) {
    const int64_t total = P * C_pad;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C_pad);
        const int64_t p = i / C_pad;
        out[i] = c < C ? f32_to_bf16_bits(in[(size_t)c * P + p]) : (uint16_t)0;
    }
}

__global__ void nhwc_to_nchw_f32_kernel(const uint16_t* __restrict__ in, float* __restrict__ out, int C, int c0, int n,
                                        int64_t P) {
    const int64_t total = P * n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = i % P;
        const int c = (int)(i / P);
        out[i] = bf16_bits_to_f32(in[(size_t)p * C + c0 + c]);
    }
}

// w_phase[ph][o][a*2+b][c] = sum of w[o][ky][kx][c] over the taps (ky, kx) that read source pixel (a, b) of phase ph = py*2 + px:
// py = 0: a = 0 <- ky 0, a = 1 <- ky 1, 2;  py = 1: a = 0 <- ky 0, 1, a = 1 <- ky 2 (rows; the same rule for columns)
__global__ __launch_bounds__(256) void upsample_phase_weights_kernel(const uint16_t* __restrict__ w, int cout_pad, int cin,
                                                                     uint16_t* __restrict__ wp) {
    const long total = (long)4 * cout_pad * 4 * cin;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % cin);
        const int tap = (int)((i / cin) & 3);
        const int o = (int)((i / ((long)cin * 4)) % cout_pad);
        const int ph = (int)(i / ((long)cin * 4 * cout_pad));
        const int py = ph >> 1, px = ph & 1, ta = tap >> 1, tb = tap & 1;
        const int ky0 = py == 0 ? (ta == 0 ? 0 : 1) : (ta == 0 ? 0 : 2), ky1 = py == 0 ? (ta == 0 ? 0 : 2) : (ta == 0 ? 1 : 2);
        const int kx0 = px == 0 ? (tb == 0 ? 0 : 1) : (tb == 0 ? 0 : 2), kx1 = px == 0 ? (tb == 0 ? 0 : 2) : (tb == 0 ? 1 : 2);
        float s = 0.f;
        for (int ky = ky0; ky <= ky1; ++ky)
            for (int kx = kx0; kx <= kx1; ++kx) s += bf16_bits_to_f32(w[((size_t)o * 9 + ky * 3 + kx) * cin + c]);
        wp[i] = f32_to_bf16_bits(s);
    }
}

}  // namespace

// OHWI [cout_pad][3][3][cin] -> [cout_pad][3][seg]: the three taps of a kernel row (3 * cin values, contiguous in the zero-haloed NHWC
// image too) followed by zeros up to seg = a whole number of 64-deep K-tiles.  The GEMM convolution (gemm256s, GemmProblem::conv_*) then
// takes channel counts that are not a multiple of 64 (the small decoder's 96): a tap ROW is its K segment instead of a tap.
__global__ void conv_w_taprow_kernel(const uint16_t* __restrict__ w, int cout_pad, int cin, int seg, uint16_t* __restrict__ out) {
    const long total = (long)cout_pad * 3 * seg;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int k = (int)(i % seg);
        const long row = i / seg;                       // o * 3 + ky
        out[i] = k < 3 * cin ? w[row * 3 * cin + k] : (uint16_t)0;
    }
}

#define VK_DONE() return hipGetLastError() == hipSuccess ? 0 : 2

int launch_upsample_phase_weights(hipStream_t stream, const uint16_t* w, int cout_pad, int cin, uint16_t* w_phase) {
    if (!w || !w_phase || cout_pad <= 0 || cin <= 0) return 1;
    const long total = (long)16 * cout_pad * cin;
    hipLaunchKernelGGL(upsample_phase_weights_kernel, dim3((unsigned)std::min<long>((total + 255) / 256, 4096)), dim3(256), 0, stream, w, cout_pad, cin, w_phase);
    VK_DONE();
}

int launch_conv_w_taprow(hipStream_t stream, const uint16_t* w, int cout_pad, int cin, int seg, uint16_t* out) {
    if (!w || !out || cout_pad <= 0 || cin <= 0 || seg < 3 * cin || (seg & 63)) return 1;
    const long total = (long)cout_pad * 3 * seg;
    hipLaunchKernelGGL(conv_w_taprow_kernel, dim3((unsigned)std::min<long>((total + 255) / 256, 4096)), dim3(256), 0, stream, w, cout_pad, cin, seg, out);
    VK_DONE();
}

int launch_conv2d(hipStream_t stream, const ConvArgs& a) {
    if ((a.ks != 1 && a.ks != 3) || a.Cin % CBK != 0 || a.Cout <= 0 || a.Hin <= 0 || a.Win <= 0) return 1;
    if (a.downsample && (a.ks != 3 || a.upsample || (a.Hin & 1) || (a.Win & 1))) return 1;
    const int Ho = a.upsample ? 2 * a.Hin : a.downsample ? a.Hin / 2 : a.Hin;
    const int Wo = a.upsample ? 2 * a.Win : a.downsample ? a.Win / 2 : a.Win;
    const int P = Ho * Wo;
    const int tiles_m = (P + CBM - 1) / CBM;
    const int cout_pad = ((a.Cout + 31) / 32) * 32;
    // the heavy decoder / encoder convolutions: 128x128x64 tiles (FLUX2MI_CONV=32 keeps the first kernel, A/B runs)
    static const int conv_force = [] {
        const char* e = F2_DEV_ENV("FLUX2MI_CONV");
        return e ? atoi(e) : 0;
    }();
    if (conv_force != 32 && a.Cin % 32 == 0 && a.Cin >= 64 && a.Cout % 32 == 0 && a.Cout >= 96 && a.y && !a.y_f32_nchw && !a.y_u8_hwc &&
        (reinterpret_cast<uintptr_t>(a.y) & 7) == 0 && (!a.resid || (reinterpret_cast<uintptr_t>(a.resid) & 7) == 0) &&
        (!a.bias || (reinterpret_cast<uintptr_t>(a.bias) & 15) == 0)) {
        const int tm128 = (P + 127) / 128, tn128 = (a.Cout + 127) / 128;
        if (a.upsample && a.w_phase && a.ks == 3 && a.Cin % 64 == 0) {   // four 2x2 convolutions over the SOURCE pixels
            const int tms = (a.Hin * a.Win + 127) / 128;
            hipLaunchKernelGGL((conv2d_g128_kernel<64, true>), dim3(tms * tn128 * 4), dim3(256), 0, stream, a, a.Hin, a.Win, cout_pad);
            VK_DONE();
        }
        if (a.Cin % 64 == 0) hipLaunchKernelGGL(conv2d_g128_kernel<64>, dim3(tm128 * tn128), dim3(256), 0, stream, a, Ho, Wo, cout_pad);
        else hipLaunchKernelGGL(conv2d_g128_kernel<32>, dim3(tm128 * tn128), dim3(256), 0, stream, a, Ho, Wo, cout_pad);
        VK_DONE();
    }
    int bn;
    if (a.Cout % 128 == 0) bn = 128;
    else if (a.Cout % 96 == 0) bn = 96;
    else if (a.Cout % 64 == 0) bn = 64;
    else bn = 32;
    const int tiles_n = (a.Cout + bn - 1) / bn;
    dim3 grid(tiles_m * tiles_n), block(256);
    switch (bn) {
        case 128: hipLaunchKernelGGL(conv2d_kernel<128>, grid, block, 0, stream, a, Ho, Wo, cout_pad); break;
        case 96: hipLaunchKernelGGL(conv2d_kernel<96>, grid, block, 0, stream, a, Ho, Wo, cout_pad); break;
        case 64: hipLaunchKernelGGL(conv2d_kernel<64>, grid, block, 0, stream, a, Ho, Wo, cout_pad); break;
        default: hipLaunchKernelGGL(conv2d_kernel<32>, grid, block, 0, stream, a, Ho, Wo, cout_pad); break;
    }
    VK_DONE();
}

static inline int64_t gn_px_per_block(int64_t pixels) {
    int64_t ppb = (pixels + 1023) / 1024;
    if (ppb < 64) ppb = 64;
    return ppb;
}
int groupnorm_partial_floats(int64_t pixels, int groups) {
    const int64_t ppb = gn_px_per_block(pixels);
    return (int)(((pixels + ppb - 1) / ppb) * groups * 2);
}
int launch_groupnorm_stats(hipStream_t stream, const uint16_t* x, int64_t pixels, int C, int groups,
                           const float* gamma, const float* beta, float eps, float* partial, float* ab) {
    if (pixels <= 0 || C <= 0 || (C & 7) || groups <= 0 || groups > 64 || C % groups || (C >> 3) > 256) return 1;
    const int64_t ppb = gn_px_per_block(pixels);
    const int nblocks = (int)((pixels + ppb - 1) / ppb);
    hipLaunchKernelGGL(gn_partial_kernel, dim3(nblocks), dim3(256), 0, stream, x, pixels, C, groups, ppb, partial);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(1), dim3(256), 0, stream, partial, nblocks, pixels, C, groups, gamma,
                       beta, eps, ab);
    VK_DONE();
}
int launch_groupnorm_finalize_channels(hipStream_t stream, const float* part, int rows, int64_t pixels, int C, int groups,
                                       const float* gamma, const float* beta, float eps, float* ab) {
    if (!part || rows <= 0 || pixels <= 0 || C <= 0 || groups <= 0 || C % groups || C / groups > 256) return 1;
#define GNF_GO(CPG_) hipLaunchKernelGGL(gn_finalize_channels_kernel<CPG_>, dim3(groups), dim3(256), 0, stream, part, rows, pixels, C, groups, gamma, beta, eps, ab)
    switch (C / groups) {
        case 3: GNF_GO(3); break;
        case 4: GNF_GO(4); break;
        case 6: GNF_GO(6); break;
        case 8: GNF_GO(8); break;
        case 12: GNF_GO(12); break;
        case 16: GNF_GO(16); break;
        default: GNF_GO(0); break;
    }
#undef GNF_GO
    VK_DONE();
}
int launch_groupnorm_apply(hipStream_t stream, const uint16_t* x, uint16_t* y, int64_t pixels, int C,
                           const float* ab, int silu) {
    if (pixels <= 0 || (C & 7)) return 1;
    const int64_t nchunks = pixels * (C >> 3);
    int64_t blocks = (nchunks + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    // blocks * 256 must be a multiple of cpb = C/8 (see the kernel): round up to a multiple of cpb / gcd(cpb, 256)
    {
        int cpb = C >> 3, g = cpb, r = 256;
        while (r) { const int tmp = g % r; g = r; r = tmp; }
        const int64_t unit = cpb / g;
        blocks = (blocks + unit - 1) / unit * unit;
    }
    hipLaunchKernelGGL(gn_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, y, nchunks, C >> 3, ab, C,
                       silu);
    VK_DONE();
}
int launch_groupnorm_apply_padded(hipStream_t stream, const uint16_t* x, uint16_t* y_padded, int H, int W, int C, const float* ab, int silu) {
    if (H <= 0 || W <= 0 || (C & 7) || (C >> 3) > 256) return 1;
    const int cpb = C >> 3, ppi = 256 / cpb;
    const uint32_t total = (uint32_t)(H + 2) * (uint32_t)(W + 2);
    // ~8 pixels per thread, at least 4 blocks per CU's worth of work on a full chip
    uint32_t ppb = (uint32_t)ppi * 8u;
    while ((total + ppb - 1) / ppb > 16384u) ppb *= 2u;
    hipLaunchKernelGGL(gn_apply_padded_kernel, dim3((total + ppb - 1) / ppb), dim3(ppi * cpb), 0, stream, x, y_padded, H, W, cpb, ppi, ab, C, silu, ppb);
    VK_DONE();
}
// GroupNorm affine + SiLU + 3x3 convolution to <= 3 channels in one kernel (conv_out_fused_kernel).  1 = not this kernel's case.
int launch_conv_out_fused(hipStream_t stream, const uint16_t* x, const float* ab, const uint16_t* w, const float* bias, int H, int W, int C,
                          int cout, float* y_f32_nchw, uint8_t* y_u8_hwc) {
    if (!x || !ab || !w || !bias || H <= 0 || W <= 0 || cout < 1 || cout > 3 || (C != 128 && C != 96) || (!y_f32_nchw && !y_u8_hwc)) return 1;
    const dim3 grid((W + 31) / 32, (H + 7) / 8);
    if (C == 128) hipLaunchKernelGGL(conv_out_fused_kernel<128>, grid, dim3(256), 0, stream, x, ab, w, bias, H, W, cout, y_f32_nchw, y_u8_hwc);
    else hipLaunchKernelGGL(conv_out_fused_kernel<96>, grid, dim3(256), 0, stream, x, ab, w, bias, H, W, cout, y_f32_nchw, y_u8_hwc);
    VK_DONE();
}
int launch_nchw_f32_to_nhwc_bf16(hipStream_t stream, const float* in, uint16_t* out, int C, int H, int W) {
    const int total = C * H * W;
    if (total <= 0) return 1;
    int blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(blocks), dim3(256), 0, stream, in, out, C, H * W);
    VK_DONE();
}
int launch_softmax_rows(hipStream_t stream, const float* in, uint16_t* out, int rows, int cols) {
    if (rows <= 0 || cols <= 0) return 1;
    if ((cols & 3) == 0 && cols <= 256 * 64 && (reinterpret_cast<uintptr_t>(in) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 7) == 0)
        hipLaunchKernelGGL(softmax_rows_reg_kernel, dim3(rows), dim3(256), 0, stream, in, out, cols);
    else
        hipLaunchKernelGGL(softmax_rows_kernel, dim3(rows), dim3(256), 0, stream, in, out, cols);
    VK_DONE();
}
int launch_transpose_bf16(hipStream_t stream, const uint16_t* in, uint16_t* out, int rows, int cols) {
    if (rows <= 0 || cols <= 0) return 1;
    hipLaunchKernelGGL(transpose_bf16_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(32, 8), 0, stream, in,
                       out, rows, cols);
    VK_DONE();
}
int launch_oihw_to_ohwi_bf16(hipStream_t stream, const uint16_t* in, uint16_t* out, int O, int I, int KH, int KW,
                             int O_pad, int I_pad) {
    if (I_pad <= 0) I_pad = I;
    if (O <= 0 || I <= 0 || O_pad < O || I_pad < I) return 1;
    const int total = O_pad * KH * KW * I_pad;
    const int blocks = (total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256;
    hipLaunchKernelGGL(oihw_to_ohwi_kernel, dim3(blocks), dim3(256), 0, stream, in, out, O, I, KH * KW, O_pad, I_pad);
    VK_DONE();
}

int launch_nchw_f32_to_nhwc_bf16_padded(hipStream_t stream, const float* in, uint16_t* out, int C, int C_pad, int H, int W) {
    if (C <= 0 || C_pad < C || H <= 0 || W <= 0) return 1;
    const int64_t total = (int64_t)H * W * C_pad;
    const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL(nchw_to_nhwc_padded_kernel, dim3(blocks), dim3(256), 0, stream, in, out, C, C_pad, (int64_t)H * W);
    VK_DONE();
}

int launch_nhwc_bf16_to_nchw_f32(hipStream_t stream, const uint16_t* in, float* out, int C, int c0, int n, int64_t P) {
    if (C <= 0 || n <= 0 || c0 < 0 || c0 + n > C || P <= 0) return 1;
    const int64_t total = P * n;
    const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL(nhwc_to_nchw_f32_kernel, dim3(blocks), dim3(256), 0, stream, in, out, C, c0, n, P);
    VK_DONE();
}
