// gemm.hip — bf16 MFMA GEMM for gfx950 with the DiT's fused epilogues.
//
//   C[M,N] = A[M,K] (bf16, row-major, lda) x W[N,K]^T (bf16, row-major [out,in], ldw)
//
// All `Linear` layers of the reference are bias-free y = x W^T with W stored [out,in]
// (Transformer/Flux2Transformer.swift:70-111, Flux2Attention.swift:77-94,
// Flux2FeedForward.swift:56,99, Flux2ParallelAttention.swift:56-64), so both MFMA
// operands are K-contiguous and are staged identically.
//
// Structure (v1, "128x128x64, one barrier per K-tile"):
//   * 256 threads = 4 waves (2x2), each wave owns a 64x64 sub-tile = 4x4 MFMA
//     16x16x32 accumulators (64 f32 VGPRs).
//   * A and W tiles (128 rows x 64 k, 16 KB each) are double-buffered in LDS
//     (64 KB / workgroup -> 2 workgroups per CU -> 2 waves per SIMD).
//   * global -> register (16 B/lane, 128-B row segments) issued before the MFMA
//     block of the current tile and written to LDS after it (T14 split), so HBM/L2
//     latency hides under the MFMAs; one __syncthreads per K-tile.
//   * LDS rows are 128 B; the 16-B chunk index is XORed with (row>>1)&7 so that the
//     ds_read_b128 lane groups are conflict-free (cdna guide §5.5 T2).
//   * blockIdx -> tile mapping: XCD-aware (blocks b, b+8 share an L2) + grouped-M
//     ordering so neighbouring tiles share A/W panels in the same L2.
//   * up to two independent problems per launch (the txt and img streams of a
//     double block have different weights and row counts).
//
// Requirements: K % 64 == 0, rows 16-byte aligned (lda, ldw multiples of 8).
// M and N are arbitrary (loads are clamped, stores are masked).
#include <string.h>
#include "dev.h"
#include "kernels.h"
#include <stdlib.h>

int launch_gemm256s(hipStream_t stream, const GemmProblem* probs, int nprob);
bool gemm256s_eligible(const GemmProblem* probs, int nprob);

namespace {

constexpr int BM = 128, BN = 128, BK = 64;

__device__ __forceinline__ void tile_coords(int local, int ntiles, int tiles_m, int tiles_n, int& tm, int& tn) {
    // XCD remap (bijective for any ntiles): the blocks with equal (local % 8) share an
    // XCD under round-robin dispatch; give each such group a contiguous chunk of tiles.
    const int q = ntiles >> 3, r = ntiles & 7;
    const int xcd = local & 7, within = local >> 3;
    const int idx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
    // grouped-M ordering inside the chunk
    constexpr int GM = 8;
    const int per_group = GM * tiles_n;
    const int g = idx / per_group;
    const int first_m = g * GM;
    const int gsz = min(tiles_m - first_m, GM);
    const int in_g = idx - g * per_group;
    tm = first_m + in_g % gsz;
    tn = in_g / gsz;
}

__global__ __launch_bounds__(256, 2) void gemm_bf16_128x128_kernel(GemmBatch batch) {
    __shared__ __attribute__((aligned(16))) uint8_t smem[2 * 2 * BM * BK * 2];  // [buf][A|W][128][64] bf16 = 64 KB

    const int bid = blockIdx.x;
    const int pi = (batch.nprob > 1 && bid >= batch.tile_start[1]) ? 1 : 0;
    // the descriptor read where it lies in the kernel-argument segment (indexing the by-value copy would move the batch to scratch)
    using KProblem = const __attribute__((address_space(4))) GemmProblem;
    KProblem& p = ((KProblem*)__builtin_amdgcn_kernarg_segment_ptr())[pi];
    static_assert(offsetof(GemmBatch, p) == 0, "the problems are the first member of the only kernel argument");
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    // split-K (single-problem launches): workgroup = (K range `split`, tile `local`), split-major
    const int splits = batch.splits;
    const int split = splits > 1 ? bid / (tiles_m * tiles_n) : 0;
    const int local = splits > 1 ? bid - split * (tiles_m * tiles_n) : bid - (pi ? batch.tile_start[1] : 0);
    int tm, tn;
    tile_coords(local, tiles_m * tiles_n, tiles_m, tiles_n, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;

    const int t = threadIdx.x;
    const int lane = t & 63, wid = t >> 6;
    const int wm = wid >> 1, wn = wid & 1;

    // ---- global -> register staging assignment: 4 x 16 B per operand per thread
    const uint16_t* a_src[4];
    const uint16_t* w_src[4];
    int lds_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = t + 256 * i;
        const int row = idx >> 3, chunk = idx & 7;
        const int ar = min(m0 + row, p.M - 1);
        const int wr = min(n0 + row, p.N - 1);
        a_src[i] = p.A + (size_t)ar * p.lda + chunk * 8;
        w_src[i] = p.W + (size_t)wr * p.ldw + chunk * 8;
        lds_off[i] = row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
    }
    u32x4 a_reg[4], w_reg[4];
    // (macros, not lambdas: by-reference lambda captures keep the staging registers in scratch)
#define GEMM_LOAD_TILE(kt_)                                                              \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                   \
        a_reg[i_] = *reinterpret_cast<const u32x4*>(a_src[i_] + (kt_) * BK);             \
        w_reg[i_] = *reinterpret_cast<const u32x4*>(w_src[i_] + (kt_) * BK);             \
    }
#define GEMM_STORE_TILE(buf_)                                                            \
    {                                                                                    \
        uint8_t* base_ = smem + (buf_) * (2 * BM * BK * 2);                              \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                               \
            *reinterpret_cast<u32x4*>(base_ + lds_off[i_]) = a_reg[i_];                  \
            *reinterpret_cast<u32x4*>(base_ + BM * BK * 2 + lds_off[i_]) = w_reg[i_];    \
        }                                                                                \
    }

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment read addresses (byte offsets inside a [128][64] bf16 tile)
    const int frow = lane & 15, fq = lane >> 4;
    const int sw = (frow >> 1) & 7;  // row base is a multiple of 16, so (row>>1)&7 == (frow>>1)&7
    int a_off[4], w_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a_off[i] = (wm * 64 + i * 16 + frow) * 128;
        w_off[i] = (wn * 64 + i * 16 + frow) * 128;
    }

    const int nk_all = p.K / BK;
    const int kt0 = (int)((long)nk_all * split / splits), nk = (int)((long)nk_all * (split + 1) / splits);   // this workgroup's K-tiles [kt0, nk)
    GEMM_LOAD_TILE(kt0)
    GEMM_STORE_TILE(kt0 & 1)
    __syncthreads();

    for (int kt = kt0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) { GEMM_LOAD_TILE(kt + 1) }
        const uint8_t* sa = smem + buf * (2 * BM * BK * 2);
        const uint8_t* sb = sa + BM * BK * 2;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int coff = (((kk * 4 + fq) ^ sw) << 4);
            bf16x8 af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i] = *reinterpret_cast<const bf16x8*>(sa + a_off[i] + coff);
                bf[i] = *reinterpret_cast<const bf16x8*>(sb + w_off[i] + coff);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) GEMM_STORE_TILE(buf ^ 1)
        __syncthreads();
    }

    // ---- epilogue.  acc[i][j][r]: row = m0 + wm*64 + i*16 + fq*4 + r, col = n0 + wn*64 + j*16 + frow
    const int row_base = m0 + wm * 64 + fq * 4;
    const int col_base = n0 + wn * 64 + frow;
    int epi = p.epi;
    if (epi == EPI_QKV_MLP) epi = (n0 >= p.n_split) ? EPI_SWIGLU : EPI_BF16;

    if (splits > 1) {   // the partial accumulator of this K range: ws[split][row][col] f32 (gemm_splitk_reduce_kernel finishes the job)
        float* C = p.splitk_ws + (size_t)split * p.M * p.N;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row_base + i * 16 + r;
                if (row < p.M) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int col = col_base + j * 16;
                        if (col < p.N) C[(size_t)row * p.N + col] = acc[i][j][r];
                    }
                }
            }
    } else if (epi == EPI_F32) {
        float* C = reinterpret_cast<float*>(p.C);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row_base + i * 16 + r;
                if (row < p.M) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int col = col_base + j * 16;
                        if (col < p.N) C[(size_t)row * p.ldc + col] = acc[i][j][r] * p.alpha;
                    }
                }
            }
    } else if (epi == EPI_BF16) {
        uint16_t* C = reinterpret_cast<uint16_t*>(p.C);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row_base + i * 16 + r;
                if (row < p.M) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int col = col_base + j * 16;
                        if (col < p.N) C[(size_t)row * p.ldc + col] = f32_to_bf16_bits(acc[i][j][r] * p.alpha);
                    }
                }
            }
    } else if (epi == EPI_GATE_RESID) {
        // hs[row, col] += gate[col] * y  (applyGate + residual, Flux2Modulation.swift:119-122)
        float* C = reinterpret_cast<float*>(p.C);
        float g[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = col_base + j * 16;
            g[j] = (col < p.N) ? p.vec[col] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row_base + i * 16 + r;
                if (row < p.M) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int col = col_base + j * 16;
                        if (col < p.N) {
                            float* dst = C + (size_t)row * p.ldc + col;
                            *dst = *dst + g[j] * acc[i][j][r];
                        }
                    }
                }
            }
    } else if (epi == EPI_SWIGLU) {
        // weight rows were interleaved at load time in blocks of 32: inside every aligned
        // group of 64 output columns, cols 0..31 are `gate` and 32..63 the matching `value`
        // rows (Flux2FeedForward.swift:59-67: gate = first half, value = second half).
        uint16_t* C = reinterpret_cast<uint16_t*>(p.C2);
        const int start = (p.epi == EPI_QKV_MLP) ? p.n_split : 0;
        const int hbase = ((n0 + wn * 64 - start) >> 1) + frow;
        const int hidden = (p.N - start) >> 1;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row_base + i * 16 + r;
                if (row < p.M) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int h = hbase + j * 16;
                        if (h < hidden) {
                            const float gate = acc[i][j][r], val = acc[i][j + 2][r];
                            C[(size_t)row * p.ldc2 + h] = f32_to_bf16_bits(silu_f(gate) * val);
                        }
                    }
                }
            }
    } else {  // EPI_BIAS_BF16: C = bf16(alpha*acc + bias[col] (+ resid[row,col]))
        uint16_t* C = reinterpret_cast<uint16_t*>(p.C);
        float bsv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = col_base + j * 16;
            bsv[j] = (p.vec != nullptr && col < p.N) ? p.vec[col] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row_base + i * 16 + r;
                if (row < p.M) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int col = col_base + j * 16;
                        if (col < p.N) {
                            float v = acc[i][j][r] * p.alpha + bsv[j];
                            if (p.resid) v += bf16_bits_to_f32(p.resid[(size_t)row * p.ldr + col]);
                            C[(size_t)row * p.ldc + col] = f32_to_bf16_bits(v);
                        }
                    }
                }
            }
    }
}

// Split-K, second half: the partial accumulators of a tile's K ranges are added in split order (fixed: bit-reproducible) and the
// problem's epilogue is applied — the same arithmetic as the one-pass epilogues above on the summed accumulator.
__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(GemmProblem p, int splits) {
    const long total = (long)p.M * p.N;
    const size_t plane = (size_t)p.M * p.N;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int row = (int)(idx / p.N), col = (int)(idx - (long)row * p.N);
        float acc = p.splitk_ws[idx];
        for (int s = 1; s < splits; ++s) acc += p.splitk_ws[(size_t)s * plane + idx];
        if (p.epi == EPI_F32) {
            reinterpret_cast<float*>(p.C)[(size_t)row * p.ldc + col] = acc * p.alpha;
        } else if (p.epi == EPI_BF16) {
            reinterpret_cast<uint16_t*>(p.C)[(size_t)row * p.ldc + col] = f32_to_bf16_bits(acc * p.alpha);
        } else if (p.epi == EPI_GATE_RESID) {
            float* dst = reinterpret_cast<float*>(p.C) + (size_t)row * p.ldc + col;
            *dst = *dst + p.vec[col] * acc;
        } else {   // EPI_BIAS_BF16
            float v = acc * p.alpha + (p.vec ? p.vec[col] : 0.f);
            if (p.resid) v += bf16_bits_to_f32(p.resid[(size_t)row * p.ldr + col]);
            reinterpret_cast<uint16_t*>(p.C)[(size_t)row * p.ldc + col] = f32_to_bf16_bits(v);
        }
    }
}

// The same reduction, four columns per thread (N % 4 == 0, rows and planes 16-byte aligned): the form behind the 256x256 kernel's
// split-K (launch_gemm), whose workspaces are tens of MB.  Same order of additions as the scalar form: bit-identical results.
__global__ __launch_bounds__(256) void gemm_splitk_reduce4_kernel(GemmProblem p, int splits) {
    const int n4 = p.N >> 2;
    const long total = (long)p.M * n4;
    const size_t plane = (size_t)p.M * p.N;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int row = (int)(idx / n4), col = (int)(idx - (long)row * n4) * 4;
        const size_t e = (size_t)row * p.N + col;
        f32x4 acc = *reinterpret_cast<const f32x4*>(p.splitk_ws + e);
        for (int s = 1; s < splits; ++s) acc += *reinterpret_cast<const f32x4*>(p.splitk_ws + (size_t)s * plane + e);
        if (p.epi == EPI_F32) {
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + (size_t)row * p.ldc + col) = acc * p.alpha;
        } else if (p.epi == EPI_BF16) {
            const f32x4 v = acc * p.alpha;
            *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p.C) + (size_t)row * p.ldc + col) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
        } else if (p.epi == EPI_GATE_RESID) {
            f32x4* dst = reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + (size_t)row * p.ldc + col);
            const f32x4 g = *reinterpret_cast<const f32x4*>(p.vec + col);
            *dst = *dst + g * acc;
        } else {   // EPI_BIAS_BF16
            f32x4 v = acc * p.alpha;
            if (p.vec) v += *reinterpret_cast<const f32x4*>(p.vec + col);
            if (p.resid) {
                const uint2 r = *reinterpret_cast<const uint2*>(p.resid + (size_t)row * p.ldr + col);
                v += f32x4{lo_bf16(r.x), hi_bf16(r.x), lo_bf16(r.y), hi_bf16(r.y)};
            }
            *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p.C) + (size_t)row * p.ldc + col) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
        }
    }
}

// Fallback for shapes the MFMA kernel does not take (K % 64 != 0), e.g. the VAE's
// 32->32 post_quant_conv.  One thread per output element, f32 accumulate.
__global__ void gemm_naive_kernel(GemmProblem p) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= p.M * p.N) return;
    const int row = idx / p.N, col = idx % p.N;
    float acc = 0.f;
    for (int k = 0; k < p.K; ++k)
        acc += bf16_bits_to_f32(p.A[(size_t)row * p.lda + k]) * bf16_bits_to_f32(p.W[(size_t)col * p.ldw + k]);
    acc *= p.alpha;
    if (p.epi == EPI_F32) {
        reinterpret_cast<float*>(p.C)[(size_t)row * p.ldc + col] = acc;
    } else if (p.epi == EPI_BF16) {
        reinterpret_cast<uint16_t*>(p.C)[(size_t)row * p.ldc + col] = f32_to_bf16_bits(acc);
    } else if (p.epi == EPI_BIAS_BF16) {
        float v = acc + (p.vec ? p.vec[col] : 0.f);
        if (p.resid) v += bf16_bits_to_f32(p.resid[(size_t)row * p.ldr + col]);
        reinterpret_cast<uint16_t*>(p.C)[(size_t)row * p.ldc + col] = f32_to_bf16_bits(v);
    }
}

}  // namespace

// K ranges per tile for the 128x128 kernel: only when its tiles would fill less than half of the 512 workgroup slots (2 per CU x 256)
// and every range keeps >= 8 K-tiles; as many ranges as fit the slots, at most 8.
static int gemm_splitk_plan(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K % BK != 0) return 1;
    const long tiles = (long)((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    const int nk = K / BK;
    if (tiles > 256 || nk < 16) return 1;
    long s = 512 / tiles;
    if (s > 8) s = 8;
    if (s > nk / 8) s = nk / 8;
    return s < 2 ? 1 : (int)s;
}
// ---- split-K on the 256x256 kernel (gemm256s.hip): launches whose 256 x 256 tiles cover at most half of the 256 CUs and whose K is long
// — the N = d GEMMs of the blocks at small sequence lengths: linear2 at 256^2 is 36 tiles (of K 12288) on 256 CUs.  The K-tiles of every
// problem are cut into `s` ranges; each range is a problem of its own in the batch (A / W advanced along K, plain f32 output into its
// plane of the workspace), one launch fills the chip, and the reduce kernel adds a problem's planes in range order and applies its real
// epilogue: deterministic, no atomics.  Unchanged (s = 1) for every launch with more than 128 tiles: the 1024^2 shapes of every model.
static bool splitk_epi_ok(int epi) { return epi == EPI_F32 || epi == EPI_BF16 || epi == EPI_GATE_RESID || epi == EPI_BIAS_BF16; }
static int gemm256s_splitk_plan(const GemmProblem* probs, int nprob, bool need_ws) {
    static const bool on = [] {
        const char* e = F2_DEV_ENV("FLUX2MI_GEMM_SPLITK");
        return e ? atoi(e) != 0 : true;
    }();
    if (!on || nprob < 1 || nprob > 2) return 1;
    long tiles = 0;
    int nk_min = 1 << 30;
    for (int i = 0; i < nprob; ++i) {
        const GemmProblem& q = probs[i];
        const int kt = q.fp8 ? 128 : 64;
        if (q.M <= 0 || q.N <= 0 || q.K <= 0 || q.K % kt != 0 || !splitk_epi_ok(q.epi) || q.qk_qw || q.qk8 || q.col_perm16 || q.c2_q8) return 1;
        if ((q.fp8 != 0) != (probs[0].fp8 != 0)) return 1;
        if (q.fp8 && (q.epi == EPI_BF16 || q.epi == EPI_BIAS_BF16 || (!q.a_scale && !q.a_bscale) || !q.w_scale)) return 1;   // e4m3: the x += gate y and f32 launches
        if ((q.N & 3) || (q.ldc & 3) || (reinterpret_cast<uintptr_t>(q.C) & 15) || (q.vec && (reinterpret_cast<uintptr_t>(q.vec) & 15)) ||
            (q.resid && ((q.ldr & 3) || (reinterpret_cast<uintptr_t>(q.resid) & 7))))
            return 1;
        tiles += (long)((q.M + 255) / 256) * ((q.N + 255) / 256);
        nk_min = q.K / kt < nk_min ? q.K / kt : nk_min;
    }
    if (tiles > 128 || nk_min < 16) return 1;
    int s = GEMM_MAX_PROBS / nprob;
    if (s > 256 / tiles) s = (int)(256 / tiles);
    if (s > nk_min / 8) s = nk_min / 8;
    if (need_ws)
        for (int i = 0; i < nprob; ++i) {
            const GemmProblem& q = probs[i];
            if (!q.splitk_ws || (reinterpret_cast<uintptr_t>(q.splitk_ws) & 15)) return 1;
            const size_t plane = (size_t)q.M * q.N * sizeof(float);
            if (q.splitk_ws_bytes < (size_t)s * plane) s = (int)(q.splitk_ws_bytes / plane);
        }
    return s < 2 ? 1 : s;
}
// Tail split (launch_gemm): a single x += gate y problem whose 256-row tiles overflow ONE round of the chip by a little (Klein-9B's N = 4096
// launches at S 4608: 18 x 16 = 288 tiles) would run as 160-row tiles on two rounds (464 tiles; for e4m3 operands those are bound by the
// L2 -> LDS stream at 0.70-0.76 of MFMA issue).  Instead: the first floor(256 / tiles_n) row tiles as one launch of exactly one round of
// 256-row tiles (0.90 of MFMA issue), the remaining rows as a split-K launch (above).  Returns the rows of the first launch, 0 = no split.
static int gemm_tail_split_rows(int M, int N, int K, bool fp8) {
    static const bool on = [] {
        const char* e = F2_DEV_ENV("FLUX2MI_GEMM_TAIL");
        return e ? atoi(e) != 0 : true;
    }();
    // bf16 only: with e4m3 operands the K ranges' f32 planes and the extra launches eat what the taller tiles give (measured on Klein-9B,
    // same box: bf16 57.46 -> 56.17 ms per step, e4m3 35.30 -> 35.40: profiles/r06_p_tail_split_ab_same_box.txt)
    if (!on || fp8 || M <= 0 || N <= 0 || (N & 255) || K / 64 < 32) return 0;
    const long tiles_n = N / 256, tiles = (long)((M + 255) / 256) * tiles_n;
    if (tiles <= 256 || tiles > 256 + 48) return 0;
    const long rows_a = (256 / tiles_n) * 256;
    if (rows_a <= 0 || rows_a >= M) return 0;
    if ((long)((M - rows_a + 255) / 256) * tiles_n > 48) return 0;
    return (int)rows_a;
}
size_t gemm_splitk_ws_bytes(int M, int N, int K, bool fp8) {
    size_t need = 0;
    if (!fp8) {
        const int s = gemm_splitk_plan(M, N, K);
        need = s > 1 ? (size_t)s * M * N * sizeof(float) : 0;
    }
    // the 256x256 kernel's split-K: what a single-problem launch of this shape would take (a two-problem launch takes at most as much per problem)
    GemmProblem q;
    memset((void*)&q, 0, sizeof q);
    q.M = M; q.N = N; q.K = K; q.epi = EPI_F32; q.ldc = N; q.fp8 = fp8 ? 1 : 0;
    static const float one = 1.0f;
    if (fp8) { q.a_scale = &one; q.w_scale = &one; }   // (presence only: the plan does not read them)
    const int rows_a = gemm_tail_split_rows(M, N, K, fp8);
    if (rows_a) q.M = M - rows_a;                       // the tail split's second launch is what splits K
    const int s2 = gemm256s_splitk_plan(&q, 1, false);
    if (s2 > 1 && (size_t)s2 * q.M * N * sizeof(float) > need) need = (size_t)s2 * q.M * N * sizeof(float);
    return need;
}

int launch_gemm(hipStream_t stream, const GemmProblem* probs, int nprob);
// split-K and the tail split on the 256x256 kernel (comments at gemm256s_splitk_plan / gemm_tail_split_rows): -1 = neither applies
static int launch_gemm_splitk_or_tail(hipStream_t stream, const GemmProblem* probs, int nprob) {
    // ---- tail split: x += gate y launches that overflow one round of 256-row tiles by a little
    bool all_gr = true;
    for (int i = 0; i < nprob; ++i) all_gr = all_gr && probs[i].epi == EPI_GATE_RESID && !probs[i].qk_qw && (probs[i].N & 255) == 0;
    if (all_gr && nprob == 2) {
        // the text / image pair of a double block: when the larger problem is one round on its own, the two go out as two launches
        // (the smaller one then splits K below)
        const long t0 = (long)((probs[0].M + 255) / 256) * (probs[0].N / 256), t1 = (long)((probs[1].M + 255) / 256) * (probs[1].N / 256);
        const int big = t0 >= t1 ? 0 : 1;
        const long tb = big ? t1 : t0, tsm = big ? t0 : t1;
        static const bool on = [] {
            const char* e = F2_DEV_ENV("FLUX2MI_GEMM_TAIL");
            return e ? atoi(e) != 0 : true;
        }();
        if (on && !probs[0].fp8 && !probs[1].fp8 && tb + tsm > 256 && tb <= 256 && tb >= 224 && tsm <= 48 && probs[big].K / 64 >= 32 && probs[1 - big].splitk_ws) {
            const int r = launch_gemm(stream, &probs[big], 1);
            return r ? r : launch_gemm(stream, &probs[1 - big], 1);
        }
    }
    if (all_gr && nprob == 1 && probs[0].splitk_ws) {
        const GemmProblem& q = probs[0];
        const int rows_a = gemm_tail_split_rows(q.M, q.N, q.K, q.fp8 != 0);
        if (rows_a) {
            const size_t esz = q.fp8 ? 1 : 2;
            GemmProblem a = q, rr = q;
            a.M = rows_a; a.splitk_ws = nullptr; a.splitk_ws_bytes = 0;
            rr.M = q.M - rows_a;
            rr.A = reinterpret_cast<const uint16_t*>(reinterpret_cast<const uint8_t*>(q.A) + (size_t)rows_a * q.lda * esz);
            rr.C = reinterpret_cast<float*>(q.C) + (size_t)rows_a * q.ldc;
            if (q.a_scale) rr.a_scale = q.a_scale + rows_a;
            if (q.a_bscale) rr.a_bscale = q.a_bscale + rows_a / 16;
            const int r = launch_gemm(stream, &a, 1);
            return r ? r : launch_gemm(stream, &rr, 1);
        }
    }
    // ---- split-K: few tiles, long K
    const int sk = gemm256s_splitk_plan(probs, nprob, true);
    if (sk <= 1) return -1;
    GemmProblem sub[GEMM_MAX_PROBS];
    int ns = 0;
    for (int i = 0; i < nprob; ++i) {
        const GemmProblem& q = probs[i];
        const int kt = q.fp8 ? 128 : 64;
        const size_t esz = q.fp8 ? 1 : 2;
        const int nk = q.K / kt, chunk = (nk + sk - 1) / sk;
        for (int j = 0; j < sk; ++j) {
            const int k0 = j * chunk, k1 = (j + 1) * chunk < nk ? (j + 1) * chunk : nk;
            GemmProblem& t = sub[ns++];
            memset((void*)&t, 0, sizeof t);
            t.A = reinterpret_cast<const uint16_t*>(reinterpret_cast<const uint8_t*>(q.A) + (size_t)k0 * kt * esz); t.lda = q.lda;
            t.W = reinterpret_cast<const uint16_t*>(reinterpret_cast<const uint8_t*>(q.W) + (size_t)k0 * kt * esz); t.ldw = q.ldw;
            t.C = q.splitk_ws + (size_t)j * q.M * q.N; t.ldc = q.N;
            t.M = q.M; t.N = q.N; t.K = (k1 > k0 ? k1 - k0 : 0) * kt;
            t.epi = EPI_F32; t.alpha = 1.0f;
            // e4m3 operands: the partial accumulators leave with their scales applied (per row / per column factors, or the block scales
            // inside the MFMA: one table row per K-tile) — the planes then add like the bf16 ones
            t.fp8 = q.fp8; t.a_scale = q.a_scale; t.w_scale = q.w_scale;
            t.a_bscale = q.a_bscale ? q.a_bscale + (size_t)k0 * q.a_bscale_ld : nullptr; t.a_bscale_ld = q.a_bscale_ld;
        }
    }
    bool ok = gemm256s_eligible(sub, ns);
    for (int i = 0; i < ns; ++i) ok = ok && sub[i].K > 0;
    if (!ok) return -1;
    const int r = launch_gemm256s(stream, sub, ns);
    if (r != 0) return r;
    for (int i = 0; i < nprob; ++i) {
        const long total4 = (long)probs[i].M * (probs[i].N >> 2);
        const unsigned blocks = (unsigned)((total4 + 255) / 256 > 4096 ? 4096 : (total4 + 255) / 256);
        hipLaunchKernelGGL(gemm_splitk_reduce4_kernel, dim3(blocks), dim3(256), 0, stream, probs[i], sk);
    }
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

int launch_gemm(hipStream_t stream, const GemmProblem* probs, int nprob) {
    if (nprob < 1 || nprob > 2) return 1;
    if (probs[0].epi == EPI_CONV)   // implicit-GEMM convolutions of the VAE: only the one-wave-per-SIMD kernel has the form
        return gemm256s_eligible(probs, nprob) ? launch_gemm256s(stream, probs, nprob) : 1;
    bool mfma_ok = true;
    int n_fp8 = 0;
    for (int i = 0; i < nprob; ++i) n_fp8 += probs[i].fp8 ? 1 : 0;
    if (n_fp8) {  // e4m3 operands: only the 256x256 kernel, no fallbacks
        if (n_fp8 != nprob) return 1;
        for (int i = 0; i < nprob; ++i) {
            const GemmProblem& p = probs[i];
            if (p.M <= 0 || p.N <= 0 || p.K <= 0 || p.K % 128 != 0 || (p.lda & 15) || (p.ldw & 15) || (!p.a_scale && !p.a_bscale) || !p.w_scale) return 1;
            // block-scaled A (kernels.h): the table is read by 4-byte aligned scalar loads, up to 12 bytes behind a tile's first row block
            if (p.a_bscale && ((reinterpret_cast<uintptr_t>(p.a_bscale) & 3) || (p.a_bscale_ld & 3) || p.a_bscale_ld < (p.M + 15) / 16 + 12)) return 1;
            if ((p.epi == EPI_SWIGLU || p.epi == EPI_QKV_MLP) && (p.N % 64 != 0 || p.n_split % 256 != 0)) return 1;
            // SwiGLU hidden as block-scaled e4m3: whole 256-column tiles (= 128 hidden columns, one scale block), 8-byte runs
            if (p.c2_q8 && ((p.epi != EPI_SWIGLU && p.epi != EPI_QKV_MLP) || !p.c2_bs || (p.N - (p.epi == EPI_QKV_MLP ? p.n_split : 0)) % 256 != 0 ||
                            (p.c2_q8_ld & 7) || (reinterpret_cast<uintptr_t>(p.c2_q8) & 7)))
                return 1;
        }
        {
            const int r = launch_gemm_splitk_or_tail(stream, probs, nprob);
            if (r >= 0) return r;
        }
        // the one-wave-per-SIMD kernel (gemm256s.hip)
        return gemm256s_eligible(probs, nprob) ? launch_gemm256s(stream, probs, nprob) : 1;
    }
    for (int i = 0; i < nprob; ++i) {
        const GemmProblem& p = probs[i];
        if (p.M <= 0 || p.N <= 0 || p.K <= 0) return 1;
        if (p.K % BK != 0 || (p.lda & 7) || (p.ldw & 7)) mfma_ok = false;
        if ((p.epi == EPI_SWIGLU || p.epi == EPI_QKV_MLP) && (p.N % 64 != 0 || p.n_split % BN != 0)) return 1;
    }
    if (!mfma_ok) {
        for (int i = 0; i < nprob; ++i) {
            const GemmProblem& p = probs[i];
            if (p.epi != EPI_F32 && p.epi != EPI_BF16 && p.epi != EPI_BIAS_BF16) return 1;
            const int total = p.M * p.N;
            hipLaunchKernelGGL(gemm_naive_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, p);
        }
        return hipGetLastError() == hipSuccess ? 0 : 2;
    }
    {
        const int r = launch_gemm_splitk_or_tail(stream, probs, nprob);
        if (r >= 0) return r;
    }
    // large problems -> 256x256 LDS-DMA kernel (gemm256s.hip); FLUX2MI_GEMM=128|256 forces a kernel
    {
        static const int force = [] {
            const char* e = F2_DEV_ENV("FLUX2MI_GEMM");
            return e ? atoi(e) : 0;
        }();
        long tiles256 = 0;
        for (int i = 0; i < nprob; ++i)
            tiles256 += (long)((probs[i].M + 255) / 256) * ((probs[i].N + 255) / 256);
        bool big = tiles256 >= 100;
        for (int i = 0; i < nprob; ++i)
            if (probs[i].epi == EPI_QKV_MLP && probs[i].n_split % 256 != 0) big = false;  // a 256-wide tile must not straddle the split
        bool fused_qk = false;   // only the 256x256 kernel (gemm256s.hip) applies GemmProblem::qk_*
        for (int i = 0; i < nprob; ++i) fused_qk |= probs[i].qk_qw != nullptr || probs[i].col_perm16 != 0;
        if (fused_qk && force == 128) return 1;
        if (fused_qk) big = true;   // the caller decides fusion per model and sequence, not per launch (dit.cpp)
        if (force == 256 || (force != 128 && big))
        {
            // the one-wave-per-SIMD kernel (gemm256s.hip)
            if (gemm256s_eligible(probs, nprob)) return launch_gemm256s(stream, probs, nprob);
            if (fused_qk) return 1;   // source offsets beyond 32 bits: only the 128x128 kernel below, which has no fused epilogue
        }
    }
    GemmBatch b;
    b.nprob = nprob;
    int tiles[2] = {0, 0};
    for (int i = 0; i < nprob; ++i) {
        b.p[i] = probs[i];
        tiles[i] = ((probs[i].M + BM - 1) / BM) * ((probs[i].N + BN - 1) / BN);
    }
    for (int i = nprob; i < GEMM_MAX_PROBS; ++i) b.p[i] = probs[0];
    for (int i = 0; i < GEMM_MAX_PROBS; ++i) b.tile_start[i] = tiles[0] + tiles[1];
    b.tile_start[0] = 0;
    b.tile_start[1] = tiles[0];
    b.splits = 1;
    if (nprob == 1 && probs[0].splitk_ws != nullptr) {
        const GemmProblem& q = probs[0];
        const int s = gemm_splitk_plan(q.M, q.N, q.K);
        const bool epi_ok = q.epi == EPI_F32 || q.epi == EPI_BF16 || q.epi == EPI_GATE_RESID || q.epi == EPI_BIAS_BF16;
        if (s > 1 && epi_ok && q.splitk_ws_bytes >= (size_t)s * q.M * q.N * sizeof(float)) b.splits = s;
    }
    hipLaunchKernelGGL(gemm_bf16_128x128_kernel, dim3((tiles[0] + tiles[1]) * b.splits), dim3(256), 0, stream, b);
    if (b.splits > 1) {
        const long total = (long)probs[0].M * probs[0].N;
        hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256)), dim3(256), 0, stream, probs[0], b.splits);
    }
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
