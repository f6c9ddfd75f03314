// gemm_ws.hip — wave-specialised bf16 GEMM: 256x128x64 tile, 4 consumer waves (MFMA only) +
// 4 loader waves (LDS-DMA only).  Same contract / epilogues as gemm.hip.
//
// Why (measured on the 8-wave 256x256 kernel): a wave that issues its own
// global->LDS traffic queues ~114 cycles per 1-KB piece behind the other waves' pieces on the
// CU's vector-memory path, in order, in front of its own MFMAs; per 64-deep K-tile that is ~900
// issue cycles next to 1024 MFMA cycles.  Here the waves that feed the matrix pipe never touch
// the vector-memory queue inside the K loop: waves 4-7 do nothing but LDS-DMA (they need ~20
// registers and sit blocked on the queue for free), waves 0-3 — one per SIMD, wave tile 128x64 —
// only ds_read and MFMA.
//
// LDS: ring of 3 stages x [A 256 rows | W 128 rows] x 64 k bf16 = 3 x 48 KB.  One s_barrier per
// K-tile for all 8 waves:
//   loader  it. t:  DMA tile t+2 -> stage (t+2)%3 ; s_waitcnt vmcnt(12) [tile t+1 landed] ; barrier
//   consumer it. t: fragments of tile t <- stage t%3 ; 64 MFMA ; lgkmcnt(0) ; barrier
// (stage (t+2)%3 held tile t-1, whose reads retired before the barrier that ended iteration t-1.)
#include "../dev.h"
#include "../kernels.h"
#include <stdlib.h>

namespace {

constexpr int WM = 256, WN = 128, WK = 64;
constexpr int A_BYTES = WM * WK * 2;          // 32 KB
constexpr int B_BYTES = WN * WK * 2;          // 16 KB
constexpr int ST_BYTES = A_BYTES + B_BYTES;   // 48 KB
constexpr int NST = 3;

__device__ __forceinline__ void tile_coords_ws(int local, int ntiles, int tiles_m, int tiles_n, int& tm, int& tn) {
    const int q = ntiles >> 3, r = ntiles & 7;
    const int xcd = local & 7, within = local >> 3;
    const int idx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
    constexpr int GM = 6;
    const int per_group = GM * tiles_n;
    const int g = idx / per_group;
    const int first_m = g * GM;
    const int gsz = min(tiles_m - first_m, GM);
    const int in_g = idx - g * per_group;
    tm = first_m + in_g % gsz;
    tn = in_g / gsz;
}

__global__ __launch_bounds__(512, 2) void gemm_bf16_ws_kernel(GemmBatch batch) {
    __shared__ __attribute__((aligned(16))) uint8_t smem[NST * ST_BYTES];  // 144 KB, the only LDS object

    const int bid = blockIdx.x;
    const int pi = (batch.nprob > 1 && bid >= batch.tile_start1) ? 1 : 0;
    const GemmProblem& p = batch.p[pi];
    const int local = bid - (pi ? batch.tile_start1 : 0);
    const int tiles_m = (p.M + WM - 1) / WM, tiles_n = (p.N + WN - 1) / WN;
    int tm, tn;
    tile_coords_ws(local, tiles_m * tiles_n, tiles_m, tiles_n, tm, tn);
    const int m0 = tm * WM, n0 = tn * WN;

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wid = __builtin_amdgcn_readfirstlane(t >> 6);  // provably wave-uniform role
    const int nk = p.K / WK;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem;

    if (wid >= 4) {
        // ================= loader waves: 12 LDS-DMA pieces (1 KB each) per K-tile and wave
        const int lw = wid - 4;
        // piece q of the A part covers rows 8q..8q+7 (lane -> row 8q + (lane>>3), LDS chunk lane&7
        // holds global chunk (lane&7) ^ ((row>>1)&7)); this wave owns A pieces 8lw..8lw+7 and
        // W pieces 4lw..4lw+3.
        uint32_t voff[12];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = (lw * 8 + j) * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ ((row >> 1) & 7);
            voff[j] = (uint32_t)(((size_t)min(m0 + row, p.M - 1) * p.lda + chunk * 8) * 2);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = (lw * 4 + j) * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ ((row >> 1) & 7);
            voff[8 + j] = (uint32_t)(((size_t)min(n0 + row, p.N - 1) * p.ldw + chunk * 8) * 2);
        }
        const uint8_t* gA = reinterpret_cast<const uint8_t*>(p.A);
        const uint8_t* gW = reinterpret_cast<const uint8_t*>(p.W);
        const uint32_t m0A = lds_base + lw * 8192;             // + j*1024 + stage
        const uint32_t m0W = lds_base + A_BYTES + lw * 4096;   // + j*1024 + stage
#define WS_DMA(voff_, sb_, m0_)                                                                       \
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2" ::"v"(voff_), "s"(m0_), "s"(sb_) : "memory");
#define WS_ISSUE(u_)                                                                  \
    {                                                                                 \
        const int u__ = (u_);                                                         \
        if (u__ < nk) {                                                               \
            const uint32_t so__ = (uint32_t)((u__ % NST) * ST_BYTES);                 \
            const uint8_t* a__ = gA + (size_t)u__ * (WK * 2);                         \
            const uint8_t* w__ = gW + (size_t)u__ * (WK * 2);                         \
            _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) WS_DMA(voff[j_], a__, m0A + so__ + j_ * 1024)      \
            _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) WS_DMA(voff[8 + j_], w__, m0W + so__ + j_ * 1024)  \
        }                                                                             \
    }
        WS_ISSUE(0)
        WS_ISSUE(1)
        if (nk > 1) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // tile 0 visible
        for (int kt = 0; kt < nk; ++kt) {
            WS_ISSUE(kt + 2)
            if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
#undef WS_DMA
#undef WS_ISSUE
        return;
    }

    // ================= consumer waves: 2 x 2, wave tile 128 x 64
    const int wm = wid >> 1, wn = wid & 1;
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fq = lane >> 4;
    const int sw = (frow >> 1) & 7;
    const int a_frag = (wm * 128 + frow) * 128;             // + i*2048
    const int b_frag = A_BYTES + (wn * 64 + frow) * 128;    // + j*2048
    const int c0 = ((0 + fq) ^ sw) << 4, c1 = ((4 + fq) ^ sw) << 4;

    __builtin_amdgcn_s_barrier();   // tile 0 visible (pairs with the loaders' first barrier)
    for (int kt = 0; kt < nk; ++kt) {
        const uint8_t* st = smem + (kt % NST) * ST_BYTES;
        bf16x8 a0[8], b0[4], a1[8], b1[4];
#pragma unroll
        for (int i = 0; i < 8; ++i) a0[i] = *reinterpret_cast<const bf16x8*>(st + a_frag + i * 2048 + c0);
#pragma unroll
        for (int j = 0; j < 4; ++j) b0[j] = *reinterpret_cast<const bf16x8*>(st + b_frag + j * 2048 + c0);
#pragma unroll
        for (int i = 0; i < 8; ++i) a1[i] = *reinterpret_cast<const bf16x8*>(st + a_frag + i * 2048 + c1);
#pragma unroll
        for (int j = 0; j < 4; ++j) b1[j] = *reinterpret_cast<const bf16x8*>(st + b_frag + j * 2048 + c1);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[i], b0[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[i], b1[j], acc[i][j], 0, 0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }

    // ---- epilogue.  acc[i][j][r]: row = m0 + wm*128 + i*16 + fq*4 + r, col = n0 + wn*64 + j*16 + frow
    const int row_base = m0 + wm * 128 + fq * 4;
    const int col_base = n0 + wn * 64 + frow;
    int epi = p.epi;
    if (epi == EPI_QKV_MLP) epi = (n0 >= p.n_split) ? EPI_SWIGLU : EPI_BF16;

    if (epi == EPI_F32) {
        float* C = reinterpret_cast<float*>(p.C);
#pragma unroll
        for (int i = 0; i < 8; ++i)
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
        for (int i = 0; i < 8; ++i)
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
        float* C = reinterpret_cast<float*>(p.C);
        float g[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = col_base + j * 16;
            g[j] = (col < p.N) ? p.vec[col] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
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
        uint16_t* C = reinterpret_cast<uint16_t*>(p.C2);
        const int start = (p.epi == EPI_QKV_MLP) ? p.n_split : 0;
        const int hbase = ((n0 + wn * 64 - start) >> 1) + frow;
        const int hidden = (p.N - start) >> 1;
#pragma unroll
        for (int i = 0; i < 8; ++i)
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
    } else {  // EPI_BIAS_BF16
        uint16_t* C = reinterpret_cast<uint16_t*>(p.C);
        float bsv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = col_base + j * 16;
            bsv[j] = (p.vec != nullptr && col < p.N) ? p.vec[col] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
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

}  // namespace

int launch_gemm_ws(hipStream_t stream, const GemmProblem* probs, int nprob) {
    GemmBatch b;
    b.nprob = nprob;
    int tiles[2] = {0, 0};
    for (int i = 0; i < nprob; ++i) {
        b.p[i] = probs[i];
        tiles[i] = ((probs[i].M + WM - 1) / WM) * ((probs[i].N + WN - 1) / WN);
    }
    if (nprob == 1) b.p[1] = probs[0];
    b.tile_start1 = tiles[0];
    hipLaunchKernelGGL(gemm_bf16_ws_kernel, dim3(tiles[0] + tiles[1]), dim3(512), 0, stream, b);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
