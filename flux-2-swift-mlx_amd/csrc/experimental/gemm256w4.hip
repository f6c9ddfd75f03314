// gemm256w4.hip — 256x256 tile, FOUR waves (one per SIMD), wave tile 128x128 with the
// 256 f32 accumulators in the unified VGPR/AGPR file (512 registers per lane at one wave
// per SIMD).  Same contract / epilogues as gemm.hip.
//
// Why: the 8-wave kernel (gemm256.hip) reads every A row-panel from LDS four times and
// every B panel twice (192 KB of ds_read per 64-deep K-tile) and its LDS array is ~85 %
// busy between LDS-DMA writes and fragment reads.  With 128x128 per wave the same K-tile
// costs 128 KB of ds_read, half the waves take part in each barrier, and the LDS reads of
// the next k-step are issued under the 64 MFMAs of the current one (software pipelining
// inside the wave instead of across two waves of a SIMD).
//
// Pipeline (sub-step = 32 k):  LDS holds 4 half-stages of [256 A rows + 256 W rows] x 32 k
// (32 KB each, 64-B rows, 16-B chunk ^ (-(row>>2))&3 so ds_read_b128 is conflict-free).
//   body(s):  lgkmcnt(0)            frags(s) are in registers
//             vmcnt(16) ; barrier   data(s+1) landed for every wave; half-stage s&3 is free
//             LDS-DMA data(s+4) -> half-stage s&3          (8 x global_load_lds_dwordx4 / thread)
//             ds_read frags(s+1) <- half-stage (s+1)&3     (16 x ds_read_b128)
//             64 x MFMA 16x16x32 on frags(s)
// DMA data has three sub-steps (~3 x 1024 MFMA cycles) to land; vmcnt is never 0 in the loop.
#include "../dev.h"
#include "../kernels.h"
#include <stdlib.h>

namespace {

constexpr int TM = 256, TN = 256, KS = 32;
constexpr int PART_BYTES = 256 * KS * 2;       // 16 KB: A part or W part of a half-stage
constexpr int HS_BYTES = 2 * PART_BYTES;       // 32 KB
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void tile_coords_w4(int local, int ntiles, int tiles_m, int tiles_n, int& tm, int& tn) {
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

template <int ABL>  // timing-only ablations: 1 = no fragment reads in the loop, 2 = no DMA in the loop
__global__ __launch_bounds__(256, 1) void gemm_bf16_256x256_w4_kernel(GemmBatch batch) {
    __shared__ __attribute__((aligned(16))) uint8_t smem[4 * HS_BYTES];  // 128 KB, the only LDS object

    const int bid = blockIdx.x;
    const int pi = (batch.nprob > 1 && bid >= batch.tile_start1) ? 1 : 0;
    const GemmProblem& p = batch.p[pi];
    const int local = bid - (pi ? batch.tile_start1 : 0);
    const int tiles_m = (p.M + TM - 1) / TM, tiles_n = (p.N + TN - 1) / TN;
    int tm, tn;
    tile_coords_w4(local, tiles_m * tiles_n, tiles_m, tiles_n, tm, tn);
    const int m0 = tm * TM, n0 = tn * TN;

    const int t = threadIdx.x;
    const int lane = t & 63, wid = t >> 6;
    const int wm = wid >> 1, wn = wid & 1;

    // ---- LDS-DMA sources.  Wave-instruction q (0..15) of a part covers rows 16q..16q+15
    // (1 KB); wave w issues q = 4w + j.  lane -> row 16q + (lane>>2), LDS chunk lane&3 holds
    // global chunk (lane&3) ^ f(row), f(row) = (-(row>>2)) & 3.  Addressing is
    // SGPR base (advances along K) + per-lane 32-bit byte offset (constant), so a DMA issue
    // costs no VALU work.
    uint32_t voffA[4], voffW[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = (wid * 4 + j) * 16 + (lane >> 2);
        const int chunk = (lane & 3) ^ ((-(row >> 2)) & 3);
        voffA[j] = (uint32_t)(((size_t)min(m0 + row, p.M - 1) * p.lda + chunk * 8) * 2);
        voffW[j] = (uint32_t)(((size_t)min(n0 + row, p.N - 1) * p.ldw + chunk * 8) * 2);
    }
    const uint8_t* gA = reinterpret_cast<const uint8_t*>(p.A);
    const uint8_t* gW = reinterpret_cast<const uint8_t*>(p.W);
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem;
    const uint32_t m0_wave = __builtin_amdgcn_readfirstlane(lds_base + wid * 4096);
    const int ns = p.K / KS;

    // one LDS-DMA piece (1 KB per wave): M0 = LDS destination, written in the same statement
#define W4_DMA(voff_, sbase_, m0_)                                                                         \
    {                                                                                                      \
        unsigned keep__;                                                                                   \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0" \
                     : "=&s"(keep__) : "v"(voff_), "s"(m0_), "s"(sbase_) : "memory");                      \
    }
#define W4_ISSUE_ALL(s_)                                                                   \
    {                                                                                      \
        const int s__ = (s_);                                                              \
        if (s__ < ns) {                                                                    \
            const uint8_t* a__ = gA + (size_t)s__ * (KS * 2);                              \
            const uint8_t* w__ = gW + (size_t)s__ * (KS * 2);                              \
            const uint32_t m__ = m0_wave + (uint32_t)((s__ & 3) * HS_BYTES);               \
            W4_DMA(voffA[0], a__, m__) W4_DMA(voffA[1], a__, m__ + 1024)                   \
            W4_DMA(voffA[2], a__, m__ + 2048) W4_DMA(voffA[3], a__, m__ + 3072)            \
            W4_DMA(voffW[0], w__, m__ + PART_BYTES) W4_DMA(voffW[1], w__, m__ + PART_BYTES + 1024)          \
            W4_DMA(voffW[2], w__, m__ + PART_BYTES + 2048) W4_DMA(voffW[3], w__, m__ + PART_BYTES + 3072)   \
        }                                                                                  \
    }

    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fq = lane >> 4;
    const int fslot = (fq ^ ((-(frow >> 2)) & 3)) << 4;
    const uint32_t a_frag = lds_base + (wm * 128 + frow) * 64 + fslot;                 // + i*1024
    const uint32_t b_frag = lds_base + PART_BYTES + (wn * 128 + frow) * 64 + fslot;    // + j*1024

    bf16x8 fa0[8], fb0[8], fa1[8], fb1[8];

    // Everything inside the K loop is inline asm, in the order written here:
    //   * MFMA with the accumulator pinned to AGPRs ("+a") and fragments in VGPRs ("v"): left to
    //     itself hipcc keeps the 256 accumulators in VGPRs and copies every freshly read fragment
    //     into AGPRs behind an lgkmcnt wait, which serialises the LDS reads in front of the MFMAs;
    //   * ds_read_b128 / LDS-DMA invisible to hipcc's scoreboard (it would wait lgkmcnt(0) in front
    //     of the first inline-asm MFMA); the hand-placed waits of W4_TOP retire them before their
    //     first consumer (guide §5.7 item 1 form (iii));
    //   * one wave per SIMD issues in order, so the 16 reads and 8 DMA pieces of a sub-step are
    //     spread between the 64 MFMAs (2 reads + 1 piece per 8 MFMAs) instead of in front of them.
#define W4_DSR(dst_, addr_, off_) \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst_) : "v"(addr_), "i"(off_) : "memory");
#define W4_M(fa_, fb_, i_, j_) \
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i_][j_]) : "v"(fa_[i_]), "v"(fb_[j_]));
    // M0 (LDS-DMA destination) is written one MFMA ahead of the DMA that reads it (the MFMA is
    // the required wait state); nothing else in this kernel uses M0.
#define W4_SETM0(mm_, off_) asm volatile("s_add_u32 m0, %0, %1" ::"s"(mm_), "i"(off_) : "memory", "scc");
#define W4_GLDS(voff_, sb_) asm volatile("global_load_lds_dwordx4 %0, %1" ::"v"(voff_), "s"(sb_) : "memory");
    // group g = the 8 MFMAs of accumulator row g with at most ONE other instruction per MFMA gap:
    // prefetch of fragments g of the next sub-step (2 ds_read_b128) and DMA piece g (s_add m0 + load).
#define W4_GROUP(RD, IS, fc_a, fc_b, fn_a, fn_b, g_, voff_, sb_, m0off_)   \
    W4_M(fc_a, fc_b, g_, 0)                                                \
    if (RD) W4_DSR(fn_a[g_], ha, g_ * 1024)                                \
    W4_M(fc_a, fc_b, g_, 1)                                                \
    if (RD) W4_DSR(fn_b[g_], hb, g_ * 1024)                                \
    W4_M(fc_a, fc_b, g_, 2)                                                \
    if (IS) W4_SETM0(mm, m0off_)                                           \
    W4_M(fc_a, fc_b, g_, 3)                                                \
    if (IS) W4_GLDS(voff_, sb_)                                            \
    W4_M(fc_a, fc_b, g_, 4)                                                \
    W4_M(fc_a, fc_b, g_, 5)                                                \
    W4_M(fc_a, fc_b, g_, 6)                                                \
    W4_M(fc_a, fc_b, g_, 7)
#define W4_BODY(RD, IS, fc_a, fc_b, fn_a, fn_b, s_)                                           \
    {                                                                                         \
        const int sb__ = (s_);                                                                \
        const uint32_t ha = a_frag + (uint32_t)(((sb__ + 1) & 3) * HS_BYTES);                 \
        const uint32_t hb = b_frag + (uint32_t)(((sb__ + 1) & 3) * HS_BYTES);                 \
        const uint8_t* sa = gA + (size_t)(sb__ + 4) * (KS * 2);                               \
        const uint8_t* sw = gW + (size_t)(sb__ + 4) * (KS * 2);                               \
        const uint32_t mm = m0_wave + (uint32_t)((sb__ & 3) * HS_BYTES);                      \
        W4_GROUP(RD, IS, fc_a, fc_b, fn_a, fn_b, 0, voffA[0], sa, 0)                          \
        W4_GROUP(RD, IS, fc_a, fc_b, fn_a, fn_b, 1, voffW[0], sw, PART_BYTES)                 \
        W4_GROUP(RD, IS, fc_a, fc_b, fn_a, fn_b, 2, voffA[1], sa, 1024)                       \
        W4_GROUP(RD, IS, fc_a, fc_b, fn_a, fn_b, 3, voffW[1], sw, PART_BYTES + 1024)          \
        W4_GROUP(RD, IS, fc_a, fc_b, fn_a, fn_b, 4, voffA[2], sa, 2048)                       \
        W4_GROUP(RD, IS, fc_a, fc_b, fn_a, fn_b, 5, voffW[2], sw, PART_BYTES + 2048)          \
        W4_GROUP(RD, IS, fc_a, fc_b, fn_a, fn_b, 6, voffA[3], sa, 3072)                       \
        W4_GROUP(RD, IS, fc_a, fc_b, fn_a, fn_b, 7, voffW[3], sw, PART_BYTES + 3072)          \
    }
#define W4_TOP(N_)                                                                 \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(" #N_ ")" ::: "memory");  \
    __builtin_amdgcn_s_barrier();                                                  \
    asm volatile("" ::: "memory");

    // ---- prologue: data(0..3) in flight, frags(0) in registers
    W4_ISSUE_ALL(0) W4_ISSUE_ALL(1) W4_ISSUE_ALL(2) W4_ISSUE_ALL(3)
    if (ns > 3) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    {
        const uint32_t ha = a_frag, hb = b_frag;
        W4_DSR(fa0[0], ha, 0) W4_DSR(fa0[1], ha, 1024) W4_DSR(fa0[2], ha, 2048) W4_DSR(fa0[3], ha, 3072)
        W4_DSR(fa0[4], ha, 4096) W4_DSR(fa0[5], ha, 5120) W4_DSR(fa0[6], ha, 6144) W4_DSR(fa0[7], ha, 7168)
        W4_DSR(fb0[0], hb, 0) W4_DSR(fb0[1], hb, 1024) W4_DSR(fb0[2], hb, 2048) W4_DSR(fb0[3], hb, 3072)
        W4_DSR(fb0[4], hb, 4096) W4_DSR(fb0[5], hb, 5120) W4_DSR(fb0[6], hb, 6144) W4_DSR(fb0[7], hb, 7168)
    }

    int s = 0;
    for (; s + 5 < ns; s += 2) {   // steady state: every prefetch and DMA exists, vmcnt(16) never drains
        W4_TOP(16)
        W4_BODY(!(ABL & 1), !(ABL & 2), fa0, fb0, fa1, fb1, s)
        W4_TOP(16)
        W4_BODY(!(ABL & 1), !(ABL & 2), fa1, fb1, fa0, fb0, s + 1)
    }
    for (; s < ns; s += 2) {       // last sub-steps: nothing younger in flight -> vmcnt(0)
        W4_TOP(0)
        { const bool rd = s + 1 < ns, is = s + 4 < ns; W4_BODY(rd, is, fa0, fb0, fa1, fb1, s) }
        W4_TOP(0)
        { const bool rd = s + 2 < ns, is = s + 5 < ns; W4_BODY(rd, is, fa1, fb1, fa0, fb0, s + 1) }
    }
    // MFMA result -> first non-MFMA reader: hipcc pads nothing behind inline asm (guide §5.7 item 2)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
#undef W4_DMA
#undef W4_ISSUE_ALL
#undef W4_DSR
#undef W4_M
#undef W4_SETM0
#undef W4_GLDS
#undef W4_GROUP
#undef W4_BODY
#undef W4_TOP

    // ---- epilogue.  acc[i][j][r]: row = m0 + wm*128 + i*16 + fq*4 + r, col = n0 + wn*128 + j*16 + frow
    const int row_base = m0 + wm * 128 + fq * 4;
    const int col_base = n0 + wn * 128 + frow;
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
                    for (int j = 0; j < 8; ++j) {
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
                    for (int j = 0; j < 8; ++j) {
                        const int col = col_base + j * 16;
                        if (col < p.N) C[(size_t)row * p.ldc + col] = f32_to_bf16_bits(acc[i][j][r] * p.alpha);
                    }
                }
            }
    } else if (epi == EPI_GATE_RESID) {
        float* C = reinterpret_cast<float*>(p.C);
        float g[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
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
                    for (int j = 0; j < 8; ++j) {
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
        const int hidden = (p.N - start) >> 1;
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2) {
            const int hbase = ((n0 + wn * 128 + g2 * 64 - start) >> 1) + frow;
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
                                const float gate = acc[i][g2 * 4 + j][r], val = acc[i][g2 * 4 + j + 2][r];
                                C[(size_t)row * p.ldc2 + h] = f32_to_bf16_bits(silu_f(gate) * val);
                            }
                        }
                    }
                }
        }
    } else {  // EPI_BIAS_BF16
        uint16_t* C = reinterpret_cast<uint16_t*>(p.C);
        float bsv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
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
                    for (int j = 0; j < 8; ++j) {
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

int launch_gemm256_w4(hipStream_t stream, const GemmProblem* probs, int nprob) {
    GemmBatch b;
    b.nprob = nprob;
    int tiles[2] = {0, 0};
    for (int i = 0; i < nprob; ++i) {
        b.p[i] = probs[i];
        tiles[i] = ((probs[i].M + TM - 1) / TM) * ((probs[i].N + TN - 1) / TN);
    }
    if (nprob == 1) b.p[1] = probs[0];
    b.tile_start1 = tiles[0];
    static const int abl = [] {
        const char* e = F2_DEV_ENV("FLUX2MI_GEMM_ABLATE");
        return e ? atoi(e) : 0;
    }();
    const dim3 grid(tiles[0] + tiles[1]), block(256);
    if (abl == 1) hipLaunchKernelGGL(gemm_bf16_256x256_w4_kernel<1>, grid, block, 0, stream, b);
    else if (abl == 2) hipLaunchKernelGGL(gemm_bf16_256x256_w4_kernel<2>, grid, block, 0, stream, b);
    else if (abl == 3) hipLaunchKernelGGL(gemm_bf16_256x256_w4_kernel<3>, grid, block, 0, stream, b);
    else hipLaunchKernelGGL(gemm_bf16_256x256_w4_kernel<0>, grid, block, 0, stream, b);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
