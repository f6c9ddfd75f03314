// gemm256r.hip — 256x256 tile, four waves (one per SIMD, wave tile 128x128, accumulators in
// AGPRs) like gemm256w4.hip, but the operands travel global -> VGPR -> LDS (register staging)
// instead of by LDS-DMA.
//
// Why: with one wave per SIMD every instruction of the wave sits in the MFMA issue stream.
// An LDS-DMA piece costs 60-150 issue cycles (MI355X guide, cycle constants) and a sub-step needs
// 8 of them per 64 MFMAs: ablating the DMA took the 4-wave kernel from 1.06 to 1.40 PF at 8192^3.
// global_load_dwordx4 (SGPR base + VGPR offset) and ds_write_b128 are ordinary cheap issues, one
// per MFMA gap.
//
// Pipeline per sub-step s (32 k), everything inline asm in program order:
//   TOP:   lgkmcnt(0) [frags(s) and the LDS writes of data(s+1) retired] ; vmcnt(8) [data(s+2) in
//          registers, data(s+3) still in flight] ; s_barrier
//   body:  64 MFMA on frags(s), with in the gaps: 16 ds_read (frags(s+1) <- half-stage (s+1)&3),
//          8 ds_write (data(s+2), staging set s&1 -> half-stage (s+2)&3), then 8 global loads
//          (data(s+4) -> staging set s&1).
#include "../dev.h"
#include "../kernels.h"
#include <stdlib.h>

namespace {

constexpr int TM = 256, TN = 256, KS = 32;
constexpr int PART_BYTES = 256 * KS * 2;       // 16 KB: A part or W part of a half-stage
constexpr int HS_BYTES = 2 * PART_BYTES;       // 32 KB
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void tile_coords_r4(int local, int ntiles, int tiles_m, int tiles_n, int& tm, int& tn) {
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
__global__ __launch_bounds__(256, 1) void gemm_bf16_256x256_r4_kernel(GemmBatch batch) {
    __shared__ __attribute__((aligned(16))) uint8_t smem[4 * HS_BYTES];  // 128 KB, the only LDS object

    const int bid = blockIdx.x;
    const int pi = (batch.nprob > 1 && bid >= batch.tile_start1) ? 1 : 0;
    const GemmProblem& p = batch.p[pi];
    const int local = bid - (pi ? batch.tile_start1 : 0);
    const int tiles_m = (p.M + TM - 1) / TM, tiles_n = (p.N + TN - 1) / TN;
    int tm, tn;
    tile_coords_r4(local, tiles_m * tiles_n, tiles_m, tiles_n, tm, tn);
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
    const int ns = p.K / KS;

    // staging registers: two sets of 8 x 16 B per thread (set u&1 carries data(u))
    u32x4 stA0[4], stW0[4], stA1[4], stW1[4];
    const uint32_t wr_addr = lds_base + wid * 4096 + lane * 16;   // + j*1024 (+ PART_BYTES) + half-stage
#define R4_GLOAD(dst_, voff_, sb_) \
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst_) : "v"(voff_), "s"(sb_) : "memory");
#define R4_DSW(addr_, src_, off_) \
    asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr_), "v"(src_), "i"(off_) : "memory");
#define R4_LOAD_ALL(stA_, stW_, s_)                                                        \
    {                                                                                      \
        const int s__ = (s_);                                                              \
        if (s__ < ns) {                                                                    \
            const uint8_t* a__ = gA + (size_t)s__ * (KS * 2);                              \
            const uint8_t* w__ = gW + (size_t)s__ * (KS * 2);                              \
            R4_GLOAD(stA_[0], voffA[0], a__) R4_GLOAD(stW_[0], voffW[0], w__)              \
            R4_GLOAD(stA_[1], voffA[1], a__) R4_GLOAD(stW_[1], voffW[1], w__)              \
            R4_GLOAD(stA_[2], voffA[2], a__) R4_GLOAD(stW_[2], voffW[2], w__)              \
            R4_GLOAD(stA_[3], voffA[3], a__) R4_GLOAD(stW_[3], voffW[3], w__)              \
        }                                                                                  \
    }
#define R4_WRITE_ALL(stA_, stW_, s_)                                                       \
    {                                                                                      \
        const int s__ = (s_);                                                              \
        if (s__ < ns) {                                                                    \
            const uint32_t w__ = wr_addr + (uint32_t)((s__ & 3) * HS_BYTES);               \
            R4_DSW(w__, stA_[0], 0) R4_DSW(w__, stW_[0], PART_BYTES)                       \
            R4_DSW(w__, stA_[1], 1024) R4_DSW(w__, stW_[1], PART_BYTES + 1024)             \
            R4_DSW(w__, stA_[2], 2048) R4_DSW(w__, stW_[2], PART_BYTES + 2048)             \
            R4_DSW(w__, stA_[3], 3072) R4_DSW(w__, stW_[3], PART_BYTES + 3072)             \
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
    // group g = the 8 MFMAs of accumulator row g with at most ONE other instruction per MFMA gap:
    // 2 fragment prefetches, 1 LDS write of data(s+2), 1 global load of data(s+4) (same staging register).
#define W4_GROUP_W(RD, WR, fc_a, fc_b, fn_a, fn_b, g_, stx_, woffx_, sty_, woffy_)   \
    W4_M(fc_a, fc_b, g_, 0)                                                \
    if (RD) W4_DSR(fn_a[g_], ha, g_ * 1024)                                \
    W4_M(fc_a, fc_b, g_, 1)                                                \
    if (RD) W4_DSR(fn_b[g_], hb, g_ * 1024)                                \
    W4_M(fc_a, fc_b, g_, 2)                                                \
    if (WR) R4_DSW(wa, stx_, woffx_)                                       \
    W4_M(fc_a, fc_b, g_, 3)                                                \
    W4_M(fc_a, fc_b, g_, 4)                                                \
    if (WR) R4_DSW(wa, sty_, woffy_)                                       \
    W4_M(fc_a, fc_b, g_, 5)                                                \
    W4_M(fc_a, fc_b, g_, 6)                                                \
    W4_M(fc_a, fc_b, g_, 7)
#define W4_GROUP_L(RD, LD, fc_a, fc_b, fn_a, fn_b, g_, stx_, voffx_, sbx_, sty_, voffy_, sby_)   \
    W4_M(fc_a, fc_b, g_, 0)                                                \
    if (RD) W4_DSR(fn_a[g_], ha, g_ * 1024)                                \
    W4_M(fc_a, fc_b, g_, 1)                                                \
    if (RD) W4_DSR(fn_b[g_], hb, g_ * 1024)                                \
    W4_M(fc_a, fc_b, g_, 2)                                                \
    if (LD) R4_GLOAD(stx_, voffx_, sbx_)                                   \
    W4_M(fc_a, fc_b, g_, 3)                                                \
    W4_M(fc_a, fc_b, g_, 4)                                                \
    if (LD) R4_GLOAD(sty_, voffy_, sby_)                                   \
    W4_M(fc_a, fc_b, g_, 5)                                                \
    W4_M(fc_a, fc_b, g_, 6)                                                \
    W4_M(fc_a, fc_b, g_, 7)
    // the 8 LDS writes go in the first half of the sub-step (retired long before the next TOP's
    // lgkmcnt(0)), the 8 global loads that refill the same staging registers in the second half
#define W4_BODY(RD, WR, LD, fc_a, fc_b, fn_a, fn_b, stA_, stW_, s_)                           \
    {                                                                                         \
        const int sb__ = (s_);                                                                \
        const uint32_t ha = a_frag + (uint32_t)(((sb__ + 1) & 3) * HS_BYTES);                 \
        const uint32_t hb = b_frag + (uint32_t)(((sb__ + 1) & 3) * HS_BYTES);                 \
        const uint32_t wa = wr_addr + (uint32_t)(((sb__ + 2) & 3) * HS_BYTES);                \
        const uint8_t* sa = gA + (size_t)(sb__ + 4) * (KS * 2);                               \
        const uint8_t* sw = gW + (size_t)(sb__ + 4) * (KS * 2);                               \
        W4_GROUP_W(RD, WR, fc_a, fc_b, fn_a, fn_b, 0, stA_[0], 0, stW_[0], PART_BYTES)                  \
        W4_GROUP_W(RD, WR, fc_a, fc_b, fn_a, fn_b, 1, stA_[1], 1024, stW_[1], PART_BYTES + 1024)        \
        W4_GROUP_W(RD, WR, fc_a, fc_b, fn_a, fn_b, 2, stA_[2], 2048, stW_[2], PART_BYTES + 2048)        \
        W4_GROUP_W(RD, WR, fc_a, fc_b, fn_a, fn_b, 3, stA_[3], 3072, stW_[3], PART_BYTES + 3072)        \
        W4_GROUP_L(RD, LD, fc_a, fc_b, fn_a, fn_b, 4, stA_[0], voffA[0], sa, stW_[0], voffW[0], sw)     \
        W4_GROUP_L(RD, LD, fc_a, fc_b, fn_a, fn_b, 5, stA_[1], voffA[1], sa, stW_[1], voffW[1], sw)     \
        W4_GROUP_L(RD, LD, fc_a, fc_b, fn_a, fn_b, 6, stA_[2], voffA[2], sa, stW_[2], voffW[2], sw)     \
        W4_GROUP_L(RD, LD, fc_a, fc_b, fn_a, fn_b, 7, stA_[3], voffA[3], sa, stW_[3], voffW[3], sw)     \
    }
#define W4_TOP(N_)                                                                 \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(" #N_ ")" ::: "memory");  \
    __builtin_amdgcn_s_barrier();                                                  \
    asm volatile("" ::: "memory");

    // ---- prologue: data(0), data(1) written to LDS; data(2), data(3) in flight; frags(0) in registers
    R4_LOAD_ALL(stA0, stW0, 0)
    R4_LOAD_ALL(stA1, stW1, 1)
    if (ns > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    R4_WRITE_ALL(stA0, stW0, 0)
    R4_LOAD_ALL(stA0, stW0, 2)
    if (ns > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    R4_WRITE_ALL(stA1, stW1, 1)
    R4_LOAD_ALL(stA1, stW1, 3)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
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
    for (; s + 5 < ns; s += 2) {   // steady state: vmcnt(8) keeps data(s+3) in flight
        W4_TOP(8)
        W4_BODY(!(ABL & 1), !(ABL & 2), !(ABL & 2), fa0, fb0, fa1, fb1, stA0, stW0, s)
        W4_TOP(8)
        W4_BODY(!(ABL & 1), !(ABL & 2), !(ABL & 2), fa1, fb1, fa0, fb0, stA1, stW1, s + 1)
    }
    for (; s < ns; s += 2) {       // last sub-steps: nothing younger in flight -> vmcnt(0)
        W4_TOP(0)
        { const bool rd = s + 1 < ns, wr = s + 2 < ns, ld = s + 4 < ns; W4_BODY(rd, wr, ld, fa0, fb0, fa1, fb1, stA0, stW0, s) }
        W4_TOP(0)
        { const bool rd = s + 2 < ns, wr = s + 3 < ns, ld = s + 5 < ns; W4_BODY(rd, wr, ld, fa1, fb1, fa0, fb0, stA1, stW1, s + 1) }
    }
    // MFMA result -> first non-MFMA reader: hipcc pads nothing behind inline asm (guide §5.7 item 2)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
#undef R4_GLOAD
#undef R4_DSW
#undef R4_LOAD_ALL
#undef R4_WRITE_ALL
#undef W4_DSR
#undef W4_M
#undef W4_GROUP_W
#undef W4_GROUP_L
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

int launch_gemm256_r4(hipStream_t stream, const GemmProblem* probs, int nprob) {
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
    if (abl == 1) hipLaunchKernelGGL(gemm_bf16_256x256_r4_kernel<1>, grid, block, 0, stream, b);
    else if (abl == 2) hipLaunchKernelGGL(gemm_bf16_256x256_r4_kernel<2>, grid, block, 0, stream, b);
    else if (abl == 3) hipLaunchKernelGGL(gemm_bf16_256x256_r4_kernel<3>, grid, block, 0, stream, b);
    else hipLaunchKernelGGL(gemm_bf16_256x256_r4_kernel<0>, grid, block, 0, stream, b);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
