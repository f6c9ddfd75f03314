// gemm256g4.hip — 256x256x64 tile, FOUR waves (one per SIMD), wave tile 128x128, the 256 f32 accumulators in AGPRs,
// operands global -> VGPR -> LDS (register staging), ONE barrier per 64-deep K-tile.
//
// Why a second large-tile kernel (measured, profiles/r02_power_*.txt): the 8-wave kernel (gemm256.hip) runs at the
// card's 1400 W cap on 8192^3; timing-only ablations show its K loop is paced by its LDS side — 192 KB of fragment
// reads + 64 KB of LDS-DMA writes per K-tile, eight barriers — not by the MFMAs (removing the MFMAs does not shorten
// it).  hipBLASLt's MT256x256x64 kernel has four waves: every A row panel is read from LDS twice instead of four times
// (128 KB per K-tile), and it finishes the same GEMM in 0.80 of the time at 0.83 of the power.  This kernel takes that
// geometry and keeps everything else of this library (contract, epilogues, tile order).
//
// Geometry / schedule
//   * 256 threads = 4 waves as 2 (M) x 2 (N); per wave 8 x 8 accumulator blocks of 16x16 (256 AGPRs), MFMA 16x16x32
//     with the W fragment as the A operand, so a lane holds 4 consecutive output columns (16-byte epilogue accesses).
//   * LDS: 2 stages x [256 A rows | 256 W rows] x 128 B = 128 KB.  16-B chunk c of row r sits at c ^ ((r >> 1) & 7)
//     (the 8-wave kernel's swizzle: conflict-free ds_read_b128 fragment reads; applied on the ds_write address here).
//   * global loads are whole 128-byte rows (8 rows per wave-instruction); a K-tile is 16 loads per wave into 16 staging
//     registers (64 VGPRs).  Sub-step s (32 k, 64 MFMAs) = half h = s & 1 of K-tile u = s >> 1.
//       body(2u)  : reads frags(2u+1) <- stage u&1     | writes R[8..15] = W rows of K-tile u+1 -> stage (u+1)&1 | loads R[8..15] <- K-tile u+2
//       body(2u+1): reads frags(2u+2) <- stage (u+1)&1 | writes R[0..7]  = A rows of K-tile u+2 -> stage u&1     | loads R[0..7]  <- K-tile u+3
//     Every auxiliary instruction sits in its own MFMA gap, reads first (their latency runs under the rest of the body),
//     then writes, then loads.  TOP(2u): lgkmcnt(0), vmcnt(8).  TOP(2u+1): lgkmcnt(0), vmcnt(8), s_barrier — the only
//     barrier of the K-tile: it publishes K-tile u+1's LDS image (RAW) and, since every wave has retired its reads of
//     stage (u+1)&1's previous tenant (K-tile u-1) before it, frees that stage for the writes behind it (WAR).
//     vmcnt never reaches 0 in the loop: 8 loads (half a K-tile) stay in flight across every wait.
#include "../dev.h"
#include "../kernels.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>

namespace {

constexpr int TM = 256, TN = 256;
constexpr int PART = 256 * 128;        // 32 KB: the A rows or the W rows of a stage
constexpr int STAGE = 2 * PART;        // 64 KB

__device__ __forceinline__ void tile_coords_g4(int local, int ntiles, int tiles_m, int tiles_n, int& tm, int& tn) {
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

__device__ __forceinline__ void g4_swap16(uint32_t& a, uint32_t& b) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void g4_swap32(uint32_t& a, uint32_t& b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}

// DIAG: timing-only build of the loop with s_memtime stamps around the waits and the barrier (never the shipped path)
// ABL (timing only, wrong results): bit 0 no fragment reads, bit 1 no LDS writes, bit 2 no global loads in the steady loop
template <bool DIAG, int ABL = 0, int DMA = 0>   // DMA: 0 register staging, 1 global_load_lds, 2 buffer_load .. lds (MUBUF, SRD + SGPR K offset)
__global__ __launch_bounds__(256, 1) void gemm_bf16_256x256_g4_kernel(GemmBatch batch) {
    __shared__ __attribute__((aligned(16))) uint8_t smem[2 * STAGE];  // 128 KB, the only LDS object

    const int bid = blockIdx.x;
    const int pi = (batch.nprob > 1 && bid >= batch.tile_start1) ? 1 : 0;
    const GemmProblem& p = batch.p[pi];
    const int local = bid - (pi ? batch.tile_start1 : 0);
    const int tiles_m = (p.M + TM - 1) / TM, tiles_n = (p.N + TN - 1) / TN;
    int tm, tn;
    tile_coords_g4(local, tiles_m * tiles_n, tiles_m, tiles_n, tm, tn);
    const int m0 = tm * TM, n0 = tn * TN;

    const int t = threadIdx.x;
    const int lane = t & 63, wid = t >> 6;
    const int wm = wid >> 1, wn = wid & 1;

    // ---- staging: load j (0..7) of a part covers rows 8(4j + wid) .. +8 of that part, lane -> row + (lane >> 3), chunk lane & 7
    // (one whole 128-byte row per 8 lanes).  R[0..7] = the A part, R[8..15] = the W part of a K-tile.
    uint32_t voffA[8], voffW[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int row = (4 * j + wid) * 8 + (lane >> 3);
        voffA[j] = (uint32_t)(((size_t)min(m0 + row, p.M - 1) * p.lda + (lane & 7) * 8) * 2);
        voffW[j] = (uint32_t)(((size_t)min(n0 + row, p.N - 1) * p.ldw + (lane & 7) * 8) * 2);
    }
    if (DMA) {   // LDS-DMA writes lane-linear: the chunk swizzle moves to the SOURCE address (guide rule 21)
        const int wr8 = wid * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((wr8 >> 1) & 7);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = (4 * j + wid) * 8 + (lane >> 3);
            voffA[j] = (uint32_t)(((size_t)min(m0 + row, p.M - 1) * p.lda + chunk * 8) * 2);
            voffW[j] = (uint32_t)(((size_t)min(n0 + row, p.N - 1) * p.ldw + chunk * 8) * 2);
        }
    }
    const uint8_t* gA = reinterpret_cast<const uint8_t*>(p.A);
    const uint8_t* gW = reinterpret_cast<const uint8_t*>(p.W);
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem;
    // LDS write address of this lane inside a part: row = (4j + wid)*8 + (lane>>3) -> + j*4096 as an immediate
    const int wrow = wid * 8 + (lane >> 3);
    const uint32_t wr_addr = lds_base + wrow * 128 + (((lane & 7) ^ ((wrow >> 1) & 7)) << 4);   // row j*32 + wrow has the same (row >> 1) & 7
    const int nk = p.K / 64;

    u32x4 R[16];
#define G4_GLOAD(dst_, voff_, sb_) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst_) : "v"(voff_), "s"(sb_) : "memory");
#define G4_DSW(addr_, src_, off_) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr_), "v"(src_), "i"(off_) : "memory");
#define G4_DSR(dst_, addr_, off_) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst_) : "v"(addr_), "i"(off_) : "memory");
    // W fragment as the A operand: acc[i][j] rows = output columns j*16 + fq*4 + r, lanes = output rows i*16 + frow
#define G4_M(fa_, fb_, i_, j_) \
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i_][j_]) : "v"(fb_[j_]), "v"(fa_[i_]));

    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- fragment reads: lane (frow, fq) reads row (block*16 + frow), chunk (4h + fq) ^ ((frow >> 1) & 7)
    const int frow = lane & 15, fq = lane >> 4;
    const int sw = (frow >> 1) & 7;
    const uint32_t a_frag = lds_base + (wm * 128 + frow) * 128;            // + i*2048 (imm) + chunk + stage
    const uint32_t b_frag = lds_base + PART + (wn * 128 + frow) * 128;     // + j*2048 (imm) + chunk + stage
    const uint32_t ch0 = ((0 + fq) ^ sw) << 4, ch1 = ((4 + fq) ^ sw) << 4;

    bf16x8 fa0[8], fb0[8], fa1[8], fb1[8];

    // one group = the 8 MFMAs of accumulator row g; aux instruction X(k) goes behind MFMA k of the group
#define G4_GROUP(fc_a, fc_b, g_, X0, X1, X2, X3, X4, X5, X6, X7) \
    G4_M(fc_a, fc_b, g_, 0) X0 G4_M(fc_a, fc_b, g_, 1) X1 G4_M(fc_a, fc_b, g_, 2) X2 G4_M(fc_a, fc_b, g_, 3) X3 \
    G4_M(fc_a, fc_b, g_, 4) X4 G4_M(fc_a, fc_b, g_, 5) X5 G4_M(fc_a, fc_b, g_, 6) X6 G4_M(fc_a, fc_b, g_, 7) X7
#define G4_NONE
    // body: MFMAs on (fc_a, fc_b); reads the next fragments (fn_a, fn_b) from ha / hb; RD / WR / LD enable the aux work.
    // WR: R[r0_ .. r0_+7] -> LDS at wa (+ part offset poff_); LD: the same registers <- voff_[0..7] + sbase.
#define G4_BODY(RD, WR, LD, fc_a, fc_b, fn_a, fn_b, r0_, poff_, voff_, sbase_)                                                    \
    {                                                                                                                             \
        G4_GROUP(fc_a, fc_b, 0, if (RD) G4_DSR(fn_a[0], ha, 0), if (RD) G4_DSR(fn_b[0], hb, 0), if (RD) G4_DSR(fn_a[1], ha, 2048), \
                 if (RD) G4_DSR(fn_b[1], hb, 2048), if (RD) G4_DSR(fn_a[2], ha, 4096), if (RD) G4_DSR(fn_b[2], hb, 4096),          \
                 if (RD) G4_DSR(fn_a[3], ha, 6144), if (RD) G4_DSR(fn_b[3], hb, 6144))                                              \
        G4_GROUP(fc_a, fc_b, 1, if (RD) G4_DSR(fn_a[4], ha, 8192), if (RD) G4_DSR(fn_b[4], hb, 8192), if (RD) G4_DSR(fn_a[5], ha, 10240), \
                 if (RD) G4_DSR(fn_b[5], hb, 10240), if (RD) G4_DSR(fn_a[6], ha, 12288), if (RD) G4_DSR(fn_b[6], hb, 12288),        \
                 if (RD) G4_DSR(fn_a[7], ha, 14336), if (RD) G4_DSR(fn_b[7], hb, 14336))                                            \
        G4_GROUP(fc_a, fc_b, 2, G4_NONE, if (WR) G4_DSW(wa, R[r0_ + 0], poff_ + 0 * 4096), G4_NONE,                                \
                 if (WR) G4_DSW(wa, R[r0_ + 1], poff_ + 1 * 4096), G4_NONE, if (WR) G4_DSW(wa, R[r0_ + 2], poff_ + 2 * 4096), G4_NONE, \
                 if (WR) G4_DSW(wa, R[r0_ + 3], poff_ + 3 * 4096))                                                                  \
        G4_GROUP(fc_a, fc_b, 3, G4_NONE, if (WR) G4_DSW(wa, R[r0_ + 4], poff_ + 4 * 4096), G4_NONE,                                \
                 if (WR) G4_DSW(wa, R[r0_ + 5], poff_ + 5 * 4096), G4_NONE, if (WR) G4_DSW(wa, R[r0_ + 6], poff_ + 6 * 4096), G4_NONE, \
                 if (WR) G4_DSW(wa, R[r0_ + 7], poff_ + 7 * 4096))                                                                  \
        G4_GROUP(fc_a, fc_b, 4, G4_NONE, if (LD) G4_GLOAD(R[r0_ + 0], voff_[0], sbase_), G4_NONE,                                  \
                 if (LD) G4_GLOAD(R[r0_ + 1], voff_[1], sbase_), G4_NONE, if (LD) G4_GLOAD(R[r0_ + 2], voff_[2], sbase_), G4_NONE, \
                 if (LD) G4_GLOAD(R[r0_ + 3], voff_[3], sbase_))                                                                   \
        G4_GROUP(fc_a, fc_b, 5, G4_NONE, if (LD) G4_GLOAD(R[r0_ + 4], voff_[4], sbase_), G4_NONE,                                  \
                 if (LD) G4_GLOAD(R[r0_ + 5], voff_[5], sbase_), G4_NONE, if (LD) G4_GLOAD(R[r0_ + 6], voff_[6], sbase_), G4_NONE, \
                 if (LD) G4_GLOAD(R[r0_ + 7], voff_[7], sbase_))                                                                   \
        G4_GROUP(fc_a, fc_b, 6, G4_NONE, G4_NONE, G4_NONE, G4_NONE, G4_NONE, G4_NONE, G4_NONE, G4_NONE)                            \
        G4_GROUP(fc_a, fc_b, 7, G4_NONE, G4_NONE, G4_NONE, G4_NONE, G4_NONE, G4_NONE, G4_NONE, G4_NONE)                            \
    }
    unsigned long long dg_t = 0, dg_lds = 0, dg_vm = 0, dg_bar = 0;
#define G4_STAMP(acc_)                                                                  \
    if (DIAG) {                                                                         \
        unsigned long long n__;                                                         \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(n__)::"memory");    \
        acc_ += n__ - dg_t;                                                             \
        dg_t = n__;                                                                     \
    }
#define G4_WAIT(VM_)                                                   \
    if (DIAG) {                                                        \
        unsigned long long z__ = 0;                                    \
        G4_STAMP(z__) (void)z__;                                            \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             \
        G4_STAMP(dg_lds)                                               \
        asm volatile("s_waitcnt vmcnt(" #VM_ ")" ::: "memory");        \
        G4_STAMP(dg_vm)                                                \
    } else                                                             \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(" #VM_ ")" ::: "memory");
#define G4_BARRIER() __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); G4_STAMP(dg_bar)

    const uint32_t m0_wave = __builtin_amdgcn_readfirstlane(lds_base + wid * 1024);   // + j*4096 + part + stage
#define G4_SETM0(mm_, off_) asm volatile("s_add_u32 m0, %0, %1" ::"s"(mm_), "i"(off_) : "memory", "scc");
    typedef int g4_i32x4 __attribute__((ext_vector_type(4)));
    auto mk_srd = [](const uint8_t* base) {   // raw buffer, stride 0, num_records = 4 GiB - 1, gfx9 word3
        const unsigned long long a = (unsigned long long)base;
        g4_i32x4 r;
        r[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
        r[1] = __builtin_amdgcn_readfirstlane((int)(uint32_t)((a >> 32) & 0xffff));
        r[2] = -1;
        r[3] = 0x00020000;
        return r;
    };
    const g4_i32x4 sa_srd = mk_srd(gA), sw_srd = mk_srd(gW);
#define G4_GLDS(voff_, sb_)                                                                                                    \
    { if (DMA == 2) asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff_), "s"(sb_##_srd), "s"(kof) : "memory"); \
      else asm volatile("global_load_lds_dwordx4 %0, %1" ::"v"(voff_), "s"(sb_) : "memory"); }
    // odd body of the DMA form: 16 reads, then the 16 LDS-DMA pieces of K-tile u+2 (M0 written one MFMA ahead of each)
#define G4_DMA2(fc_a, fc_b, g_, IS, va_, oa_, vb_, ob_, sa_, sb_)                                   \
    G4_GROUP(fc_a, fc_b, g_, if (IS) G4_SETM0(mm, oa_), if (IS) G4_GLDS(va_, sa_), G4_NONE, G4_NONE,       \
             if (IS) G4_SETM0(mm, ob_), if (IS) G4_GLDS(vb_, sb_), G4_NONE, G4_NONE)
#define G4_BODY_DMA(RD, IS, fc_a, fc_b, fn_a, fn_b, sa_, sw_)                                                                     \
    {                                                                                                                             \
        G4_GROUP(fc_a, fc_b, 0, if (RD) G4_DSR(fn_a[0], ha, 0), if (RD) G4_DSR(fn_b[0], hb, 0), if (RD) G4_DSR(fn_a[1], ha, 2048), \
                 if (RD) G4_DSR(fn_b[1], hb, 2048), if (RD) G4_DSR(fn_a[2], ha, 4096), if (RD) G4_DSR(fn_b[2], hb, 4096),          \
                 if (RD) G4_DSR(fn_a[3], ha, 6144), if (RD) G4_DSR(fn_b[3], hb, 6144))                                              \
        G4_GROUP(fc_a, fc_b, 1, if (RD) G4_DSR(fn_a[4], ha, 8192), if (RD) G4_DSR(fn_b[4], hb, 8192), if (RD) G4_DSR(fn_a[5], ha, 10240), \
                 if (RD) G4_DSR(fn_b[5], hb, 10240), if (RD) G4_DSR(fn_a[6], ha, 12288), if (RD) G4_DSR(fn_b[6], hb, 12288),        \
                 if (RD) G4_DSR(fn_a[7], ha, 14336), if (RD) G4_DSR(fn_b[7], hb, 14336))                                            \
        G4_DMA2(fc_a, fc_b, 2, IS, voffA[0], 0 * 4096, voffA[1], 1 * 4096, sa_, sa_)                                               \
        G4_DMA2(fc_a, fc_b, 3, IS, voffA[2], 2 * 4096, voffA[3], 3 * 4096, sa_, sa_)                                               \
        G4_DMA2(fc_a, fc_b, 4, IS, voffA[4], 4 * 4096, voffA[5], 5 * 4096, sa_, sa_)                                               \
        G4_DMA2(fc_a, fc_b, 5, IS, voffA[6], 6 * 4096, voffA[7], 7 * 4096, sa_, sa_)                                               \
        G4_DMA2(fc_a, fc_b, 6, IS, voffW[0], PART + 0 * 4096, voffW[1], PART + 1 * 4096, sw_, sw_)                                 \
        G4_DMA2(fc_a, fc_b, 7, IS, voffW[2], PART + 2 * 4096, voffW[3], PART + 3 * 4096, sw_, sw_)                                 \
    }
    // the remaining four W pieces ride in the EVEN body behind its reads (groups 2, 3)
#define G4_BODY_DMA_EVEN(IS, fc_a, fc_b, fn_a, fn_b, sw_)                                                                         \
    {                                                                                                                             \
        G4_GROUP(fc_a, fc_b, 0, G4_DSR(fn_a[0], ha, 0), G4_DSR(fn_b[0], hb, 0), G4_DSR(fn_a[1], ha, 2048),                         \
                 G4_DSR(fn_b[1], hb, 2048), G4_DSR(fn_a[2], ha, 4096), G4_DSR(fn_b[2], hb, 4096),                                  \
                 G4_DSR(fn_a[3], ha, 6144), G4_DSR(fn_b[3], hb, 6144))                                                             \
        G4_GROUP(fc_a, fc_b, 1, G4_DSR(fn_a[4], ha, 8192), G4_DSR(fn_b[4], hb, 8192), G4_DSR(fn_a[5], ha, 10240),                  \
                 G4_DSR(fn_b[5], hb, 10240), G4_DSR(fn_a[6], ha, 12288), G4_DSR(fn_b[6], hb, 12288),                               \
                 G4_DSR(fn_a[7], ha, 14336), G4_DSR(fn_b[7], hb, 14336))                                                           \
        G4_DMA2(fc_a, fc_b, 2, IS, voffW[4], PART + 4 * 4096, voffW[5], PART + 5 * 4096, sw_, sw_)                                 \
        G4_DMA2(fc_a, fc_b, 3, IS, voffW[6], PART + 6 * 4096, voffW[7], PART + 7 * 4096, sw_, sw_)                                 \
        G4_GROUP(fc_a, fc_b, 4, G4_NONE, G4_NONE, G4_NONE, G4_NONE, G4_NONE, G4_NONE, G4_NONE, G4_NONE)                            \
        G4_GROUP(fc_a, fc_b, 5, G4_NONE, G4_NONE, G4_NONE, G4_NONE, G4_NONE, G4_NONE, G4_NONE, G4_NONE)                            \
        G4_GROUP(fc_a, fc_b, 6, G4_NONE, G4_NONE, G4_NONE, G4_NONE, G4_NONE, G4_NONE, G4_NONE, G4_NONE)                            \
        G4_GROUP(fc_a, fc_b, 7, G4_NONE, G4_NONE, G4_NONE, G4_NONE, G4_NONE, G4_NONE, G4_NONE, G4_NONE)                            \
    }
    unsigned long long tr_c0 = 0, tr_r0 = 0, tr_c1 = 0, tr_r1 = 0;
    // ---- DMA == 3: one body per K-tile, the fragment reads bunched at its two ends so that a stage is free early and every
    // LDS-DMA piece has more than a K-tile to land:
    //   S0 (64 MFMAs on K-step 0 of tile u): gaps 0-15 read K-step 1 of tile u <- stage u&1; gap 21: lgkmcnt(0) + barrier B1 (every
    //      wave has read all of stage u&1: WAR); groups 3-7: 10 pieces of tile u+2 -> stage u&1
    //   S1 (64 MFMAs on K-step 1): groups 0-2: the other 6 pieces; group 5: vmcnt(16) (tile u+1's pieces, issued during tile u-1,
    //      have landed; this tile's 16 stay in flight) + barrier B2 (RAW); groups 6-7 read K-step 0 of tile u+1 <- stage (u+1)&1
#define G4_V_S0(IS, sa_, sw_)                                                                                                     \
    {                                                                                                                             \
        G4_GROUP(fa0, fb0, 0, G4_DSR(fa1[0], ha, 0), G4_DSR(fb1[0], hb, 0), G4_DSR(fa1[1], ha, 2048), G4_DSR(fb1[1], hb, 2048),     \
                 G4_DSR(fa1[2], ha, 4096), G4_DSR(fb1[2], hb, 4096), G4_DSR(fa1[3], ha, 6144), G4_DSR(fb1[3], hb, 6144))           \
        G4_GROUP(fa0, fb0, 1, G4_DSR(fa1[4], ha, 8192), G4_DSR(fb1[4], hb, 8192), G4_DSR(fa1[5], ha, 10240), G4_DSR(fb1[5], hb, 10240), \
                 G4_DSR(fa1[6], ha, 12288), G4_DSR(fb1[6], hb, 12288), G4_DSR(fa1[7], ha, 14336), G4_DSR(fb1[7], hb, 14336))       \
        G4_GROUP(fa0, fb0, 2, G4_NONE, G4_NONE, G4_NONE, G4_NONE, G4_NONE,                                                         \
                 asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");, G4_NONE, G4_NONE)                                 \
        G4_DMA2(fa0, fb0, 3, IS, voffA[0], 0 * 4096, voffA[1], 1 * 4096, sa_, sa_)                                                 \
        G4_DMA2(fa0, fb0, 4, IS, voffA[2], 2 * 4096, voffA[3], 3 * 4096, sa_, sa_)                                                 \
        G4_DMA2(fa0, fb0, 5, IS, voffA[4], 4 * 4096, voffA[5], 5 * 4096, sa_, sa_)                                                 \
        G4_DMA2(fa0, fb0, 6, IS, voffA[6], 6 * 4096, voffA[7], 7 * 4096, sa_, sa_)                                                 \
        G4_DMA2(fa0, fb0, 7, IS, voffW[0], PART + 0 * 4096, voffW[1], PART + 1 * 4096, sw_, sw_)                                   \
    }
#define G4_V_S1(IS, RD, sw_)                                                                                                      \
    {                                                                                                                             \
        G4_DMA2(fa1, fb1, 0, IS, voffW[2], PART + 2 * 4096, voffW[3], PART + 3 * 4096, sw_, sw_)                                   \
        G4_DMA2(fa1, fb1, 1, IS, voffW[4], PART + 4 * 4096, voffW[5], PART + 5 * 4096, sw_, sw_)                                   \
        G4_DMA2(fa1, fb1, 2, IS, voffW[6], PART + 6 * 4096, voffW[7], PART + 7 * 4096, sw_, sw_)                                   \
        G4_GROUP(fa1, fb1, 3, G4_NONE, G4_NONE, G4_NONE, G4_NONE, G4_NONE, G4_NONE, G4_NONE, G4_NONE)                              \
        G4_GROUP(fa1, fb1, 4, G4_NONE, G4_NONE, G4_NONE, G4_NONE, G4_NONE, G4_NONE, G4_NONE, G4_NONE)                              \
        G4_GROUP(fa1, fb1, 5, if (IS) asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");                               \
                 else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");,                                                \
                 G4_NONE, G4_NONE, G4_NONE, G4_NONE, G4_NONE, G4_NONE, G4_NONE)                                                    \
        G4_GROUP(fa1, fb1, 6, if (RD) G4_DSR(fa0[0], ha2, 0), if (RD) G4_DSR(fb0[0], hb2, 0), if (RD) G4_DSR(fa0[1], ha2, 2048),   \
                 if (RD) G4_DSR(fb0[1], hb2, 2048), if (RD) G4_DSR(fa0[2], ha2, 4096), if (RD) G4_DSR(fb0[2], hb2, 4096),          \
                 if (RD) G4_DSR(fa0[3], ha2, 6144), if (RD) G4_DSR(fb0[3], hb2, 6144))                                              \
        G4_GROUP(fa1, fb1, 7, if (RD) G4_DSR(fa0[4], ha2, 8192), if (RD) G4_DSR(fb0[4], hb2, 8192), if (RD) G4_DSR(fa0[5], ha2, 10240), \
                 if (RD) G4_DSR(fb0[5], hb2, 10240), if (RD) G4_DSR(fa0[6], ha2, 12288), if (RD) G4_DSR(fb0[6], hb2, 12288),        \
                 if (RD) G4_DSR(fa0[7], ha2, 14336), if (RD) G4_DSR(fb0[7], hb2, 14336))                                            \
    }
    // ---- DMA == 4: the DMA == 3 body with the 16 pieces spread evenly over gaps 25-125 of the 128 (one every ~6.6 MFMAs, the
    // last three behind B2 between the fragment reads), so that the CU's L2 -> LDS stream never pauses: vmcnt(13) at B2
#define G4_OFF(n_) ((n_) < 8 ? (n_) * 4096 : PART + ((n_) - 8) * 4096)
#define G4_P(n_) if (is) G4_SETM0(mm, G4_OFF(n_))
#define G4_QA(n_) if (is) G4_GLDS(voffA[n_], sa)
#define G4_QW(n_) if (is) G4_GLDS(voffW[n_], sw)
#define G4_E_S0(IS)                                                                                                               \
    {                                                                                                                             \
        G4_GROUP(fa0, fb0, 0, G4_DSR(fa1[0], ha, 0), G4_DSR(fb1[0], hb, 0), G4_DSR(fa1[1], ha, 2048), G4_DSR(fb1[1], hb, 2048),     \
                 G4_DSR(fa1[2], ha, 4096), G4_DSR(fb1[2], hb, 4096), G4_DSR(fa1[3], ha, 6144), G4_DSR(fb1[3], hb, 6144))           \
        G4_GROUP(fa0, fb0, 1, G4_DSR(fa1[4], ha, 8192), G4_DSR(fb1[4], hb, 8192), G4_DSR(fa1[5], ha, 10240), G4_DSR(fb1[5], hb, 10240), \
                 G4_DSR(fa1[6], ha, 12288), G4_DSR(fb1[6], hb, 12288), G4_DSR(fa1[7], ha, 14336), G4_DSR(fb1[7], hb, 14336))       \
        G4_GROUP(fa0, fb0, 2, G4_NONE, G4_NONE, G4_NONE, G4_NONE, G4_NONE,                                                         \
                 asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");, G4_NONE, G4_NONE)                                 \
        G4_GROUP(fa0, fb0, 3, G4_NONE, G4_P(0), G4_QA(0), G4_NONE, G4_NONE, G4_NONE, G4_NONE, G4_P(1))                             \
        G4_GROUP(fa0, fb0, 4, G4_QA(1), G4_NONE, G4_NONE, G4_NONE, G4_NONE, G4_NONE, G4_P(2), G4_QA(2))                            \
        G4_GROUP(fa0, fb0, 5, G4_NONE, G4_NONE, G4_NONE, G4_NONE, G4_P(3), G4_QA(3), G4_NONE, G4_NONE)                             \
        G4_GROUP(fa0, fb0, 6, G4_NONE, G4_NONE, G4_NONE, G4_P(4), G4_QA(4), G4_NONE, G4_NONE, G4_NONE)                             \
        G4_GROUP(fa0, fb0, 7, G4_NONE, G4_P(5), G4_QA(5), G4_NONE, G4_NONE, G4_NONE, G4_NONE, G4_P(6))                             \
    }
#define G4_E_S1(IS, RD)                                                                                                           \
    {                                                                                                                             \
        G4_GROUP(fa1, fb1, 0, G4_QA(6), G4_NONE, G4_NONE, G4_NONE, G4_NONE, G4_NONE, G4_P(7), G4_QA(7))                            \
        G4_GROUP(fa1, fb1, 1, G4_NONE, G4_NONE, G4_NONE, G4_NONE, G4_NONE, G4_P(8), G4_QW(0), G4_NONE)                             \
        G4_GROUP(fa1, fb1, 2, G4_NONE, G4_NONE, G4_NONE, G4_P(9), G4_QW(1), G4_NONE, G4_NONE, G4_NONE)                             \
        G4_GROUP(fa1, fb1, 3, G4_NONE, G4_NONE, G4_P(10), G4_QW(2), G4_NONE, G4_NONE, G4_NONE, G4_NONE)                            \
        G4_GROUP(fa1, fb1, 4, G4_P(11), G4_QW(3), G4_NONE, G4_NONE, G4_NONE, G4_NONE, G4_NONE, G4_P(12))                           \
        G4_GROUP(fa1, fb1, 5, G4_QW(4), G4_NONE, G4_NONE, G4_NONE,                                                                 \
                 if (IS) asm volatile("s_waitcnt vmcnt(13)\n\ts_barrier" ::: "memory");                                            \
                 else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");,                                                \
                 G4_NONE, G4_NONE, G4_NONE)                                                                                        \
        G4_GROUP(fa1, fb1, 6, if (RD) G4_DSR(fa0[0], ha2, 0) G4_P(13), if (RD) G4_DSR(fb0[0], hb2, 0) G4_QW(5),                     \
                 if (RD) G4_DSR(fa0[1], ha2, 2048), if (RD) G4_DSR(fb0[1], hb2, 2048), if (RD) G4_DSR(fa0[2], ha2, 4096),          \
                 if (RD) G4_DSR(fb0[2], hb2, 4096), if (RD) G4_DSR(fa0[3], ha2, 6144) G4_P(14), if (RD) G4_DSR(fb0[3], hb2, 6144) G4_QW(6)) \
        G4_GROUP(fa1, fb1, 7, if (RD) G4_DSR(fa0[4], ha2, 8192), if (RD) G4_DSR(fb0[4], hb2, 8192), if (RD) G4_DSR(fa0[5], ha2, 10240), \
                 if (RD) G4_DSR(fb0[5], hb2, 10240) G4_P(15), if (RD) G4_DSR(fa0[6], ha2, 12288) G4_QW(7), if (RD) G4_DSR(fb0[6], hb2, 12288), \
                 if (RD) G4_DSR(fa0[7], ha2, 14336), if (RD) G4_DSR(fb0[7], hb2, 14336))                                            \
    }
    if (DMA == 3 || DMA == 4) {
        {
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                if (st < nk) {
                    const uint8_t* a0 = gA + st * 128, *w0 = gW + st * 128;
                    const uint32_t mm2 = m0_wave + (uint32_t)(st * STAGE);
#define G4_PRO(va_, off_, sbase_) { unsigned keep__; asm volatile("s_mov_b32 %0, m0\n\ts_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %4\n\ts_mov_b32 m0, %0" : "=&s"(keep__) : "v"(va_), "s"(mm2), "i"(off_), "s"(sbase_) : "memory", "scc"); }
                    G4_PRO(voffA[0], 0 * 4096, a0) G4_PRO(voffA[1], 1 * 4096, a0) G4_PRO(voffA[2], 2 * 4096, a0) G4_PRO(voffA[3], 3 * 4096, a0)
                    G4_PRO(voffA[4], 4 * 4096, a0) G4_PRO(voffA[5], 5 * 4096, a0) G4_PRO(voffA[6], 6 * 4096, a0) G4_PRO(voffA[7], 7 * 4096, a0)
                    G4_PRO(voffW[0], PART + 0 * 4096, w0) G4_PRO(voffW[1], PART + 1 * 4096, w0) G4_PRO(voffW[2], PART + 2 * 4096, w0) G4_PRO(voffW[3], PART + 3 * 4096, w0)
                    G4_PRO(voffW[4], PART + 4 * 4096, w0) G4_PRO(voffW[5], PART + 5 * 4096, w0) G4_PRO(voffW[6], PART + 6 * 4096, w0) G4_PRO(voffW[7], PART + 7 * 4096, w0)
#undef G4_PRO
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const uint32_t ha = a_frag + ch0, hb = b_frag + ch0;
            G4_DSR(fa0[0], ha, 0) G4_DSR(fa0[1], ha, 2048) G4_DSR(fa0[2], ha, 4096) G4_DSR(fa0[3], ha, 6144)
            G4_DSR(fa0[4], ha, 8192) G4_DSR(fa0[5], ha, 10240) G4_DSR(fa0[6], ha, 12288) G4_DSR(fa0[7], ha, 14336)
            G4_DSR(fb0[0], hb, 0) G4_DSR(fb0[1], hb, 2048) G4_DSR(fb0[2], hb, 4096) G4_DSR(fb0[3], hb, 6144)
            G4_DSR(fb0[4], hb, 8192) G4_DSR(fb0[5], hb, 10240) G4_DSR(fb0[6], hb, 12288) G4_DSR(fb0[7], hb, 14336)
        }
        if (batch.trace) asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(tr_c0), "=s"(tr_r0)::"memory");
        // steady state with compile-time flags (a run-time `if` around an asm statement is a scalar branch in an MFMA gap)
#define G4_V_ITER(FLAGS)                                                                                            \
        {                                                                                                           \
            const uint32_t st_cur = (uint32_t)((u & 1) * STAGE), st_nxt = (uint32_t)(((u + 1) & 1) * STAGE);        \
            FLAGS                                                                                                   \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                      \
            const uint32_t ha = a_frag + st_cur + ch1, hb = b_frag + st_cur + ch1;                                  \
            const uint32_t ha2 = a_frag + st_nxt + ch0, hb2 = b_frag + st_nxt + ch0;                                \
            const uint32_t mm = m0_wave + st_cur;                                                                   \
            const uint8_t* sa = gA + (size_t)(u + 2) * 128;                                                         \
            const uint8_t* sw = gW + (size_t)(u + 2) * 128;                                                         \
            const uint32_t kof = (uint32_t)(u + 2) * 128u;                                                          \
            (void)kof; (void)mm; (void)sa; (void)sw;                                                                \
            if (DMA == 4) {                                                                                         \
                G4_E_S0(is)                                                                                         \
                G4_E_S1(is, rd)                                                                                     \
            } else {                                                                                                \
                G4_V_S0(is, sa, sw)                                                                                 \
                G4_V_S1(is, rd, sw)                                                                                 \
            }                                                                                                       \
        }
        int u = 0;
        for (; u + 2 < nk; ++u) G4_V_ITER(constexpr bool is = !(ABL & 4); constexpr bool rd = true;)
        for (; u < nk; ++u) G4_V_ITER(constexpr bool is = false; const bool rd = u + 1 < nk;)
#undef G4_V_ITER
    } else if (DMA) {
        // ---- LDS-DMA form.  K-tile u+2 -> stage u&1: 12 pieces in body(2u+1) (behind the barrier that frees the stage), the last
        // 4 in body(2u+2); all 16 are waited for (vmcnt(0)) at the top of body(2u+3), a K-tile and a half after the first was issued.
        {
            const uint32_t mm = m0_wave;
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                if (st < nk) {
                    const uint8_t* a0 = gA + st * 128, *w0 = gW + st * 128;
                    const uint32_t mm2 = m0_wave + (uint32_t)(st * STAGE);
#define G4_PRO(va_, off_, sbase_) { unsigned keep__; asm volatile("s_mov_b32 %0, m0\n\ts_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %4\n\ts_mov_b32 m0, %0" : "=&s"(keep__) : "v"(va_), "s"(mm2), "i"(off_), "s"(sbase_) : "memory", "scc"); }
                    G4_PRO(voffA[0], 0 * 4096, a0) G4_PRO(voffA[1], 1 * 4096, a0) G4_PRO(voffA[2], 2 * 4096, a0) G4_PRO(voffA[3], 3 * 4096, a0)
                    G4_PRO(voffA[4], 4 * 4096, a0) G4_PRO(voffA[5], 5 * 4096, a0) G4_PRO(voffA[6], 6 * 4096, a0) G4_PRO(voffA[7], 7 * 4096, a0)
                    G4_PRO(voffW[0], PART + 0 * 4096, w0) G4_PRO(voffW[1], PART + 1 * 4096, w0) G4_PRO(voffW[2], PART + 2 * 4096, w0) G4_PRO(voffW[3], PART + 3 * 4096, w0)
                    G4_PRO(voffW[4], PART + 4 * 4096, w0) G4_PRO(voffW[5], PART + 5 * 4096, w0) G4_PRO(voffW[6], PART + 6 * 4096, w0) G4_PRO(voffW[7], PART + 7 * 4096, w0)
#undef G4_PRO
                }
            }
            (void)mm;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const uint32_t ha = a_frag + ch0, hb = b_frag + ch0;
            G4_DSR(fa0[0], ha, 0) G4_DSR(fa0[1], ha, 2048) G4_DSR(fa0[2], ha, 4096) G4_DSR(fa0[3], ha, 6144)
            G4_DSR(fa0[4], ha, 8192) G4_DSR(fa0[5], ha, 10240) G4_DSR(fa0[6], ha, 12288) G4_DSR(fa0[7], ha, 14336)
            G4_DSR(fb0[0], hb, 0) G4_DSR(fb0[1], hb, 2048) G4_DSR(fb0[2], hb, 4096) G4_DSR(fb0[3], hb, 6144)
            G4_DSR(fb0[4], hb, 8192) G4_DSR(fb0[5], hb, 10240) G4_DSR(fb0[6], hb, 12288) G4_DSR(fb0[7], hb, 14336)
        }
        if (batch.trace) asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(tr_c0), "=s"(tr_r0)::"memory");
        for (int u = 0; u < nk; ++u) {
            const uint32_t st_cur = (uint32_t)((u & 1) * STAGE), st_nxt = (uint32_t)(((u + 1) & 1) * STAGE);
            const bool is_even = u >= 1 && u + 1 < nk;       // last 4 W pieces of K-tile u+1 -> stage (u+1)&1 (freed by the barrier of body 2u-1)
            const bool is_odd = u + 2 < nk;                  // first 12 pieces of K-tile u+2 -> stage u&1
            {   // body 2u
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const uint32_t ha = a_frag + st_cur + ch1, hb = b_frag + st_cur + ch1;
                const uint32_t mm = m0_wave + st_nxt;
                const uint8_t* sw = gW + (size_t)(u + 1) * 128;
                const uint32_t kof = (uint32_t)(u + 1) * 128u;
                G4_BODY_DMA_EVEN(is_even && !(ABL & 4), fa0, fb0, fa1, fb1, sw)
            }
            {   // body 2u+1
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(0)" ::: "memory");
                G4_BARRIER()
                const uint32_t ha = a_frag + st_nxt + ch0, hb = b_frag + st_nxt + ch0;
                const uint32_t mm = m0_wave + st_cur;
                const uint8_t* sa = gA + (size_t)(u + 2) * 128;
                const uint8_t* sw = gW + (size_t)(u + 2) * 128;
                const bool rd = u + 1 < nk;
                const uint32_t kof = (uint32_t)(u + 2) * 128u;
                G4_BODY_DMA(rd, is_odd && !(ABL & 4), fa1, fb1, fa0, fb0, sa, sw)
            }
        }
    } else {
    // ---- prologue: K-tile 0 -> stage 0; K-tile 1 in flight, its A part written to stage 1; frags(0) in registers
    {
        const uint8_t* a0 = gA, *w0 = gW;
#pragma unroll
        for (int j = 0; j < 8; ++j) { G4_GLOAD(R[j], voffA[j], a0) }
#pragma unroll
        for (int j = 0; j < 8; ++j) { G4_GLOAD(R[8 + j], voffW[j], w0) }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t wa = wr_addr;
        G4_DSW(wa, R[0], 0 * 4096) G4_DSW(wa, R[1], 1 * 4096) G4_DSW(wa, R[2], 2 * 4096) G4_DSW(wa, R[3], 3 * 4096)
        G4_DSW(wa, R[4], 4 * 4096) G4_DSW(wa, R[5], 5 * 4096) G4_DSW(wa, R[6], 6 * 4096) G4_DSW(wa, R[7], 7 * 4096)
        G4_DSW(wa, R[8], PART + 0 * 4096) G4_DSW(wa, R[9], PART + 1 * 4096) G4_DSW(wa, R[10], PART + 2 * 4096) G4_DSW(wa, R[11], PART + 3 * 4096)
        G4_DSW(wa, R[12], PART + 4 * 4096) G4_DSW(wa, R[13], PART + 5 * 4096) G4_DSW(wa, R[14], PART + 6 * 4096) G4_DSW(wa, R[15], PART + 7 * 4096)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the writes have read their registers
        if (nk > 1) {
            const uint8_t* a1 = gA + 128, *w1 = gW + 128;
#pragma unroll
            for (int j = 0; j < 8; ++j) { G4_GLOAD(R[j], voffA[j], a1) }
#pragma unroll
            for (int j = 0; j < 8; ++j) { G4_GLOAD(R[8 + j], voffW[j], w1) }
        }
        G4_BARRIER()
        const uint32_t ha = a_frag + ch0, hb = b_frag + ch0;
        G4_DSR(fa0[0], ha, 0) G4_DSR(fa0[1], ha, 2048) G4_DSR(fa0[2], ha, 4096) G4_DSR(fa0[3], ha, 6144)
        G4_DSR(fa0[4], ha, 8192) G4_DSR(fa0[5], ha, 10240) G4_DSR(fa0[6], ha, 12288) G4_DSR(fa0[7], ha, 14336)
        G4_DSR(fb0[0], hb, 0) G4_DSR(fb0[1], hb, 2048) G4_DSR(fb0[2], hb, 4096) G4_DSR(fb0[3], hb, 6144)
        G4_DSR(fb0[4], hb, 8192) G4_DSR(fb0[5], hb, 10240) G4_DSR(fb0[6], hb, 12288) G4_DSR(fb0[7], hb, 14336)
        if (nk > 1) {   // the A part of K-tile 1 (what body(-1) would have written), then K-tile 2's A part behind it
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            const uint32_t wa1 = wr_addr + STAGE;
            G4_DSW(wa1, R[0], 0 * 4096) G4_DSW(wa1, R[1], 1 * 4096) G4_DSW(wa1, R[2], 2 * 4096) G4_DSW(wa1, R[3], 3 * 4096)
            G4_DSW(wa1, R[4], 4 * 4096) G4_DSW(wa1, R[5], 5 * 4096) G4_DSW(wa1, R[6], 6 * 4096) G4_DSW(wa1, R[7], 7 * 4096)
            if (nk > 2) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const uint8_t* a2 = gA + 256;
#pragma unroll
                for (int j = 0; j < 8; ++j) { G4_GLOAD(R[j], voffA[j], a2) }
            }
        }
    }

    // dev diagnostic (batch.trace, FLUX2MI_GEMM_TRACE with the dev library): shader-clock and 100 MHz real-time stamps
    // around the K loop of every wave -> cycles per K-tile and the clock the chip holds inside the loop
    if (batch.trace) asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(tr_c0), "=s"(tr_r0)::"memory");
    // ---- K loop.  Even body 2u: frags(2u) in (fa0, fb0); odd body 2u+1: (fa1, fb1).
    int u = 0;
    for (; u + 3 <= nk - 1; ++u) {   // steady state: K-tiles u+1 (writes) and u+2 (loads of both halves) exist, and u+3's A part
        const uint32_t st_cur = (uint32_t)((u & 1) * STAGE), st_nxt = (uint32_t)(((u + 1) & 1) * STAGE);
        {   // body 2u: reads frags(2u+1) (second half of K-tile u); writes W part of K-tile u+1; loads W part of K-tile u+2
            G4_WAIT(8)
            const uint32_t ha = a_frag + st_cur + ch1, hb = b_frag + st_cur + ch1;
            const uint32_t wa = wr_addr + st_nxt;
            const uint8_t* sb = gW + (size_t)(u + 2) * 128;
            G4_BODY(!(ABL & 1), !(ABL & 2), !(ABL & 4), fa0, fb0, fa1, fb1, 8, PART, voffW, sb)
        }
        {   // body 2u+1: reads frags(2u+2) (first half of K-tile u+1); writes A part of K-tile u+2 -> stage u&1; loads A part of K-tile u+3
            G4_WAIT(8)
            G4_BARRIER()
            const uint32_t ha = a_frag + st_nxt + ch0, hb = b_frag + st_nxt + ch0;
            const uint32_t wa = wr_addr + st_cur;
            const uint8_t* sb = gA + (size_t)(u + 3) * 128;
            G4_BODY(!(ABL & 1), !(ABL & 2), !(ABL & 4), fa1, fb1, fa0, fb0, 0, 0, voffA, sb)
        }
    }
    for (; u < nk; ++u) {            // last K-tiles: guarded aux work, full waits
        const uint32_t st_cur = (uint32_t)((u & 1) * STAGE), st_nxt = (uint32_t)(((u + 1) & 1) * STAGE);
        {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(0)" ::: "memory");
            const uint32_t ha = a_frag + st_cur + ch1, hb = b_frag + st_cur + ch1;
            const uint32_t wa = wr_addr + st_nxt;
            const uint8_t* sb = gW + (size_t)(u + 2) * 128;
            const bool wr = u + 1 < nk, ld = u + 2 < nk;
            G4_BODY(true, wr, ld, fa0, fb0, fa1, fb1, 8, PART, voffW, sb)
        }
        {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(0)" ::: "memory");
            G4_BARRIER()
            const uint32_t ha = a_frag + st_nxt + ch0, hb = b_frag + st_nxt + ch0;
            const uint32_t wa = wr_addr + st_cur;
            const uint8_t* sb = gA + (size_t)(u + 3) * 128;
            const bool rd = u + 1 < nk, wr = u + 2 < nk, ld = u + 3 < nk;
            G4_BODY(rd, wr, ld, fa1, fb1, fa0, fb0, 0, 0, voffA, sb)
        }
    }
    }
    // MFMA result -> first non-MFMA reader: hipcc pads nothing behind inline asm (guide §5.7 item 2)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    if (batch.trace) {
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(tr_c1), "=s"(tr_r1)::"memory");
        if (lane == 0) {
            unsigned long long* o = batch.trace + ((size_t)bid * 4 + wid) * 4;
            o[0] = tr_c1 - tr_c0; o[1] = tr_r1 - tr_r0; o[2] = (unsigned long long)nk;
            o[3] = DIAG ? (dg_lds & 0xfffff) | ((dg_vm & 0xfffff) << 20) | ((dg_bar & 0xfffff) << 40) : tr_c0;
        }
    }
#undef G4_GLOAD
#undef G4_DSW
#undef G4_DSR
#undef G4_M
#undef G4_GROUP
#undef G4_NONE
#undef G4_BODY
#undef G4_WAIT
#undef G4_BARRIER

    // ---- epilogue.  acc[i][j][r]: output row m0 + wm*128 + i*16 + frow, columns n0 + wn*128 + j*16 + fq*4 + r
    const int row_base = m0 + wm * 128 + frow;
    const int col_base = n0 + wn * 128 + fq * 4;
    const int epi = p.epi;
    if (epi == EPI_F32) {
        float* C = reinterpret_cast<float*>(p.C);
        const bool vec_ok = (p.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = row_base + i * 16;
            if (row < p.M) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int col = col_base + j * 16;
                    float* dst = C + (size_t)row * p.ldc + col;
                    if (vec_ok && col + 3 < p.N) {
                        *reinterpret_cast<f32x4*>(dst) = acc[i][j] * p.alpha;
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (col + r < p.N) dst[r] = acc[i][j][r] * p.alpha;
                    }
                }
            }
        }
    } else if (epi == EPI_BF16) {
        // 16-byte stores: the lane pairs (fq, fq^1) exchange halves so that each lane owns 8 consecutive columns of one block pair
        uint16_t* C = reinterpret_cast<uint16_t*>(p.C);
        const bool fast = (p.ldc & 7) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0 && n0 + TN <= p.N;
        const int wcol = n0 + wn * 128 + (fq & 1) * 16 + (fq >> 1) * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = row_base + i * 16;
#pragma unroll
            for (int jp = 0; jp < 4; ++jp) {
                const f32x4 vx = acc[i][2 * jp] * p.alpha, vy = acc[i][2 * jp + 1] * p.alpha;
                if (fast) {
                    uint32_t x0 = pack_bf16x2(vx[0], vx[1]), x1 = pack_bf16x2(vx[2], vx[3]);
                    uint32_t y0 = pack_bf16x2(vy[0], vy[1]), y1 = pack_bf16x2(vy[2], vy[3]);
                    g4_swap16(x0, y0);
                    g4_swap16(x1, y1);
                    if (row < p.M) *reinterpret_cast<uint4*>(C + (size_t)row * p.ldc + wcol + jp * 32) = make_uint4(x0, x1, y0, y1);
                } else if (row < p.M) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int c0 = col_base + (2 * jp) * 16 + r, c1 = c0 + 16;
                        if (c0 < p.N) C[(size_t)row * p.ldc + c0] = f32_to_bf16_bits(vx[r]);
                        if (c1 < p.N) C[(size_t)row * p.ldc + c1] = f32_to_bf16_bits(vy[r]);
                    }
                }
            }
        }
    } else if (epi == EPI_GATE_RESID) {
        float* C = reinterpret_cast<float*>(p.C);
        const bool vec_ok = (p.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0;
        f32x4 g[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int col = col_base + j * 16 + r;
                g[j][r] = (col < p.N) ? p.vec[col] : 0.f;
            }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = row_base + i * 16;
            if (row < p.M) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int col = col_base + j * 16;
                    float* dst = C + (size_t)row * p.ldc + col;
                    if (vec_ok && col + 3 < p.N) {
                        f32x4* d4 = reinterpret_cast<f32x4*>(dst);
                        *d4 = *d4 + g[j] * acc[i][j];
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (col + r < p.N) dst[r] = dst[r] + g[j][r] * acc[i][j][r];
                    }
                }
            }
        }
    }
}

}  // namespace

// EPI_F32 / EPI_GATE_RESID, bf16 operands, K % 64 == 0 (the caller has validated the problems).  0 ok / 2 launch error.
int launch_gemm256_g4(hipStream_t stream, const GemmProblem* probs, int nprob) {
    GemmBatch b;
    b.nprob = nprob;
    int tiles[2] = {0, 0};
    for (int i = 0; i < nprob; ++i) {
        b.p[i] = probs[i];
        tiles[i] = ((probs[i].M + TM - 1) / TM) * ((probs[i].N + TN - 1) / TN);
    }
    if (nprob == 1) b.p[1] = probs[0];
    b.tile_start1 = tiles[0];
    b.flags = 0;
    b.trace = nullptr;
    static const char* trace_env = F2_DEV_ENV("FLUX2MI_GEMM_TRACE");
    static int seen = 0;
    const bool trace_now = trace_env && ++seen % 16 == 0;   // a warm launch every now and then
    const size_t n = (size_t)(tiles[0] + tiles[1]) * 16;
    if (trace_now && hipMalloc((void**)&b.trace, n * 8) != hipSuccess) b.trace = nullptr;
    static const bool diag = F2_DEV_ENV("FLUX2MI_GEMM_DIAG") != nullptr;
    const dim3 grid(tiles[0] + tiles[1]), block(256);
#ifdef FLUX2MI_DEV
    static const int abl = [] { const char* e = F2_DEV_ENV("FLUX2MI_GEMM_ABLATE"); return e ? atoi(e) : 0; }();
    if (diag && b.trace) hipLaunchKernelGGL((gemm_bf16_256x256_g4_kernel<true>), grid, block, 0, stream, b);
    else if (abl == 1) hipLaunchKernelGGL((gemm_bf16_256x256_g4_kernel<false, 1>), grid, block, 0, stream, b);
    else if (abl == 2) hipLaunchKernelGGL((gemm_bf16_256x256_g4_kernel<false, 2>), grid, block, 0, stream, b);
    else if (abl == 3) hipLaunchKernelGGL((gemm_bf16_256x256_g4_kernel<false, 3>), grid, block, 0, stream, b);
    else if (abl == 4) hipLaunchKernelGGL((gemm_bf16_256x256_g4_kernel<false, 4>), grid, block, 0, stream, b);
    else if (abl == 6) hipLaunchKernelGGL((gemm_bf16_256x256_g4_kernel<false, 6>), grid, block, 0, stream, b);
    else if (abl == 7) hipLaunchKernelGGL((gemm_bf16_256x256_g4_kernel<false, 7>), grid, block, 0, stream, b);
    else if (abl == 100) hipLaunchKernelGGL((gemm_bf16_256x256_g4_kernel<false, 0, 1>), grid, block, 0, stream, b);
    else if (abl == 104) hipLaunchKernelGGL((gemm_bf16_256x256_g4_kernel<false, 4, 1>), grid, block, 0, stream, b);
    else if (abl == 300) hipLaunchKernelGGL((gemm_bf16_256x256_g4_kernel<false, 0, 3>), grid, block, 0, stream, b);
    else if (abl == 304) hipLaunchKernelGGL((gemm_bf16_256x256_g4_kernel<false, 4, 3>), grid, block, 0, stream, b);
    else if (abl == 400) hipLaunchKernelGGL((gemm_bf16_256x256_g4_kernel<false, 0, 4>), grid, block, 0, stream, b);
    else if (abl == 404) hipLaunchKernelGGL((gemm_bf16_256x256_g4_kernel<false, 4, 4>), grid, block, 0, stream, b);
    else if (abl == 200) hipLaunchKernelGGL((gemm_bf16_256x256_g4_kernel<false, 0, 2>), grid, block, 0, stream, b);
    else if (abl == 204) hipLaunchKernelGGL((gemm_bf16_256x256_g4_kernel<false, 4, 2>), grid, block, 0, stream, b);
    else
#endif
    hipLaunchKernelGGL((gemm_bf16_256x256_g4_kernel<false>), grid, block, 0, stream, b);
    if (b.trace) {
        std::vector<unsigned long long> h(n);
        (void)hipStreamSynchronize(stream);
        (void)hipMemcpy(h.data(), b.trace, n * 8, hipMemcpyDeviceToHost);
        (void)hipFree(b.trace);
        double cyc = 0, rt = 0, kt = 0, dl = 0, dv = 0, db = 0;
        for (size_t w = 0; w < n / 4; ++w) {
            cyc += (double)h[w * 4]; rt += (double)h[w * 4 + 1]; kt += (double)h[w * 4 + 2];
            if (diag) { dl += (double)(h[w * 4 + 3] & 0xfffff); dv += (double)((h[w * 4 + 3] >> 20) & 0xfffff); db += (double)((h[w * 4 + 3] >> 40) & 0xfffff); }
        }
        if (diag) fprintf(stderr, "[g4 diag] per K-tile: lgkmcnt waits %.0f, vmcnt waits %.0f, barrier %.0f cycles (stamps ~40 each included)\n", dl / kt, dv / kt, db / kt);
        fprintf(stderr, "[g4 trace] M %d N %d K %d: %.0f shader cycles per K-tile, in-loop clock %.0f MHz, %.3f us per K-tile\n", probs[0].M,
                probs[0].N, probs[0].K, cyc / kt, cyc / rt * 100.0, rt / kt * 0.01);
    }
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
