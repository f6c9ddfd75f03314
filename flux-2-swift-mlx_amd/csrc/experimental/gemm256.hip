// gemm256.hip — the 8-WAVE large-tile GEMM of round 1 / early round 2 (bf16 and fp8): superseded by gemm256s.hip (one wave per SIMD) and
// built only into libflux2mi_dev.so (FLUX2MI_GEMM_V=2: A/B runs; the strided-operand layout experiments).
// 256x256x64 tile, 8 waves, LDS-DMA staging that
// stays in flight across barriers (counted vmcnt), 4 MFMA phases per K-tile.
//
// Same contract and epilogues as gemm.hip (C = A W^T, W [out,in]); used for the DiT's large
// problems, where it replaces the 128x128 kernel whose LDS write traffic (ds_write_b128 of
// both operands) competed with the fragment reads.
//
// Geometry
//   * 512 threads = 8 waves as 2 (M) x 4 (N); wave tile 128 x 64 = 8 x 4 MFMA 16x16x32
//     accumulators (128 f32 registers).
//   * LDS = 2 stages x 4 half-tile slots x 16 KB = 128 KB (1 workgroup per CU, 2 waves/SIMD).
//     A K-tile (64 k) is staged as four half-tiles, cut along what a PHASE consumes:
//        A0 = for each wave-row, its first 64 rows;  A1 = its second 64 rows;
//        B0 = for each wave-column, its first 32 weight rows;  B1 = its second 32.
//     so a slot is free again right after the phase that reads it.
//   * global -> LDS by global_load_lds_dwordx4 (1 KB per wave-instruction, 2 per thread per
//     half-tile).  The LDS image stays lane-linear; the bank swizzle (16-B chunk ^ (row>>1)&7)
//     is applied to the per-lane SOURCE address and to the fragment reads (guide rule 21).
//   * schedule: half-tiles form one sequence e = 4u + {A0,B0,B1,A1} (u = K-tile).  Phase
//     phi = 4t + p issues element phi + 7 and consumes: p0 A0,B0 -> quadrant (m0,n0);
//     p1 B1 -> (m0,n1); p2 A1 -> (m1,n1); p3 nothing new -> (m1,n0).  At the end of every
//     phase `s_waitcnt vmcnt(10)` (5 younger half-tiles x 2 loads may stay in flight) + one
//     raw s_barrier make element phi + 2 — exactly what phase phi + 1 reads — visible.
//     WAR: a slot is re-filled at least one barrier after the phase that read it.
//     The last two K-tiles wait vmcnt(0) (nothing younger is issued any more).
#include "../dev.h"
#include "../kernels.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>

namespace {
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));


constexpr int BM2 = 256, BN2 = 256, BK2 = 64;
constexpr int SLOT_BYTES = 128 * BK2 * 2;      // 16 KB
constexpr int STAGE2_BYTES = 4 * SLOT_BYTES;   // 64 KB
enum { KIND_A0 = 0, KIND_B0 = 1, KIND_B1 = 2, KIND_A1 = 3 };

typedef __attribute__((ext_vector_type(2))) long i64x2;
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// v_permlane16_swap_b32 a, b: rows (16 lanes) 1 and 3 of a <-> rows 0 and 2 of b.  With a = X, b = Y of lane
// (frow, fq) afterwards {a, b} = {X of fq&~1, X of fq|1} on even fq and {Y of fq&~1, Y of fq|1} on odd fq:
// two 8-byte column groups that were split over two lanes become one 16-byte run per lane (guide T21).
__device__ __forceinline__ void swap16(uint32_t& a, uint32_t& b) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}

// v_permlane32_swap_b32 a, b: lanes 32-63 of a <-> lanes 0-31 of b (inline asm: see attention.hip's xhalf_swap)
__device__ __forceinline__ void swap32(uint32_t& a, uint32_t& b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}

__device__ __forceinline__ void tile_coords256(int local, int ntiles, int tiles_m, int tiles_n, int& tm, int& tn) {
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

// STAGGER: waves 4-7 (wave-row 1; they share SIMDs with waves 0-3) run half a phase behind
// waves 0-3: every phase has two barriers (reads | MFMAs) and the second group takes one extra
// barrier up front, so one group's MFMA segment runs beside the other group's LDS-read segment
// on each SIMD instead of both reading, then both computing (MI355X guide, "Two waves per
// SIMD" item 9).  Data published by LDS-DMA is then only guaranteed one phase later for the
// leading group, hence vmcnt(8) instead of vmcnt(10).
// ABL bits (timing only): 1 no fragment reads, 2 no DMA, 4 DMA from an L2-resident source, 8 no vmcnt waits, 16 no MFMA,
// 32 no epilogue (the accumulators are dropped after the K loop), 64 bf16 / SwiGLU stores as whole rows (wrong places)
// FP8: operands are OCP e4m3 bytes.  The byte geometry of the tile is unchanged (128-B rows, so a
// K-tile holds 128 k instead of 64); a lane's two 16-B fragment reads are the 32-byte operand of
// ONE block-scaled v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3, unit block scales), the form that
// runs at twice the bf16 rate — the plain _fp8_fp8 16x16x32 runs at the bf16 rate.  Which k of
// the tile lands on which operand byte is the same permutation for A and W, which a dot product
// does not see.  The epilogue multiplies by a_scale[row] * w_scale[col].
// VAR (schedule experiments, FLUX2MI_GEMM_VAR): bit 0 = wait for the fragment reads AFTER the phase's first
// barrier; bit 1 = one counted vmcnt per K-tile (before the first barrier of phase 3) instead of one per phase.
// NI: MFMA row blocks per wave (8 = the 256-row tile; 7, 6, 5 = 224, 192, 160 rows).  The shorter tiles exist for
// wave quantisation only: M = 4608 x N = 3072 is 216 tiles of 256^2 on 256 CUs but 252 tiles of 224 x 256, one round
// either way, so the launch takes 7/8 of the time.  A wave-row keeps its first 64 rows as half-tile A0 and the
// remaining NI*16 - 64 as A1 (the A1 slot is still filled with 64 rows per wave-row: the surplus is never read).
template <bool STAGGER, int ABL, bool HALFBAR = false, bool DIAG = false, bool FP8 = false, int VAR = 0, int NI = 8>
__global__ __launch_bounds__(512, 2) void gemm_bf16_256x256_kernel(GemmBatch batch) {
    __shared__ __attribute__((aligned(16))) uint8_t smem[2 * STAGE2_BYTES];  // 128 KB, the ONLY LDS object

    const int bid = blockIdx.x;
    const int pi = (batch.nprob > 1 && bid >= batch.tile_start1) ? 1 : 0;
    const GemmProblem& p = batch.p[pi];
    const int local = bid - (pi ? batch.tile_start1 : 0);
    constexpr int BMT = 32 * NI;     // tile rows
    constexpr int WROWS = 16 * NI;   // rows of one wave-row
    constexpr int NI1 = NI - 4;      // row blocks of the second half-tile
    const int tiles_m = (p.M + BMT - 1) / BMT, tiles_n = (p.N + BN2 - 1) / BN2;
    int tm, tn;
    tile_coords256(local, tiles_m * tiles_n, tiles_m, tiles_n, tm, tn);
    const int m0 = tm * BMT, n0 = tn * BN2;

    const int t = threadIdx.x;
    const int lane = t & 63, wid = t >> 6;
    const int wm = wid >> 2, wn = wid & 3;
    // FLUX2MI_GEMM_TRACE (tools/gemm_trace.py): per-workgroup s_memtime stamps of the production kernel —
    // entry, first K-tile ready, K loop done, epilogue stores issued, stores retired — plus the hardware id
    unsigned long long tr0 = 0, tr1 = 0, tr2 = 0, tr3 = 0, tr4 = 0;
#define G2_TR(v_) if (batch.trace) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v_)::"memory"); }
    G2_TR(tr0)

    // ---- LDS-DMA source addresses: per kind, 2 instructions per thread.
    // instruction j of wave w fills slot bytes [(j*8+w)*1024, +1024): rows (j*8+w)*8 .. +8,
    // lane -> row + (lane>>3), LDS chunk c' = lane&7 holds global chunk c' ^ ((row>>1)&7).
    constexpr int ES = FP8 ? 1 : 2;          // bytes per element
    constexpr int TILE_K = 128 / ES;         // k per K-tile (a tile row is always 128 B)
    const uint8_t* src[4][2];
    const uint8_t* gA = reinterpret_cast<const uint8_t*>(p.A);
    const uint8_t* gW = reinterpret_cast<const uint8_t*>(p.W);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int rho = (j * 8 + wid) * 8 + (lane >> 3);             // slot row 0..127
        const int chunk = (lane & 7) ^ ((rho >> 1) & 7);
        const int a_row0 = (rho >> 6) * WROWS + (rho & 63);          // A0: wave-row's first 64 rows
        const int b_row0 = (rho >> 5) * 64 + (rho & 31);             // B0: wave-col's first 32 rows
        src[KIND_A0][j] = gA + (size_t)min(m0 + a_row0, p.M - 1) * p.lda * ES + chunk * 16;
        src[KIND_A1][j] = gA + (size_t)min(m0 + a_row0 + 64, p.M - 1) * p.lda * ES + chunk * 16;
        src[KIND_B0][j] = gW + (size_t)min(n0 + b_row0, p.N - 1) * p.ldw * ES + chunk * 16;
        src[KIND_B1][j] = gW + (size_t)min(n0 + b_row0 + 32, p.N - 1) * p.ldw * ES + chunk * 16;
    }
    // LDS-DMA in the scalar-base + 32-bit-vector-offset form (inline asm: the builtin only takes per-lane 64-bit pointers): eight
    // offset registers instead of sixteen pointer registers, and no 64-bit vector add + v_readfirstlane pair in front of every
    // piece.  bf16: +3-4 % on every production shape (profiles/r02_d_gemm_asm_dma_ab.log, two interleaved rounds on one box).
    // FP8: the pointers did not fit next to the 8-register operand tuples of the block-scaled MFMA at all — hipcc had spilled
    // them and reloaded each one with scratch_load + s_waitcnt vmcnt(0) in front of its DMA, DRAINING the pipeline in every
    // phase (196 B/lane of scratch since round 1).  The loads are invisible to hipcc's counters, which is what this loop wants
    // anyway: every wait on them is one of the hand-counted s_waitcnt vmcnt below.
    constexpr bool ASM_DMA = !(VAR & 8);   // VAR bit 3 (dev A/B, FLUX2MI_GEMM_VAR=11): the builtin with per-lane 64-bit pointers instead
    uint32_t soff[4][2];
    if (ASM_DMA) {
#pragma unroll
        for (int kd = 0; kd < 4; ++kd)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                soff[kd][j] = (uint32_t)(src[kd][j] - ((kd == KIND_A0 || kd == KIND_A1) ? gA : gW));
    }
    const uint32_t lds_wave_s = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem + wid * 1024);
    // (M0 is not saved / restored: nothing else in this kernel uses it — gfx950 LDS instructions do not read M0, and the .s of every
    // instantiation has no other reference to it; tools/check_m0.sh greps for that after a compiler change)
#define G2_DMA_ASM(voff_, sbase_, m0_)                                                                                  \
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2" ::"v"(voff_), "s"(m0_), "s"(sbase_) : "memory");
    const long long ksA = p.a_kstride ? p.a_kstride : 128, ksB = p.w_kstride ? p.w_kstride : 128;
    const int lds_wave = wid * 1024;  // + j*8192 + slot*SLOT_BYTES + stage*STAGE2_BYTES
    const int nk = p.K / TILE_K;
    const int total_elems = nk * 4;

    // Shorter tiles (NI < 8): the second half-tile of a wave-row has NI1*16 < 64 rows, i.e. slot rows [0, NI1*16) and
    // [64, 64 + NI1*16): the 8-row groups of waves >= 2*NI1 (both of their instructions) are never read, so those waves
    // skip their A1 loads.  Every wave still has exactly A0, B0, B1 of K-tile kt+2 as its 6 youngest loads at the
    // counted wait, so vmcnt(6) means the same thing for all of them.
    const bool a1_skip = NI1 < 4 && __builtin_amdgcn_readfirstlane(wid) >= 2 * NI1 && !(batch.flags & 2);   // FLUX2MI_GEMM_FLAGS=3: load them anyway (A/B runs)
#define G2_ISSUE(kind_, u_)                                                                                   \
    {                                                                                                         \
        const int u__ = (u_);                                                                                 \
        if (u__ < nk && !((kind_) == KIND_A1 && a1_skip)) {                                                   \
            uint8_t* dst__ = smem + (u__ & 1) * STAGE2_BYTES + (kind_) * SLOT_BYTES + lds_wave;               \
            const long long ko__ = u__ * (((kind_) == KIND_A0 || (kind_) == KIND_A1) ? ksA : ksB);            \
            if (ASM_DMA) {                                                                                    \
                const uint8_t* b__ = (((kind_) == KIND_A0 || (kind_) == KIND_A1) ? gA : gW) + ko__;           \
                const uint32_t m__ = lds_wave_s + (uint32_t)((u__ & 1) * STAGE2_BYTES + (kind_) * SLOT_BYTES); \
                G2_DMA_ASM(soff[kind_][0], b__, m__)                                                          \
                G2_DMA_ASM(soff[kind_][1], b__, m__ + 8192)                                                   \
            } else {                                                                                          \
            __builtin_amdgcn_global_load_lds((gptr_t)(src[kind_][0] + ko__), (lptr_t)(dst__), 16, 0, 0);      \
            __builtin_amdgcn_global_load_lds((gptr_t)(src[kind_][1] + ko__), (lptr_t)(dst__ + 8192), 16, 0, 0); \
            }                                                                                                 \
        }                                                                                                     \
    }

    f32x4 acc[NI][4];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- fragment read offsets inside a slot
    const int frow = lane & 15, fq = lane >> 4;
    const int sw = (frow >> 1) & 7;
    const int a_frag = (wm * 64 + frow) * 128;   // + i*16*128
    const int b_frag = (wn * 32 + frow) * 128;   // + j*16*128
    const int c0 = ((0 + fq) ^ sw) << 4, c1 = ((4 + fq) ^ sw) << 4;   // kk = 0, 1

    // PIPE (VAR bit 2, bf16): every phase's MFMAs run on fragments read during the PREVIOUS phase, so no LDS read latency sits
    // between a phase's barrier and its MFMAs; the rows of the second half-tile then need their own registers (a1f).
    constexpr bool PIPE = (VAR & 4) && !FP8;
    bf16x8 a[4][2], a1f[PIPE ? 4 : 1][2], b0[2][2], b1[2][2];
    // FP8: a lane's two 16-byte chunks are ONE 32-byte operand of the block-scaled MFMA.  The 8-register tuples are formed where
    // the fragments are READ (the two halves die at once and coalesce into the tuple); forming them at the MFMA kept both the
    // halves and the tuples live and cost 50 spilled registers inside the K loop (196 B/lane of scratch, scratch loads on the
    // vmcnt counter of the LDS-DMA pipeline).
    i32x8 a8[4], b08[2], b18[2];

#define G2_READ_A(slot_)                                                                   \
    _Pragma("unroll") for (int i_ = 0; i_ < ((slot_) == KIND_A1 ? NI1 : 4); ++i_) {        \
        const uint8_t* r__ = st + (slot_) * SLOT_BYTES + a_frag + i_ * 2048;               \
        if (FP8) {                                                                         \
            a8[i_] = __builtin_shufflevector(*reinterpret_cast<const i32x4*>(r__ + c0), *reinterpret_cast<const i32x4*>(r__ + c1), 0, 1, 2, 3, 4, 5, 6, 7); \
        } else if (PIPE && (slot_) == KIND_A1) {                                           \
            a1f[PIPE ? i_ : 0][0] = *reinterpret_cast<const bf16x8*>(r__ + c0);            \
            a1f[PIPE ? i_ : 0][1] = *reinterpret_cast<const bf16x8*>(r__ + c1);            \
        } else {                                                                           \
            a[i_][0] = *reinterpret_cast<const bf16x8*>(r__ + c0);                         \
            a[i_][1] = *reinterpret_cast<const bf16x8*>(r__ + c1);                         \
        }                                                                                  \
    }
#define G2_READ_B(dst_, slot_)                                                             \
    _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) {                                     \
        const uint8_t* r__ = st + (slot_) * SLOT_BYTES + b_frag + j_ * 2048;               \
        if (FP8) {                                                                         \
            dst_##8[j_] = __builtin_shufflevector(*reinterpret_cast<const i32x4*>(r__ + c0), *reinterpret_cast<const i32x4*>(r__ + c1), 0, 1, 2, 3, 4, 5, 6, 7); \
        } else {                                                                           \
            dst_[j_][0] = *reinterpret_cast<const bf16x8*>(r__ + c0);                      \
            dst_[j_][1] = *reinterpret_cast<const bf16x8*>(r__ + c1);                      \
        }                                                                                  \
    }
#define G2_MFMA(mh_, nh_, bsrc_)                                                                         \
    __builtin_amdgcn_s_setprio(1);                                                                       \
    if (!(ABL & 16)) {                                                                                   \
        if (FP8) {                                                                                       \
            /* one block-scaled 16x16x128 per (i, j): the lane's 32 bytes are its two 16-B chunks;  */   \
            /* scales are e8m0 127 = 1.0 (the real scales are per row/column, applied in the epilogue) */ \
            _Pragma("unroll") for (int i_ = 0; i_ < ((mh_) ? NI1 : 4); ++i_) {                                           \
                _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) {                                       \
                    acc[(mh_) * 4 + i_][(nh_) * 2 + j_] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4( \
                        bsrc_##8[j_], a8[i_], acc[(mh_) * 4 + i_][(nh_) * 2 + j_], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F); \
                }                                                                                        \
            }                                                                                            \
        } else {                                                                                         \
            _Pragma("unroll") for (int kk_ = 0; kk_ < 2; ++kk_)                                          \
                _Pragma("unroll") for (int i_ = 0; i_ < ((mh_) ? NI1 : 4); ++i_)                                         \
                    _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                     \
                        acc[(mh_) * 4 + i_][(nh_) * 2 + j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(  \
                            bsrc_[j_][kk_], (PIPE && (mh_)) ? a1f[PIPE ? i_ : 0][kk_] : a[i_][kk_],      \
                            acc[(mh_) * 4 + i_][(nh_) * 2 + j_], 0, 0, 0);                               \
        }                                                                                                \
    }                                                                                                    \
    __builtin_amdgcn_s_setprio(0);
    // DIAG: timing-only build (FLUX2MI_GEMM_DIAG=1): s_memtime stamps around the four segments of a
    // phase, summed per wave and written to p.C2 (a debug buffer no kernel reads); never shipped.
    unsigned long long dg_t = 0, dg_r = 0, dg_w1 = 0, dg_m = 0, dg_w2 = 0, dg_dma = 0, dg_rd = 0;
#define G2_STAMP(acc_)                                                                          \
    if (DIAG) {                                                                                 \
        unsigned long long n__;                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                      \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(n__)::"memory");             \
        __builtin_amdgcn_sched_barrier(0);                                                      \
        acc_ += n__ - dg_t;                                                                     \
        dg_t = n__;                                                                             \
    }
#define G2_STAMP_NW(acc_)                                                                       \
    if (DIAG) {                                                                                 \
        unsigned long long n__;                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                      \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(n__)::"memory");             \
        __builtin_amdgcn_sched_barrier(0);                                                      \
        acc_ += n__ - dg_t;                                                                     \
        dg_t = n__;                                                                             \
    }
#define G2_SYNC()                        \
    asm volatile("" ::: "memory");       \
    __builtin_amdgcn_s_barrier();        \
    asm volatile("" ::: "memory");
#define G2_READS_DONE()                                          \
    if (STAGGER) {                                               \
        if (!(VAR & 1)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        G2_STAMP(dg_r)                                           \
        G2_SYNC()                                                \
        if ((VAR & 1) && !(VAR & 16)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
        G2_STAMP(dg_w1)                                          \
    }
#define G2_PHASE_END()                                                          \
    if ((ABL & 8) || (VAR & 2)) {}                                              \
    else if (drain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            \
    else if (STAGGER) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");          \
    else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");                      \
    G2_STAMP(dg_m)                                                              \
    G2_SYNC()                                                                   \
    G2_STAMP(dg_w2)
#define G2_TILE_WAIT()                                                          \
    if (VAR & 2) {                                                              \
        if (drain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             \
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                   \
    }

    const int wm_u = __builtin_amdgcn_readfirstlane(wid) >> 2;  // provably wave-uniform
#define G2_ISSUE_L(k_, u_) if (ABL == 4) G2_ISSUE(k_, ((u_) < nk ? ((u_) & 1) : nk)) else if (!(ABL & 2)) G2_ISSUE(k_, u_)
#define G2_READ_A_L(s_) if (!(ABL & 1)) { G2_READ_A(s_) }
#define G2_READ_B_L(d_, s_) if (!(ABL & 1)) { G2_READ_B(d_, s_) }
    if (HALFBAR) {
        // Variant with TWO barriers per K-tile (after phases 1 and 3), lockstep waves.  A slot may
        // only be refilled after a barrier that follows its last read, so phase phi issues element
        // phi + 6:  p0 -> B1(t+1), p1 -> A1(t+1), p2 -> A0(t+2), p3 -> B0(t+2); vmcnt(6) before each
        // barrier leaves the 3 youngest half-tiles in flight.
        G2_ISSUE(KIND_A0, 0) G2_ISSUE(KIND_B0, 0) G2_ISSUE(KIND_B1, 0) G2_ISSUE(KIND_A1, 0)
        G2_ISSUE(KIND_A0, 1) G2_ISSUE(KIND_B0, 1)
        if (nk < 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        G2_SYNC()
        for (int kt = 0; kt < nk; ++kt) {
            const uint8_t* st = smem + (kt & 1) * STAGE2_BYTES;
            const bool drain = kt + 2 >= nk;
            G2_READ_B_L(b0, KIND_B0)
            G2_READ_A_L(KIND_A0)
            G2_READ_B_L(b1, KIND_B1)
            G2_ISSUE_L(KIND_B1, kt + 1)
            G2_MFMA(0, 0, b0)
            G2_ISSUE_L(KIND_A1, kt + 1)
            G2_MFMA(0, 1, b1)
            if (drain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            G2_SYNC()
            G2_READ_A_L(KIND_A1)
            G2_ISSUE_L(KIND_A0, kt + 2)
            G2_MFMA(1, 1, b1)
            G2_ISSUE_L(KIND_B0, kt + 2)
            G2_MFMA(1, 0, b0)
            if (drain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            G2_SYNC()
        }
    } else {
    // ---- prologue: elements 0..6 = K-tile 0 (A0,B0,B1,A1) + K-tile 1 (A0,B0,B1)
    G2_ISSUE(KIND_A0, 0) G2_ISSUE(KIND_B0, 0) G2_ISSUE(KIND_B1, 0) G2_ISSUE(KIND_A1, 0)
    G2_ISSUE(KIND_A0, 1) G2_ISSUE(KIND_B0, 1) G2_ISSUE(KIND_B1, 1)
    {
        const bool drain = nk < 3;
        G2_TILE_WAIT()
        G2_PHASE_END()
    }
    if (STAGGER && wm_u == 1) { G2_SYNC() }
    G2_TR(tr1)

    if (ABL & 1) {  // operands read once; wrong results by design
        const uint8_t* st = smem;
        G2_READ_B(b0, KIND_B0) G2_READ_B(b1, KIND_B1) G2_READ_A(KIND_A0)
    }
    if (DIAG) { unsigned long long z__ = 0; G2_STAMP(z__) dg_r = dg_w1 = dg_m = dg_w2 = dg_dma = dg_rd = z__ * 0; }
    if (PIPE) {
        // Software-pipelined fragment reads (bf16).  The whole K-tile kt+1 is visible to every wave from phase 3 of K-tile kt on:
        // the K-tile's counted wait sits in phase 2 (vmcnt(4): only A0, B0 of kt+2 may still fly), so that BOTH wave groups —
        // which run half a phase apart — have executed it before either passes the barrier that opens its phase 3.
        //   p0: read B1(kt) | issue A1(kt+1) | MFMA (m0,n0) on a, b0      p2: issue B0(kt+2), vmcnt(4) | MFMA (m1,n0) on a1f, b0
        //   p1: read A1(kt) | issue A0(kt+2) | MFMA (m0,n1) on a, b1      p3: issue B1(kt+2) | read A0, B0(kt+1) | MFMA (m1,n1) on a1f, b1
        // hipcc places the lgkmcnt waits itself (counted, in front of the first MFMA that consumes a fragment).
        {
            const uint8_t* st = smem;
            G2_READ_B(b0, KIND_B0)
            G2_READ_A(KIND_A0)
        }
        for (int kt = 0; kt < nk; ++kt) {
            const uint8_t* st = smem + (kt & 1) * STAGE2_BYTES;
            const bool drain = kt + 2 >= nk;
            G2_READ_B(b1, KIND_B1)
            G2_ISSUE(KIND_A1, kt + 1)
            G2_SYNC()
            G2_MFMA(0, 0, b0)
            G2_SYNC()
            G2_READ_A(KIND_A1)
            G2_ISSUE(KIND_A0, kt + 2)
            G2_SYNC()
            G2_MFMA(0, 1, b1)
            G2_SYNC()
            G2_ISSUE(KIND_B0, kt + 2)
            if (drain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            G2_SYNC()
            G2_MFMA(1, 0, b0)
            G2_SYNC()
            G2_ISSUE(KIND_B1, kt + 2)
            G2_SYNC()
            if (kt + 1 < nk) {
                st = smem + ((kt + 1) & 1) * STAGE2_BYTES;
                G2_READ_B(b0, KIND_B0)
                G2_READ_A(KIND_A0)
            }
            G2_MFMA(1, 1, b1)
            G2_SYNC()
        }
    } else
    for (int kt = 0; kt < nk; ++kt) {
        const uint8_t* st = smem + (kt & 1) * STAGE2_BYTES;
        const bool drain = kt + 2 >= nk;
        // phase 0: issue A1(kt+1); consume A0, B0 -> quadrant (m0, n0)
        G2_READ_B_L(b0, KIND_B0)      // fragment reads first: their LDS latency runs under the DMA queue wait
        G2_READ_A_L(KIND_A0)
        G2_ISSUE_L(KIND_A1, kt + 1)
        G2_STAMP_NW(dg_dma)
        G2_READS_DONE()
        G2_MFMA(0, 0, b0)
        G2_PHASE_END()
        // phase 1: issue A0(kt+2); consume B1 -> (m0, n1)
        G2_READ_B_L(b1, KIND_B1)
        G2_ISSUE_L(KIND_A0, kt + 2)
        G2_STAMP_NW(dg_dma)
        G2_READS_DONE()
        G2_MFMA(0, 1, b1)
        G2_PHASE_END()
        // phase 2: issue B0(kt+2); consume A1 -> (m1, n1)
        G2_READ_A_L(KIND_A1)
        G2_ISSUE_L(KIND_B0, kt + 2)
        G2_STAMP_NW(dg_dma)
        G2_READS_DONE()
        G2_MFMA(1, 1, b1)
        G2_PHASE_END()
        // phase 3: issue B1(kt+2); b0 still in registers -> (m1, n0)
        G2_ISSUE_L(KIND_B1, kt + 2)
        G2_STAMP_NW(dg_dma)
        G2_TILE_WAIT()
        G2_READS_DONE()
        G2_MFMA(1, 0, b0)
        G2_PHASE_END()
    }
    }
    if (STAGGER && wm_u == 0) { G2_SYNC() }
    G2_TR(tr2)
    if (DIAG && p.C2 != nullptr && lane == 0) {
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(p.C2) + ((size_t)blockIdx.x * 8 + wid) * 4;
        dbg[0] = dg_r; dbg[1] = dg_w1; dbg[2] = dg_m; dbg[3] = dg_w2 + (dg_dma << 32);
    }
#undef G2_ISSUE
#undef G2_READ_A
#undef G2_READ_B
#undef G2_MFMA
#undef G2_PHASE_END
#undef G2_TILE_WAIT
#undef G2_READS_DONE
#undef G2_SYNC
#undef G2_STAMP

    // ---- epilogue.  The MFMAs take the W fragment as the A operand and the activation fragment as
    // B, so the accumulator tile is the TRANSPOSE of the output block: lane (frow, fq) holds, for
    // output row m0 + wm*128 + i*16 + frow, the FOUR CONSECUTIVE columns n0 + wn*64 + j*16 + fq*4 + r.
    // Every store is then 16 B (f32) or 8 B (bf16) per lane instead of 4 / 2 B: a quarter of the
    // memory instructions (the scalar form spent ~17k cycles per tile in the store queue).
    const int row_base = m0 + wm * WROWS + frow;
    const int col_base = n0 + wn * 64 + fq * 4;
    if ((ABL & 32) && p.alpha != 12345.f) return;   // timing only: the K loop without its epilogue
    int epi = p.epi;
    if (epi == EPI_QKV_MLP) epi = (n0 >= p.n_split) ? EPI_SWIGLU : EPI_BF16;
    if (FP8) {  // de-quantise: per-row activation scale x per-column weight scale.  One column block at a time (4 + 1 scale
                // registers live next to the 128 accumulators, not 16 + 8: the difference was spilled INTO THE K LOOP, see below)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float w4[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) w4[r] = p.w_scale[min(col_base + j * 16 + r, p.N - 1)];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const float as = p.a_scale[min(row_base + i * 16, p.M - 1)];
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] *= as * w4[r];
            }
        }
    }

    if (p.qk_qw != nullptr && epi == EPI_BF16 && n0 < p.qk_cols) {
        // ---- fused QK RMSNorm + RoPE (Flux2Attention.swift:137-157).  A 256-column tile is two heads: waves wn 0,1
        // hold head A's 128 columns, wn 2,3 head B's.  Sum of squares: 16 values in the lane, the 4 fq lanes of a row
        // by two lane exchanges, the partner wave through LDS (free now; one barrier on each side).
        float2 ps[NI];                                  // positions of the wave's two RoPE axes, per row block
#pragma unroll
        for (int i = 0; i < NI; ++i)
            ps[i] = *reinterpret_cast<const float2*>(p.qk_pos + (size_t)(p.qk_row0 + min(row_base + i * 16, p.M - 1)) * 4 + (wn & 1) * 2);
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);   // [8 waves][128 rows]
        float ss[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            float t2 = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) t2 += acc[i][j][r] * acc[i][j][r];
            t2 += __shfl_xor(t2, 16);
            t2 += __shfl_xor(t2, 32);
            ss[i] = t2;
        }
        if (fq == 0) {
#pragma unroll
            for (int i = 0; i < NI; ++i) red[wid * 128 + i * 16 + frow] = ss[i];
        }
        __syncthreads();
        const bool is_k = n0 >= (p.qk_cols >> 1);
        const uint16_t* wv = (is_k ? p.qk_kw : p.qk_qw) + (wn & 1) * 64 + fq * 4;
        const float qm = is_k ? 1.0f : p.qk_qmul;
        float wj[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint2 u = *reinterpret_cast<const uint2*>(wv + j * 16);
            wj[j][0] = lo_bf16(u.x) * qm; wj[j][1] = hi_bf16(u.x) * qm; wj[j][2] = lo_bf16(u.y) * qm; wj[j][3] = hi_bf16(u.y) * qm;
        }
        // the lane's columns inside the head: (wn&1)*64 + j*16 + fq*4 + r = axis (wn&1)*2 + (j>>1), pair (j&1)*8 + fq*2 + (r>>1)
        // of that axis: four frequencies per lane, in revolutions (v_sin_f32 / v_cos_f32 take x / 2pi)
        float wrev[2][2];
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
            for (int pr = 0; pr < 2; ++pr)
                wrev[jb][pr] = __builtin_amdgcn_exp2f(-(float)(jb * 8 + fq * 2 + pr) * p.qk_l2t16) * 0.15915494309189535f;
        uint16_t* C = reinterpret_cast<uint16_t*>(p.C);
        const int wcol = n0 + wn * 64 + (fq & 1) * 16 + (fq >> 1) * 8;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const float tot = ss[i] + red[(wid ^ 1) * 128 + i * 16 + frow];
            const float rinv = rsqrtf(tot * (1.0f / 128.0f) + 1e-6f);
            const int row = row_base + i * 16;
            uint32_t pk[4][2];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float pos = (j >> 1) ? ps[i].y : ps[i].x;
                const float f0 = __builtin_amdgcn_fractf(pos * wrev[j & 1][0]), f1 = __builtin_amdgcn_fractf(pos * wrev[j & 1][1]);
                const float c0 = __builtin_amdgcn_cosf(f0), s0 = __builtin_amdgcn_sinf(f0);
                const float c1 = __builtin_amdgcn_cosf(f1), s1 = __builtin_amdgcn_sinf(f1);
                const float a0 = acc[i][j][0] * rinv * wj[j][0], a1 = acc[i][j][1] * rinv * wj[j][1];
                const float a2 = acc[i][j][2] * rinv * wj[j][2], a3 = acc[i][j][3] * rinv * wj[j][3];
                pk[j][0] = pack_bf16x2(a0 * c0 - a1 * s0, a1 * c0 + a0 * s0);
                pk[j][1] = pack_bf16x2(a2 * c1 - a3 * s1, a3 * c1 + a2 * s1);
            }
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                swap16(pk[2 * jp][0], pk[2 * jp + 1][0]);
                swap16(pk[2 * jp][1], pk[2 * jp + 1][1]);
                if (row < p.M)
                    *reinterpret_cast<uint4*>(C + (size_t)row * p.ldc + wcol + jp * 32) =
                        make_uint4(pk[2 * jp][0], pk[2 * jp][1], pk[2 * jp + 1][0], pk[2 * jp + 1][1]);
            }
        }
    } else
    if (epi == EPI_F32) {
        float* C = reinterpret_cast<float*>(p.C);
        const bool vec_ok = (p.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int row = row_base + i * 16;
            if (row < p.M) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
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
        uint16_t* C = reinterpret_cast<uint16_t*>(p.C);
        const bool vec_ok = (p.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(C) & 7) == 0;
        if ((batch.flags & 1) && (p.ldc & 7) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0 && n0 + BN2 <= p.N) {
            // 16-byte stores: half the store instructions of the 8-byte form below.  col_perm16: the four-column
            // groups of a 16-group go fq 0,2,1,3, so the 16-byte runs pair fq with fq^2 (lanes l, l^32): after the
            // exchange lanes fq 0,1 hold block 2jp at (fq&1)*8 and lanes fq 2,3 block 2jp+1
            const int wcol = p.col_perm16 ? n0 + wn * 64 + (fq >> 1) * 16 + (fq & 1) * 8
                                          : n0 + wn * 64 + (fq & 1) * 16 + (fq >> 1) * 8;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int row = row_base + i * 16;
#pragma unroll
                for (int jp = 0; jp < 2; ++jp) {
                    const f32x4 vx = acc[i][2 * jp] * p.alpha, vy = acc[i][2 * jp + 1] * p.alpha;
                    uint32_t x0 = pack_bf16x2(vx[0], vx[1]), x1 = pack_bf16x2(vx[2], vx[3]);
                    uint32_t y0 = pack_bf16x2(vy[0], vy[1]), y1 = pack_bf16x2(vy[2], vy[3]);
                    if (p.col_perm16) {
                        swap32(x0, y0);
                        swap32(x1, y1);
                    } else {
                        swap16(x0, y0);
                        swap16(x1, y1);
                    }
                    if (ABL & 64) {   // timing only: the same bytes as whole 128-B rows (8 full lines per instruction)
                        const int fr = m0 + wm * WROWS + i * 16 + jp * 8 + (lane >> 3);
                        if (fr < p.M) *reinterpret_cast<uint4*>(C + (size_t)fr * p.ldc + n0 + wn * 64 + (lane & 7) * 8) = make_uint4(x0, x1, y0, y1);
                    } else
                    if (row < p.M)
                        *reinterpret_cast<uint4*>(C + (size_t)row * p.ldc + wcol + jp * 32) = make_uint4(x0, x1, y0, y1);
                }
            }
        } else
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int row = row_base + i * 16;
            if (row < p.M) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    // col = the output column (bounds), scol = where it is stored (col_perm16: a valid key of a
                    // partial last group may land past N, inside the padded row)
                    const int col = col_base + j * 16;
                    const int scol = p.col_perm16 ? n0 + wn * 64 + j * 16 + ((fq & 1) * 2 + (fq >> 1)) * 4 : col;
                    uint16_t* dst = C + (size_t)row * p.ldc + scol;
                    const f32x4 v = acc[i][j] * p.alpha;
                    if (vec_ok && col + 3 < p.N) {
                        *reinterpret_cast<uint2*>(dst) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (col + r < p.N) dst[r] = f32_to_bf16_bits(v[r]);
                    }
                }
            }
        }
    } else if (epi == EPI_GATE_RESID) {
        float* C = reinterpret_cast<float*>(p.C);
        const bool vec_ok = (p.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0;
        f32x4 g[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int col = col_base + j * 16 + r;
                g[j][r] = (col < p.N) ? p.vec[col] : 0.f;
            }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int row = row_base + i * 16;
            if (row < p.M) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
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
    } else if (epi == EPI_SWIGLU) {
        // weight rows are interleaved in blocks of 32 (gate | value) inside each 64-column group =
        // one wave's columns: gate = accumulator blocks j 0,1, value = blocks 2,3 of the same lane
        uint16_t* C = reinterpret_cast<uint16_t*>(p.C2);
        const int start = (p.epi == EPI_QKV_MLP) ? p.n_split : 0;
        const int hbase = ((n0 + wn * 64 - start) >> 1) + fq * 4;
        const int hidden = (p.N - start) >> 1;
        const bool vec_ok = (p.ldc2 & 3) == 0 && (reinterpret_cast<uintptr_t>(C) & 7) == 0;
        if ((batch.flags & 1) && (p.ldc2 & 7) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0 && n0 + BN2 <= p.N) {
            const int wh = ((n0 + wn * 64 - start) >> 1) + (fq & 1) * 16 + (fq >> 1) * 8;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int row = row_base + i * 16;
                float ox[4], oy[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    ox[r] = silu_f(acc[i][0][r]) * acc[i][2][r];
                    oy[r] = silu_f(acc[i][1][r]) * acc[i][3][r];
                }
                uint32_t x0 = pack_bf16x2(ox[0], ox[1]), x1 = pack_bf16x2(ox[2], ox[3]);
                uint32_t y0 = pack_bf16x2(oy[0], oy[1]), y1 = pack_bf16x2(oy[2], oy[3]);
                swap16(x0, y0);
                swap16(x1, y1);
                if (ABL & 64) {
                    const int fr = m0 + wm * WROWS + (wn & 1) * 64 + i * 8 + (lane >> 3);
                    const int fh = ((n0 - start) >> 1) + (wn >> 1) * 64 + (lane & 7) * 8;
                    if (fr < p.M) *reinterpret_cast<uint4*>(C + (size_t)fr * p.ldc2 + fh) = make_uint4(x0, x1, y0, y1);
                } else
                if (row < p.M) *reinterpret_cast<uint4*>(C + (size_t)row * p.ldc2 + wh) = make_uint4(x0, x1, y0, y1);
            }
        } else
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int row = row_base + i * 16;
            if (row < p.M) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int h = hbase + j * 16;
                    uint16_t* dst = C + (size_t)row * p.ldc2 + h;
                    float o[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = silu_f(acc[i][j][r]) * acc[i][j + 2][r];
                    if (vec_ok && h + 3 < hidden) {
                        *reinterpret_cast<uint2*>(dst) = make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (h + r < hidden) dst[r] = f32_to_bf16_bits(o[r]);
                    }
                }
            }
        }
    } else {  // EPI_BIAS_BF16
        uint16_t* C = reinterpret_cast<uint16_t*>(p.C);
        const bool vec_ok = (p.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(C) & 7) == 0 &&
                            (p.resid == nullptr || ((p.ldr & 3) == 0 && (reinterpret_cast<uintptr_t>(p.resid) & 7) == 0));
        f32x4 bsv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int col = col_base + j * 16 + r;
                bsv[j][r] = (p.vec != nullptr && col < p.N) ? p.vec[col] : 0.f;
            }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int row = row_base + i * 16;
            if (row < p.M) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int col = col_base + j * 16;
                    uint16_t* dst = C + (size_t)row * p.ldc + col;
                    f32x4 v = acc[i][j] * p.alpha + bsv[j];
                    if (vec_ok && col + 3 < p.N) {
                        if (p.resid) {
                            const uint2 rv = *reinterpret_cast<const uint2*>(p.resid + (size_t)row * p.ldr + col);
                            v[0] += bf16_bits_to_f32((uint16_t)(rv.x & 0xffff)); v[1] += bf16_bits_to_f32((uint16_t)(rv.x >> 16));
                            v[2] += bf16_bits_to_f32((uint16_t)(rv.y & 0xffff)); v[3] += bf16_bits_to_f32((uint16_t)(rv.y >> 16));
                        }
                        *reinterpret_cast<uint2*>(dst) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (col + r < p.N) {
                                float x = v[r];
                                if (p.resid) x += bf16_bits_to_f32(p.resid[(size_t)row * p.ldr + col + r]);
                                dst[r] = f32_to_bf16_bits(x);
                            }
                    }
                }
            }
        }
    }
    if (batch.trace) {
        G2_TR(tr3)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        G2_TR(tr4)
        if (lane == 0) {
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            unsigned long long* o = batch.trace + ((size_t)bid * 8 + wid) * 8;
            o[0] = tr0; o[1] = tr1; o[2] = tr2; o[3] = tr3; o[4] = tr4;
            o[5] = (unsigned long long)hw | ((unsigned long long)xcc << 32); o[6] = (unsigned long long)bid; o[7] = (unsigned long long)p.epi;
        }
    }
#undef G2_TR
}

}  // namespace

int gemm256_pick_ni(const GemmProblem* probs, int nprob);   // gemm256s.hip

// returns 0 ok / 2 launch error; the caller has validated the problems
int launch_gemm256(hipStream_t stream, const GemmProblem* probs, int nprob) {
    GemmBatch b;
    b.nprob = nprob;
    auto count_tiles = [&](int ni, int* per) {
        int tot = 0;
        for (int i = 0; i < nprob; ++i) {
            const int n = ((probs[i].M + 32 * ni - 1) / (32 * ni)) * ((probs[i].N + BN2 - 1) / BN2);
            if (per) per[i] = n;
            tot += n;
        }
        return tot;
    };
    // the ablation / diagnostic / schedule-variant instantiations only exist for the 256-row tile
    static const bool dev_env = F2_DEV_ENV("FLUX2MI_GEMM_ABLATE") || F2_DEV_ENV("FLUX2MI_GEMM_VAR") || F2_DEV_ENV("FLUX2MI_GEMM_STAGGER") ||
                                F2_DEV_ENV("FLUX2MI_GEMM_DIAG");
    const int ni = dev_env ? 8 : gemm256_pick_ni(probs, nprob);
    int tiles[2] = {0, 0};
    for (int i = 0; i < nprob; ++i) {
        b.p[i] = probs[i];
        const GemmProblem& q = probs[i];   // fused QK norm + RoPE: whole heads per tile, 16-byte row stores
        if (q.qk_qw && (!q.qk_kw || !q.qk_pos || q.qk_cols % 512 != 0 || q.qk_cols > q.N || (q.N & 255) || (q.ldc & 7) ||
                        (reinterpret_cast<uintptr_t>(q.C) & 15) || (q.epi != EPI_BF16 && q.epi != EPI_QKV_MLP) ||
                        (q.epi == EPI_QKV_MLP && q.qk_cols > q.n_split) || dev_env))
            return 1;
    }
    count_tiles(ni, tiles);
    if (nprob == 1) b.p[1] = probs[0];
    b.tile_start1 = tiles[0];
    // bit 0: 16-byte bf16 / SwiGLU epilogue stores (FLUX2MI_GEMM_FLAGS=0 for A/B runs: linear1 0.617 -> 0.604 ms)
    static const int flags = [] {
        const char* e = F2_DEV_ENV("FLUX2MI_GEMM_FLAGS");
        return e ? atoi(e) : 1;
    }();
    b.flags = flags;
    b.trace = nullptr;
    static const char* trace_path = F2_DEV_ENV("FLUX2MI_GEMM_TRACE");
    static int trace_seen = 0;
    const bool trace_now = trace_path && probs[0].epi == EPI_QKV_MLP && !probs[0].fp8 && ++trace_seen == 30;   // a warm linear1
    const size_t trace_n = (size_t)(tiles[0] + tiles[1]) * 8 * 8;
    if (trace_now && hipMalloc((void**)&b.trace, trace_n * 8) != hipSuccess) b.trace = nullptr;

    const bool fp8 = probs[0].fp8 != 0;
#define G2_GO_NI(FP8_, VAR_)                                                                                                  \
    switch (ni) {                                                                                                       \
        case 7: hipLaunchKernelGGL((gemm_bf16_256x256_kernel<true, 0, false, false, FP8_, VAR_, 7>), dim3(tiles[0] + tiles[1]), dim3(512), 0, stream, b); break; \
        case 6: hipLaunchKernelGGL((gemm_bf16_256x256_kernel<true, 0, false, false, FP8_, VAR_, 6>), dim3(tiles[0] + tiles[1]), dim3(512), 0, stream, b); break; \
        case 5: hipLaunchKernelGGL((gemm_bf16_256x256_kernel<true, 0, false, false, FP8_, VAR_, 5>), dim3(tiles[0] + tiles[1]), dim3(512), 0, stream, b); break; \
        default: hipLaunchKernelGGL((gemm_bf16_256x256_kernel<true, 0, false, false, FP8_, VAR_, 8>), dim3(tiles[0] + tiles[1]), dim3(512), 0, stream, b); break; \
    }
    // VAR 19 = one counted vmcnt per K-tile (bit 1), the fragment-read wait behind the phase's first barrier (bit 0) left to
    // hipcc's counted lgkmcnt waits in front of each MFMA pair (bit 4: +1-1.7 % over an explicit lgkmcnt(0), r02_d_gemm_lgkm_ab.log)
    if (fp8) {
        G2_GO_NI(true, 19)
        return hipGetLastError() == hipSuccess ? 0 : 2;
    }
    const dim3 grid(tiles[0] + tiles[1]), block(512);
#ifdef FLUX2MI_DEV   // every schedule variant, timing-only ablation and diagnostic build below exists in libflux2mi_dev.so only
    static const int abl = [] {
        const char* e = F2_DEV_ENV("FLUX2MI_GEMM_ABLATE");
        return e ? atoi(e) : 0;
    }();
    static const int stagger = [] {
        const char* e = F2_DEV_ENV("FLUX2MI_GEMM_STAGGER");
        return e ? atoi(e) : 1;
    }();
    // schedule variant (see VAR on the kernel): default 3 = one counted vmcnt per K-tile (measured +7 %
    // over a wait in every phase: linear1 0.649 -> 0.605 ms, 8192^3 1224 -> 1326 TF on one box) and
    // the fragment-read wait behind the phase's first barrier (+0..1 %)
    static const int var = [] {
        const char* e = F2_DEV_ENV("FLUX2MI_GEMM_VAR");
        return e ? atoi(e) : 19;
    }();
    if (abl == 32 || abl == 64) {
        if (abl == 32) hipLaunchKernelGGL((gemm_bf16_256x256_kernel<true, 32, false, false, false, 3>), grid, block, 0, stream, b);
        else hipLaunchKernelGGL((gemm_bf16_256x256_kernel<true, 64, false, false, false, 3>), grid, block, 0, stream, b);
        return hipGetLastError() == hipSuccess ? 0 : 2;
    }
    if (var == 2 && abl == 0 && stagger == 0) {   // lockstep waves, one wait per K-tile (A/B only)
        hipLaunchKernelGGL((gemm_bf16_256x256_kernel<false, 0, false, false, false, 2>), grid, block, 0, stream, b);
        return hipGetLastError() == hipSuccess ? 0 : 2;
    }
    if (var == 3 && abl == 0 && stagger == 1 && ni == 8) {    // explicit lgkmcnt(0) behind the phase's first barrier (round 1's form)
        hipLaunchKernelGGL((gemm_bf16_256x256_kernel<true, 0, false, false, false, 3, 8>), grid, block, 0, stream, b);
        return hipGetLastError() == hipSuccess ? 0 : 2;
    }
    if (var == 7 && abl == 0 && stagger == 1 && ni == 8) {    // software-pipelined fragment reads (A/B)
        hipLaunchKernelGGL((gemm_bf16_256x256_kernel<true, 0, false, false, false, 7, 8>), grid, block, 0, stream, b);
        return hipGetLastError() == hipSuccess ? 0 : 2;
    }
    if (var == 11 && abl == 0 && stagger == 1 && ni == 8) {   // the builtin LDS-DMA form (round 1's), for A/B
        hipLaunchKernelGGL((gemm_bf16_256x256_kernel<true, 0, false, false, false, 11, 8>), grid, block, 0, stream, b);
        return hipGetLastError() == hipSuccess ? 0 : 2;
    }
    if (!(((var >= 1 && var <= 3) || var == 19) && abl == 0 && stagger == 1)) {
        static const int diag = [] {
            const char* e = F2_DEV_ENV("FLUX2MI_GEMM_DIAG");
            return e ? atoi(e) : 0;
        }();
        if (diag && nprob == 1 && probs[0].epi == EPI_F32) {
            // segment-share diagnostic: per-wave cycle sums for (reads+DMA issue, barrier 1, MFMA+vmcnt, barrier 2)
            static unsigned long long* dbg = nullptr;
            const size_t n = (size_t)grid.x * 8 * 4;
            if (!dbg) (void)hipMalloc((void**)&dbg, 4096 * 8 * 4 * 8);
            if (dbg && grid.x <= 4096) {
                b.p[0].C2 = dbg;
                (void)hipMemsetAsync(dbg, 0, n * 8, stream);
                if (abl == 8) hipLaunchKernelGGL((gemm_bf16_256x256_kernel<true, 8, false, true>), grid, block, 0, stream, b);
                else hipLaunchKernelGGL((gemm_bf16_256x256_kernel<true, 0, false, true>), grid, block, 0, stream, b);
                std::vector<unsigned long long> h(n);
                (void)hipMemcpyAsync(h.data(), dbg, n * 8, hipMemcpyDeviceToHost, stream);
                (void)hipStreamSynchronize(stream);
                double s[2][4] = {{0}};
                for (size_t w = 0; w < (size_t)grid.x * 8; ++w)
                    for (int k = 0; k < 4; ++k) s[(w & 7) >> 2][k] += (double)(k == 3 ? (h[w * 4 + k] & 0xffffffffull) : h[w * 4 + k]);
                double dma[2] = {0, 0};
                for (size_t w = 0; w < (size_t)grid.x * 8; ++w) dma[(w & 7) >> 2] += (double)(h[w * 4 + 3] >> 32);
                const double nw = (double)grid.x * 4;
                const int nkt = probs[0].K / 64;
                for (int g = 0; g < 2; ++g)
                    fprintf(stderr, "[gemm diag] group %d per K-tile cycles: dma-issue %.0f  reads+lgkm-wait %.0f  barrier1 %.0f  mfma+vmcnt %.0f  barrier2 %.0f  (total %.0f)\n", g,
                            dma[g] / nw / nkt, s[g][0] / nw / nkt, s[g][1] / nw / nkt, s[g][2] / nw / nkt, s[g][3] / nw / nkt,
                            (s[g][0] + s[g][1] + s[g][2] + s[g][3]) / nw / nkt);
                return 0;
            }
        }
        if (abl == 1) hipLaunchKernelGGL((gemm_bf16_256x256_kernel<true, 1>), grid, block, 0, stream, b);
        else if (abl == 2) hipLaunchKernelGGL((gemm_bf16_256x256_kernel<true, 2>), grid, block, 0, stream, b);
        else if (abl == 3) hipLaunchKernelGGL((gemm_bf16_256x256_kernel<true, 3>), grid, block, 0, stream, b);
        else if (abl == 4) hipLaunchKernelGGL((gemm_bf16_256x256_kernel<true, 4>), grid, block, 0, stream, b);
        else if (abl == 17) hipLaunchKernelGGL((gemm_bf16_256x256_kernel<true, 17>), grid, block, 0, stream, b);
        else if (stagger == 2) hipLaunchKernelGGL((gemm_bf16_256x256_kernel<false, 0, true>), grid, block, 0, stream, b);
        else if (stagger) hipLaunchKernelGGL((gemm_bf16_256x256_kernel<true, 0>), grid, block, 0, stream, b);
        else hipLaunchKernelGGL((gemm_bf16_256x256_kernel<false, 0>), grid, block, 0, stream, b);
        return hipGetLastError() == hipSuccess ? 0 : 2;
    }
    if (var == 1 || var == 2) {
        if (var == 1) hipLaunchKernelGGL((gemm_bf16_256x256_kernel<true, 0, false, false, false, 1>), grid, block, 0, stream, b);
        else hipLaunchKernelGGL((gemm_bf16_256x256_kernel<true, 0, false, false, false, 2>), grid, block, 0, stream, b);
        return hipGetLastError() == hipSuccess ? 0 : 2;
    }
#endif
    (void)grid; (void)block;
    G2_GO_NI(false, 19)
    if (b.trace) {
        std::vector<unsigned long long> h(trace_n);
        (void)hipStreamSynchronize(stream);
        (void)hipMemcpy(h.data(), b.trace, trace_n * 8, hipMemcpyDeviceToHost);
        (void)hipFree(b.trace);
        if (FILE* f = fopen(trace_path, "wb")) { fwrite(h.data(), 8, trace_n, f); fclose(f); }
    }
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
