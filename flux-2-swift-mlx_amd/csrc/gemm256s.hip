// gemm256s.hip — the large-tile bf16 GEMM with ONE WAVE PER SIMD: 256x256x64 tile, 4 waves of 128x128, the accumulators
// in AGPRs, operands by LDS-DMA, every instruction of the K loop hand-placed (inline asm in issue order).
//
// Same contract, tile order and epilogues as the 8-wave kernel of round 1 it replaced (removed in round 3; profiles/NOTEBOOK.md):
// C = A W^T, W [out,in], bf16 or e4m3 (FP8 instantiation) operands.  Why (measured, profiles/r02_f_*): per K-tile the 8-wave form reads
// 192 KB of fragments from LDS and crosses 8 barriers; with 128x128 wave tiles it is 128 KB and 2 barriers, and with one
// wave per SIMD the MFMAs issue back to back (16 cycles each) with the auxiliary instructions in the gaps: 2180-2290
// shader cycles per 64-deep K-tile against 2048 of pure MFMA issue, where the 8-wave kernel needs 2650.
//
// Geometry
//   * 256 threads = 4 waves as 2 (M) x 2 (N); wave tile (16 NI) x 128 = NI x 8 accumulator blocks of 16x16 (NI = 8: 256
//     AGPRs); MFMA 16x16x32 with the W fragment as the A operand, so a lane holds 4 consecutive output columns.
//   * LDS = 2 stages x [256 A rows | 256 W rows] x 128 B = 128 KB.  16-B chunk c of row r sits at c ^ ((r >> 1) & 7);
//     LDS-DMA writes lane-linear, so the swizzle is applied to the per-lane SOURCE address (and to the fragment reads).
//   * a K-tile is 16 LDS-DMA pieces per wave (global_load_lds_dwordx4, 1 KB each: 8 rows x 128 B; scalar base + 32-bit
//     lane offset); piece j of wave w covers rows (4j + w) 8 .. + 8 of its part.  Tile heights 224 / 192 / 160 (NI 7 / 6 /
//     5: wave quantisation, gemm256_pick_ni) keep the LDS image; for NI <= 6 the two A pieces no wave reads are skipped.
//
// Schedule of one K-tile u (T = 16 NI MFMAs per wave; fragments of BOTH K-steps live in registers, so a stage is free
// after the first reads of the iteration and every piece has more than a K-tile to land):
//   S0, MFMAs on K-step 0:  gaps 0 .. R-1: read K-step 1 of tile u <- stage u&1   (R = NI + 8 fragment reads)
//                           gap R+5: lgkmcnt(0) + barrier B1 — every wave has read all of stage u&1 (WAR)
//   from B1+3 to T-2:       the NP pieces of tile u+2 -> stage u&1, one every ~6 MFMAs (M0 written one gap ahead), so the
//                           CU's L2 -> LDS stream never pauses and never bursts
//   S1, MFMAs on K-step 1:  gap B2: vmcnt(c) + barrier — tile u+1's pieces (issued during tile u-1) have landed, the c
//                           pieces of this iteration issued so far stay in flight (RAW)
//                           gaps B2+2 ..: read K-step 0 of tile u+1 <- stage (u+1)&1, done 6 gaps before the end
//   top of the next iteration: lgkmcnt(0).
// The steady-state body is instantiated with compile-time flags: a run-time `if` around an asm statement is a scalar
// branch in an MFMA gap (measured: 2630 cycles per K-tile with them, 2180 without).
#include "dev.h"
#include "kernels.h"
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#include <utility>
#include <vector>

namespace {

constexpr int TN = 256;
constexpr int PART = 256 * 128;   // 32 KB: the A rows or the W rows of a stage
constexpr int STAGE = 2 * PART;   // 64 KB

__device__ __forceinline__ void tile_coords_s(int local, int ntiles, int tiles_m, int tiles_n, int& tm, int& tn) {
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

// v_permlane16_swap_b32 / v_permlane32_swap_b32: rows of 16 lanes traded between two registers (odd rows of a <-> even rows of b; 32: halves)
__device__ __forceinline__ void s_swap16(uint32_t& a, uint32_t& b) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void s_swap32(uint32_t& a, uint32_t& b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}

// WIDE (convolutions with <= 128 output channels): a 512 x 128 tile, the four waves stacked along M (wave tile 128 x 128 as before);
// per K-tile and wave 16 A pieces + 4 W pieces, LDS 2 x (64 + 16) KB = all 160 KB.
template <int NI, bool WIDE = false>
struct Sched {
    static constexpr int G = NI * 8;                 // MFMAs (= gaps) per K-step
    static constexpr int T = 2 * G;
    static constexpr int R = NI + 8;                 // fragment reads per K-step
    static constexpr int B1 = R + 5;
    static constexpr int RD1 = T - R - 6;            // first gap of the next tile's K-step-0 reads
    static constexpr int B2 = RD1 - 2;
    static constexpr int NPA = WIDE ? 16 : (NI <= 6 ? 6 : 8);   // live A pieces per wave (rows (j & 3) 32 + .. >= 16 NI are never read)
    static constexpr int NPW = WIDE ? 4 : 8;
    static constexpr int NP = NPA + NPW;
    static constexpr int Q0 = B1 + 3, Q1 = T - 2;
    static constexpr int q_gap(int n) { return Q0 + (n * (Q1 - Q0) + (NP - 1) / 2) / (NP - 1); }
    static constexpr int piece_at_q(int gap) {
        for (int n = 0; n < NP; ++n)
            if (q_gap(n) == gap) return n;
        return -1;
    }
    static constexpr int piece_at_p(int gap) { return piece_at_q(gap + 1); }
    static constexpr int before_b2() {   // B2's wait comes first in its gap
        int c = 0;
        for (int n = 0; n < NP; ++n) c += q_gap(n) < B2 ? 1 : 0;
        return c;
    }
    static constexpr int a_j(int n) { return NPA != 6 ? n : n + (n >= 3 ? 1 : 0); }   // NPA 6: j = 0,1,2,4,5,6
    static constexpr bool spaced() {
        for (int n = 0; n + 1 < NP; ++n)
            if (q_gap(n + 1) < q_gap(n) + 2) return false;
        return true;
    }
    static_assert(B1 < Q0 && B2 > B1 && RD1 + R <= T - 6 + 0 && spaced() && (!WIDE || NI == 8), "schedule");
};

// FP8 (e4m3 operands, v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales: 32 cycles each, a K-tile = 128 k = NI x 8 MFMAs).
// A lane's operand is 32 bytes = its two 16-byte fragment reads, so the operands of a whole K-tile are (NI + 8) x 8 VGPRs and cannot be
// double-buffered.  They are replaced IN PLACE as soon as their last MFMA has issued (an MFMA reads its sources while it issues; the
// LDS data lands a read latency later):
//   H0, gaps 0-31: rows 0-3 (column-major).  Gaps 0..2(NI-4)-1 read the A operands of rows 4.. of THIS tile <- stage u&1; B1 behind them
//       (WAR: every wave has read all of the stage); the pieces of tile u+2 follow, spread to the end of the iteration
//   H1, gaps 32..: rows 4.. COLUMN-major, so W operand j is free after its (NI-4) MFMAs and is re-read for tile u+1 right behind
//       them; gap 32: vmcnt(c) + barrier B2 (RAW: tile u+1 has landed); gaps 33-40: the A operands of rows 0-3 of tile u+1
//   top of the next iteration: a counted lgkmcnt — the youngest reads (W operands first used behind B1) may still be in flight
template <int NI>
struct SchedF {
    static constexpr int G = NI * 8;                  // MFMAs (= 32-cycle gaps) per K-tile
    static constexpr int H1N = NI - 4;                // rows of the second half
    static constexpr int RA1 = 2 * H1N;               // reads of A rows 4.. (this tile), gaps 0..RA1-1
    static constexpr int B1 = RA1 + 4;
    static constexpr int B2 = 32;
    static constexpr int RA0 = 33;                    // first of the 8 reads of the next tile's A rows 0-3
    static constexpr int w_gap(int n) { return (n / 2 == 7) ? G - 1 : 32 + (n / 2 + 1) * H1N - 1 + (n % 2); }   // half n % 2 of W operand n / 2
    static constexpr int NPA = NI <= 6 ? 6 : 8;
    static constexpr int NPW = 8;
    static constexpr int NP = NPA + 8;
    static constexpr int Q0 = B1 + 2, Q1 = G - 2;
    static constexpr int q_gap(int n) { return Q0 + (n * (Q1 - Q0) + (NP - 1) / 2) / (NP - 1); }
    static constexpr int piece_at_q(int gap) {
        for (int n = 0; n < NP; ++n)
            if (q_gap(n) == gap) return n;
        return -1;
    }
    static constexpr int piece_at_p(int gap) { return piece_at_q(gap + 1); }
    static constexpr int before_b2() {
        int c = 0;
        for (int n = 0; n < NP; ++n) c += q_gap(n) < B2 ? 1 : 0;
        return c;
    }
    static constexpr int a_j(int n) { return NPA == 8 ? n : n + (n >= 3 ? 1 : 0); }
    static constexpr bool spaced() {
        for (int n = 0; n + 1 < NP; ++n)
            if (q_gap(n + 1) < q_gap(n) + 2) return false;
        return true;
    }
    // The wait at the top of an iteration: how many of the previous iteration's next-tile reads, counted from the youngest, may still
    // be in flight — those whose operand is first used behind B1 (which waits for everything).  Issue order inside a gap: A rows 0-3
    // (RA0 + n), the NI = 5 leftovers, then W; A rows 0-3 are needed at gaps 0-3, W operand j at gap 4 j (MFMA 4 j issues BEFORE a
    // filler of the same gap, so "behind B1" is strict).
    static constexpr int a0_gap(int n) { return RA0 + n < G ? RA0 + n : G - 1; }
    static constexpr int top_allow() {
        int allow = 0;
        // walk the 24 reads from the youngest: key = (gap, class, index); classes: 0 = A in its own gap, 1 = A leftover, 2 = W
        bool used[24] = {};
        for (int step = 0; step < 24; ++step) {
            int best = -1, bg = -1, bc = -1, bi = -1;
            for (int k = 0; k < 24; ++k) {
                if (used[k]) continue;
                const bool is_w = k >= 8;
                const int g = is_w ? w_gap(k - 8) : a0_gap(k);
                const int c = is_w ? 2 : (RA0 + k < G ? 0 : 1);
                const int idx = is_w ? k - 8 : k;
                if (g > bg || (g == bg && (c > bc || (c == bc && idx > bi)))) { best = k; bg = g; bc = c; bi = idx; }
            }
            used[best] = true;
            const int need = best >= 8 ? 4 * ((best - 8) / 2) : 0;
            if (need > B1) ++allow; else break;
        }
        return allow < 15 ? allow : 15;
    }
    static_assert(B1 < Q0 && B1 < B2 && RA1 <= 8 && spaced(), "fp8 schedule");
};

// e8m0 1.0 in every byte: the A block-scale "table" of a problem that has none (GemmProblem::a_bscale), read with a row stride of 0
__device__ const uint32_t gs_unit_scales[4] = {0x7F7F7F7Fu, 0x7F7F7F7Fu, 0x7F7F7F7Fu, 0x7F7F7F7Fu};

template <int I> using ic = std::integral_constant<int, I>;
template <bool B> using bc = std::integral_constant<bool, B>;
// f(ic<0>{}), f(ic<1>{}), ... in order
template <class F, int... I>
__device__ __forceinline__ void for_seq(F&& f, std::integer_sequence<int, I...>) {
    (f(ic<I>{}), ...);
}
#define GS_INLINE __attribute__((always_inline))

// CONV: the A source of a K-tile jumps by conv_row_jump at every tap row (GemmProblem::conv_*, kernels.h) and the epilogue is EPI_CONV;
// a separate instantiation, so that the DiT's kernels stay instruction for instruction what they were.
// CFAST: the convolution epilogue's lean form, chosen by the launcher when every problem of the batch qualifies (gemm256s_conv_fast_ok);
// a compile-time choice, because as a run-time branch next to the generic form hipcc merges the two bodies back into the generic one.
// GRT: every problem of the batch is EPI_GATE_RESID on whole, aligned column tiles: the epilogue is the transposed-domain form alone.  Its own
// instantiation because next to the other epilogues hipcc's register allocation spills accumulators at the loop's exit (measured: 118 spills).
// BT: every problem of the batch is a plain EPI_BF16 store (no QK norm + RoPE columns) on whole, aligned column tiles — the V^T launches: the bf16 rows leave
// through the transposed domain as well (own instantiation for the same reason as GRT).
template <int NI, bool FP8 = false, bool CONV = false, bool WIDE = false, bool CFAST = false, bool GRT = false, bool BT = false>
__global__ __launch_bounds__(256, 1) void gemm256s_kernel(GemmBatch batch) {
    static_assert(!BT || (!CONV && !GRT), "the transposed bf16 store belongs to the plain GEMM");
    static_assert(!CFAST || CONV, "the lean epilogue is the convolution's");
    static_assert(!GRT || !CONV, "the transposed gate-residual epilogue belongs to the plain GEMM");
    static_assert(!WIDE || (CONV && !FP8 && NI == 8), "the 512 x 128 tile exists for the convolution form only");
    constexpr int PARTA = WIDE ? 4 * PART / 2 : PART;          // bytes of the A rows of a stage (512 or 256 rows x 128 B)
    constexpr int STG = WIDE ? PARTA + PART / 2 : STAGE;       // A rows + W rows (128 or 256 rows)
    constexpr int TN = WIDE ? 128 : 256;
    __shared__ __attribute__((aligned(16))) uint8_t smem[2 * STG];  // 128 KB (WIDE: 160 KB), the only LDS object of the K loop
    __shared__ unsigned bsred[2][8];   // fp8 SwiGLU epilogue (GemmProblem::c2_q8): block maxima [wave row][MFMA row block], as float bits
    if constexpr (FP8 && !CONV) {
        if (threadIdx.x < 16) (&bsred[0][0])[threadIdx.x] = 0u;   // visible behind the prologue's barrier
    }
    using S = std::conditional_t<FP8, SchedF<NI>, Sched<NI, WIDE>>;
    constexpr int ES = FP8 ? 1 : 2;   // bytes per operand element: the byte geometry of a tile is the same (128-byte rows)
    constexpr int BMT = WIDE ? 512 : 32 * NI, WROWS = WIDE ? 128 : 16 * NI;

    const int bid = blockIdx.x;
    // problem of this workgroup: the last one whose first workgroup is <= bid (constant indices: the by-value batch stays in the
    // kernel-argument segment)
    int pi = 0, pstart = 0;
    if (batch.nprob > 1) {
#pragma unroll
        for (int k = 1; k < GEMM_MAX_PROBS; ++k)
            if (k < batch.nprob && bid >= batch.tile_start[k]) { pi = k; pstart = batch.tile_start[k]; }
    }
    // the descriptor of this tile's problem, read where it lies in the kernel-argument segment (scalar loads): indexing the by-value
    // copy with `pi` makes hipcc move the whole batch to scratch once the struct is large enough
    using KProblem = const __attribute__((address_space(4))) GemmProblem;
    KProblem& p = ((KProblem*)__builtin_amdgcn_kernarg_segment_ptr())[pi];
    static_assert(offsetof(GemmBatch, p) == 0, "the problems are the first member of the only kernel argument");
    const int local = bid - pstart;
    const int tiles_m = (p.M + BMT - 1) / BMT, tiles_n = (p.N + TN - 1) / TN;
    int tm, tn;
    tile_coords_s(local, tiles_m * tiles_n, tiles_m, tiles_n, tm, tn);
    const int m0 = tm * BMT, n0 = tn * TN;

    const int t = threadIdx.x;
    const int lane = t & 63, wid = t >> 6;
    const int wm = WIDE ? wid : wid >> 1, wn = WIDE ? 0 : wid & 1;

    // ---- LDS-DMA sources: piece j of a part -> rows (4j + wid) 8 + (lane >> 3), LDS chunk lane & 7 <- global chunk
    // (lane & 7) ^ ((row >> 1) & 7).  A part row r belongs to wave-row r >> 7: tile row (r & 127) + (r >> 7) WROWS.
    constexpr int NVA = WIDE ? 16 : 8, NVW = WIDE ? 4 : 8;
    uint32_t voffA[NVA], voffW[NVW];
    {
        const int wr8 = wid * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((wr8 >> 1) & 7);
#pragma unroll
        for (int j = 0; j < NVA; ++j) {
            const int r = j * 32 + wr8;
            const int arow = (r & 127) + (r >> 7) * WROWS;
            voffA[j] = (uint32_t)(((size_t)min(m0 + arow, p.M - 1) * p.lda + chunk * (16 / ES)) * ES);
        }
#pragma unroll
        for (int j = 0; j < NVW; ++j) {
            const int r = j * 32 + wr8;
            voffW[j] = (uint32_t)(((size_t)min(n0 + r, p.N - 1) * p.ldw + chunk * (16 / ES)) * ES);
        }
    }
    const uint8_t* gA = reinterpret_cast<const uint8_t*>(p.A);
    const uint8_t* gW = reinterpret_cast<const uint8_t*>(p.W);
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem;
    const uint32_t m0_wave = __builtin_amdgcn_readfirstlane(lds_base + wid * 1024);   // + j*4096 + part + stage
    const int nk = p.K / (FP8 ? 128 : 64);
    const int cseg = CONV ? p.conv_seg : 0x7fffffff;
    const long long cjump = CONV ? p.conv_row_jump : 0;
    // A source of K-tile kt (bytes from gA): at most three tap rows
    auto a_src = [&](int kt) GS_INLINE -> const uint8_t* {
        if constexpr (CONV) return gA + (size_t)kt * 128 + (size_t)((kt >= cseg ? cjump : 0) + (kt >= 2 * cseg ? cjump : 0));
        else return gA + (size_t)kt * 128;
    };

    f32x4 acc[NI][8];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- fragment reads: lane (frow, fq) reads row (block*16 + frow), chunk (4h + fq) ^ ((frow >> 1) & 7) of K-step h
    const int frow = lane & 15, fq = lane >> 4;
    const int sw = (frow >> 1) & 7;
    const uint32_t a_frag = lds_base + (wm * 128 + frow) * 128;
    const uint32_t b_frag = lds_base + PARTA + (wn * 128 + frow) * 128;
    const uint32_t ch0 = ((0 + fq) ^ sw) << 4, ch1 = ((4 + fq) ^ sw) << 4;

    bf16x8 fa0[NI], fb0[8], fa1[NI], fb1[8];

#define GS_DSR(dst_, addr_, off_) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst_) : "v"(addr_), "i"(off_) : "memory")
    // read r of a K-step: A0, B0, A1, B1, .., A(NI-1), B(NI-1), B(NI), .., B7
    auto frag_read = [&](auto rc, bf16x8 (&fa)[NI], bf16x8 (&fb)[8], uint32_t ha, uint32_t hb) GS_INLINE {
        constexpr int r = decltype(rc)::value;
        (void)fa; (void)fb; (void)ha; (void)hb;
        if constexpr (r < 2 * NI) {
            if constexpr ((r & 1) == 0) GS_DSR(fa[r / 2], ha, (r / 2) * 2048);
            else GS_DSR(fb[r / 2], hb, (r / 2) * 2048);
        } else {
            GS_DSR(fb[r - NI], hb, (r - NI) * 2048);
        }
    };
    // piece n of a K-tile: M0 <- LDS destination (P), then the DMA (Q); A pieces first
    auto piece_p = [&](auto nc, uint32_t mm) GS_INLINE {
        constexpr int n = decltype(nc)::value;
        constexpr int off = n < S::NPA ? S::a_j(n) * 4096 : PARTA + (n - S::NPA) * 4096;
        asm volatile("s_add_u32 m0, %0, %1" ::"s"(mm), "i"(off) : "memory", "scc");   // s_add writes SCC: hipcc may hold a loop condition there
    };
    auto piece_q = [&](auto nc, const uint8_t* sa, const uint8_t* sw_) GS_INLINE {
        constexpr int n = decltype(nc)::value;
        (void)voffA; (void)voffW;   // named outside `if constexpr`, or the generic lambda does not capture them
        if constexpr (n < S::NPA) asm volatile("global_load_lds_dwordx4 %0, %1" ::"v"(voffA[S::a_j(n)]), "s"(sa) : "memory");
        else asm volatile("global_load_lds_dwordx4 %0, %1" ::"v"(voffW[n - S::NPA]), "s"(sw_) : "memory");
    };

    // dev diagnostic (batch.trace, FLUX2MI_GEMM_TRACE with the dev library): shader-clock and 100 MHz stamps around prologue + K loop
    unsigned long long tr_c0 = 0, tr_r0 = 0;
    if (batch.trace) asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(tr_c0), "=s"(tr_r0)::"memory");
    if constexpr (!FP8) {
    // ---- prologue: K-tiles 0 and 1 -> stages 0 and 1; K-step 0 of tile 0 -> (fa0, fb0)
    {
        auto fill_all = [&](uint32_t mm, const uint8_t* sa, const uint8_t* sw_) GS_INLINE {
            for_seq([&](auto nc) GS_INLINE {
                piece_p(nc, mm);
                asm volatile("s_nop 0" ::: "memory");
                piece_q(nc, sa, sw_);
            }, std::make_integer_sequence<int, S::NP>{});
        };
        fill_all(m0_wave, gA, gW);
        if (nk > 1) {
            fill_all(m0_wave + (uint32_t)STG, a_src(1), gW + 128);
            asm volatile("s_waitcnt vmcnt(%0)" ::"i"(S::NP) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_barrier" ::: "memory");
        const uint32_t ha = a_frag + ch0, hb = b_frag + ch0;
        for_seq([&](auto rc) GS_INLINE { frag_read(rc, fa0, fb0, ha, hb); }, std::make_integer_sequence<int, S::R>{});
    }

    // ---- one K-tile.  IS: the pieces of tile u+2 are issued; RD: the next tile's first fragments are read
    auto ktile = [&](auto is_c, auto rd_c, int u) GS_INLINE {
        constexpr bool IS = decltype(is_c)::value, RD = decltype(rd_c)::value;
        const uint32_t st_cur = (uint32_t)((u & 1) * STG), st_nxt = (uint32_t)(((u + 1) & 1) * STG);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const uint32_t ha = a_frag + st_cur + ch1, hb = b_frag + st_cur + ch1;
        const uint32_t ha2 = a_frag + st_nxt + ch0, hb2 = b_frag + st_nxt + ch0;
        const uint32_t mm = m0_wave + st_cur;
        const uint8_t* sa = a_src(u + 2);
        const uint8_t* sw_ = gW + (size_t)(u + 2) * 128;
        auto gap = [&](auto gc) GS_INLINE {
            constexpr int I = decltype(gc)::value;
            (void)acc; (void)fa0; (void)fb0; (void)fa1; (void)fb1; (void)ha; (void)hb; (void)ha2; (void)hb2; (void)mm; (void)sa; (void)sw_;
            constexpr int step = I / S::G, g = (I % S::G) / 8, j = I % 8;
            // W fragment as the A operand: acc[i][j] rows = output columns j*16 + fq*4 + r, lanes = output rows i*16 + frow
            if constexpr (step == 0)
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[g][j]) : "v"(fb0[j]), "v"(fa0[g]));
            else
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[g][j]) : "v"(fb1[j]), "v"(fa1[g]));
            if constexpr (I < S::R) frag_read(ic<I>{}, fa1, fb1, ha, hb);
            if constexpr (I == S::B1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if constexpr (I == S::B2) {
                if constexpr (IS) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"i"(S::before_b2()) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            }
            if constexpr (IS && S::piece_at_p(I) >= 0) piece_p(ic<S::piece_at_p(I)>{}, mm);
            if constexpr (IS && S::piece_at_q(I) >= 0) piece_q(ic<S::piece_at_q(I)>{}, sa, sw_);
            if constexpr (RD && I >= S::RD1 && I < S::RD1 + S::R) frag_read(ic<I - S::RD1>{}, fa0, fb0, ha2, hb2);
        };
        for_seq(gap, std::make_integer_sequence<int, S::T>{});
    };
    int u = 0;
    for (; u + 2 < nk; ++u) ktile(bc<true>{}, bc<true>{}, u);
    if (u + 1 < nk) { ktile(bc<false>{}, bc<true>{}, u); ++u; }
    ktile(bc<false>{}, bc<false>{}, u);
    } else {
    // ==== FP8 K loop (see SchedF)
    typedef uint32_t gs_u32x8 __attribute__((ext_vector_type(8)));
    gs_u32x8 wop[8], aop[NI];                       // 32-byte operands; .lo = chunk fq, .hi = chunk 4 + fq of the lane's row, each half written by its own
                                                    // ds_read_b128 into the SAME register tuple
    int unit_scale = 0x7F7F7F7F;                    // e8m0 127 = 1.0 in every byte; the real scales are per row / column, in the epilogue
    asm volatile("" : "+v"(unit_scale));
    // A block scales (GemmProblem::a_bscale): one e8m0 byte per 16-row block and K-tile = per MFMA row block i, uniform over the wave.
    // Per K-tile the wave's NI bytes [rb0, rb0 + NI) of table row kt arrive as two overlapping scalar loads (dwords 0-1 and 1-2 from
    // the 4-byte aligned address below rb0), are shifted down by rb0 % 4 bytes once the K-tile's lgkmcnt(0) has passed, and are moved
    // into the two VGPRs the MFMAs take their activation scale from (byte i % 4 of asc[i / 4], selected by op_sel): asc0 (rows 0-3)
    // behind MFMA 31 for the next tile, asc1 (rows 4..) at gap 0 for this one — replaced in place like the operands.
    const int bs_rb0 = __builtin_amdgcn_readfirstlane((m0 + wm * WROWS) >> 4);
    const uint32_t bs_sh = (uint32_t)(bs_rb0 & 3) * 8u;
    const bool bs_on = p.a_bscale != nullptr;
    const uint8_t* bs_ptr = bs_on ? p.a_bscale + (bs_rb0 & ~3) : reinterpret_cast<const uint8_t*>(gs_unit_scales);
    const int bs_ld = bs_on ? p.a_bscale_ld : 0;
    unsigned long long bs_l0, bs_l1, bs_t0, bs_t1;   // loaded dword pairs, shifted pairs (the low halves are the scales)
    uint32_t asc0, asc1;
    asm volatile("s_load_dwordx2 %0, %2, 0x0\n\ts_load_dwordx2 %1, %2, 0x4\n\ts_waitcnt lgkmcnt(0)" : "=&s"(bs_l0), "=&s"(bs_l1) : "s"(bs_ptr) : "memory");
    asm volatile("s_lshr_b64 %0, %2, %4\n\ts_lshr_b64 %1, %3, %4" : "=&s"(bs_t0), "=&s"(bs_t1) : "s"(bs_l0), "s"(bs_l1), "s"(bs_sh) : "scc");
    asm volatile("v_mov_b32 %0, %1" : "=v"(asc0) : "s"((uint32_t)bs_t0));
    asm volatile("v_mov_b32 %0, %1" : "=v"(asc1) : "s"((uint32_t)bs_t1));
    {
        auto fill_all = [&](uint32_t mm, const uint8_t* sa, const uint8_t* sw_) GS_INLINE {
            for_seq([&](auto nc) GS_INLINE {
                piece_p(nc, mm);
                asm volatile("s_nop 0" ::: "memory");
                piece_q(nc, sa, sw_);
            }, std::make_integer_sequence<int, S::NP>{});
        };
        fill_all(m0_wave, gA, gW);
        if (nk > 1) {
            fill_all(m0_wave + (uint32_t)STG, gA + 128, gW + 128);
            asm volatile("s_waitcnt vmcnt(%0)" ::"i"(S::NP) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_barrier" ::: "memory");
        const uint32_t hal = a_frag + ch0, hah = a_frag + ch1, hbl = b_frag + ch0, hbh = b_frag + ch1;
        for_seq([&](auto ic_) GS_INLINE {
            constexpr int i = decltype(ic_)::value;
            (void)aop; (void)hal; (void)hah;
            GS_DSR(aop[i].lo, hal, i * 2048);
            GS_DSR(aop[i].hi, hah, i * 2048);
        }, std::make_integer_sequence<int, 4>{});
        for_seq([&](auto jc) GS_INLINE {
            constexpr int j = decltype(jc)::value;
            (void)wop; (void)hbl; (void)hbh;
            GS_DSR(wop[j].lo, hbl, j * 2048);
            GS_DSR(wop[j].hi, hbh, j * 2048);
        }, std::make_integer_sequence<int, 8>{});
    }
    auto ktile8 = [&](auto is_c, auto rd_c, int u) GS_INLINE {
        constexpr bool IS = decltype(is_c)::value, RD = decltype(rd_c)::value;
        const uint32_t st_cur = (uint32_t)((u & 1) * STG), st_nxt = (uint32_t)(((u + 1) & 1) * STG);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(S::top_allow()) : "memory");   // the youngest reads belong to operands first used behind B1
        const uint32_t hal = a_frag + st_cur + ch0, hah = a_frag + st_cur + ch1;                    // A rows 4.. of this tile
        const uint32_t hal2 = a_frag + st_nxt + ch0, hah2 = a_frag + st_nxt + ch1;                  // next tile
        const uint32_t hbl2 = b_frag + st_nxt + ch0, hbh2 = b_frag + st_nxt + ch1;
        const uint32_t mm = m0_wave + st_cur;
        const uint8_t* sa = gA + (size_t)(u + 2) * 128;
        const uint8_t* sw_ = gW + (size_t)(u + 2) * 128;
        const uint8_t* bsp = bs_ptr + (size_t)(u + 1) * (size_t)bs_ld;   // the next tile's block scales
        auto gap = [&](auto gc) GS_INLINE {
            constexpr int I = decltype(gc)::value;
            (void)acc; (void)wop; (void)aop; (void)unit_scale; (void)hal; (void)hah; (void)hal2; (void)hah2; (void)hbl2; (void)hbh2;
            (void)mm; (void)sa; (void)sw_; (void)bsp; (void)bs_l0; (void)bs_l1; (void)bs_t0; (void)bs_t1; (void)asc0; (void)asc1; (void)bs_sh;
            // both halves column-major: W operand j is first needed at gap 4 j, i.e. the later it was re-read the later it is used
            constexpr int i = I < 32 ? I % 4 : 4 + (I - 32) % S::H1N, j = I < 32 ? I / 4 : (I - 32) / S::H1N;
            // W operand first: acc[i][j] rows = output columns j*16 + fq*4 + r, lanes = output rows i*16 + frow
            // scale operands: the W operand's (unit), then the activation's: byte i % 4 of asc[i / 4] (op_sel / op_sel_hi = the two bits of the byte index)
            if constexpr (I == 0) asm volatile("v_mov_b32 %0, %1" : "+v"(asc1) : "s"((uint32_t)bs_t1));   // this tile's rows 4..: free since the previous tile's last MFMA
            if constexpr (i < 4) {
                if constexpr (i == 0) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "+a"(acc[i][j]) : "v"(wop[j]), "v"(aop[i]), "v"(unit_scale), "v"(asc0));
                if constexpr (i == 1) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel:[0,1,0] op_sel_hi:[0,0,0]" : "+a"(acc[i][j]) : "v"(wop[j]), "v"(aop[i]), "v"(unit_scale), "v"(asc0));
                if constexpr (i == 2) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "+a"(acc[i][j]) : "v"(wop[j]), "v"(aop[i]), "v"(unit_scale), "v"(asc0));
                if constexpr (i == 3) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "+a"(acc[i][j]) : "v"(wop[j]), "v"(aop[i]), "v"(unit_scale), "v"(asc0));
            } else {
                if constexpr (i == 4) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "+a"(acc[i][j]) : "v"(wop[j]), "v"(aop[i]), "v"(unit_scale), "v"(asc1));
                if constexpr (i == 5) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel:[0,1,0] op_sel_hi:[0,0,0]" : "+a"(acc[i][j]) : "v"(wop[j]), "v"(aop[i]), "v"(unit_scale), "v"(asc1));
                if constexpr (i == 6) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "+a"(acc[i][j]) : "v"(wop[j]), "v"(aop[i]), "v"(unit_scale), "v"(asc1));
                if constexpr (i == 7) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "+a"(acc[i][j]) : "v"(wop[j]), "v"(aop[i]), "v"(unit_scale), "v"(asc1));
            }
            if constexpr (RD && I == 1) asm volatile("s_load_dwordx2 %0, %2, 0x0\n\ts_load_dwordx2 %1, %2, 0x4" : "=&s"(bs_l0), "=&s"(bs_l1) : "s"(bsp) : "memory");
            if constexpr (I < S::RA1) {                      // A rows 4.. of this tile: operand 4 + I / 2, half I % 2
                constexpr int ai = 4 + I / 2;
                if constexpr (I % 2 == 0) GS_DSR(aop[ai].lo, hal, ai * 2048);
                else GS_DSR(aop[ai].hi, hah, ai * 2048);
            }
            if constexpr (I == S::B1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            // (the scalar loads of gap 1 are behind that wait) the next tile's bytes, shifted to bit 0
            if constexpr (RD && I == S::B1 + 1) asm volatile("s_lshr_b64 %0, %2, %4\n\ts_lshr_b64 %1, %3, %4" : "=&s"(bs_t0), "=&s"(bs_t1) : "s"(bs_l0), "s"(bs_l1), "s"(bs_sh) : "scc");
            if constexpr (RD && I == 33) asm volatile("v_mov_b32 %0, %1" : "+v"(asc0) : "s"((uint32_t)bs_t0));   // rows 0-3 of the next tile: free behind MFMA 31
            if constexpr (I == S::B2) {
                if constexpr (IS) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"i"(S::before_b2()) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            }
            if constexpr (IS && S::piece_at_p(I) >= 0) piece_p(ic<S::piece_at_p(I)>{}, mm);
            if constexpr (IS && S::piece_at_q(I) >= 0) piece_q(ic<S::piece_at_q(I)>{}, sa, sw_);
            if constexpr (RD && I >= S::RA0 && I < S::RA0 + 8 && I < S::G) {   // the next tile's A rows 0-3 (free behind MFMA 31)
                constexpr int n = I - S::RA0, ai = n / 2;
                if constexpr (n % 2 == 0) GS_DSR(aop[ai].lo, hal2, ai * 2048);
                else GS_DSR(aop[ai].hi, hah2, ai * 2048);
            }
            if constexpr (RD && I == S::G - 1 && S::RA0 + 8 > S::G) {          // NI = 5: the reads that did not fit behind gap 32
                for_seq([&](auto nc) GS_INLINE {
                    constexpr int n = S::G - S::RA0 + decltype(nc)::value, ai = n / 2;
                    (void)aop; (void)hal2; (void)hah2;
                    if constexpr (n % 2 == 0) GS_DSR(aop[ai].lo, hal2, ai * 2048);
                    else GS_DSR(aop[ai].hi, hah2, ai * 2048);
                }, std::make_integer_sequence<int, (S::RA0 + 8 > S::G ? S::RA0 + 8 - S::G : 0)>{});
            }
            if constexpr (RD) {   // the next tile's W operands, in place, behind the last MFMA of their column
                for_seq([&](auto nc) GS_INLINE {
                    constexpr int n = decltype(nc)::value, wj = n / 2;
                    (void)wop; (void)hbl2; (void)hbh2;
                    if constexpr (S::w_gap(n) == I) {
                        if constexpr (n % 2 == 0) GS_DSR(wop[wj].lo, hbl2, wj * 2048);
                        else GS_DSR(wop[wj].hi, hbh2, wj * 2048);
                    }
                }, std::make_integer_sequence<int, 16>{});
            }
        };
        for_seq(gap, std::make_integer_sequence<int, S::G>{});
    };
    {
        int u = 0;
        for (; u + 2 < nk; ++u) ktile8(bc<true>{}, bc<true>{}, u);
        if (u + 1 < nk) { ktile8(bc<false>{}, bc<true>{}, u); ++u; }
        ktile8(bc<false>{}, bc<false>{}, u);
    }
    }
#undef GS_DSR
    // MFMA result -> first non-MFMA reader: hipcc pads nothing behind inline asm (guide §5.7 item 2)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
#ifdef FLUX2MI_DEV
    unsigned long long tr_c1 = 0;
#endif
    if (batch.trace) {
        unsigned long long c1, r1;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1)::"memory");
        if (lane == 0) {
            unsigned long long* o = batch.trace + ((size_t)bid * 4 + wid) * 8;
            o[0] = c1 - tr_c0; o[1] = r1 - tr_r0; o[2] = (unsigned long long)nk; o[3] = 0;
        }
#ifdef FLUX2MI_DEV
        tr_c1 = c1;
#endif
    }
    __builtin_amdgcn_sched_barrier(0);   // the epilogue's address arithmetic stays behind the K loop (hoisted into the last K-tile it costs the
                                         // registers the fp8 operands need: spills)

    // ---- epilogue (with a wave's 128 columns taken as two 64-column halves h = 0, 1:
    // its `wn` is wn*2 + h here and its acc[i][j] is acc[i][4h + j]).  Lane (frow, fq) holds, for output row
    // m0 + wm*WROWS + i*16 + frow, the FOUR CONSECUTIVE columns n0 + wn*128 + j*16 + fq*4 + r of block j = 0..7.
    const int row_base = m0 + wm * WROWS + frow;
    // FP8: de-quantise at the point of use (per-row activation scale x per-column weight scale).  The accumulators are never rewritten:
    // a pass that scales them in place needs new AGPRs while the old values are live, and hipcc then spills accumulators right behind
    // the asm MFMA that produced them — a store of a register whose MFMA is still in flight.
    float fs_row[NI];
    f32x4 fs_col[8];
    if constexpr (FP8) {
#pragma unroll
        for (int i = 0; i < NI; ++i) fs_row[i] = p.a_bscale ? 1.0f : p.a_scale[(unsigned)min(row_base + i * 16, p.M - 1)];
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) fs_col[j][r] = p.w_scale[min(n0 + wn * 128 + j * 16 + fq * 4 + r, p.N - 1)];
    }
    auto accv = [&](int i, int j) GS_INLINE -> f32x4 {
        if constexpr (FP8) return acc[i][j] * (fs_col[j] * fs_row[i]);
        else return acc[i][j];
    };
    auto gs_pack = [&](float lo, float hi) GS_INLINE -> uint32_t {
        if constexpr (FP8) return pack_bf16x2_3op(lo, hi);
        else return pack_bf16x2(lo, hi);
    };
    if constexpr (CONV) {
#ifdef FLUX2MI_DEV
        auto conv_trace_end = [&]() GS_INLINE {
            if (batch.trace) {   // dev diagnostic: the convolution epilogue, loop end -> everything issued -> every store retired (shader cycles)
                unsigned long long c2, c3;
                asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c2)::"memory");
                asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c3)::"memory");
                if (lane == 0) batch.trace[((size_t)bid * 4 + wid) * 8 + 3] = ((c2 - tr_c1) & 0xffffffffull) | ((c3 - c2) << 32);
            }
        };
#else
        auto conv_trace_end = [&]() GS_INLINE {};
#endif
        // ---- EPI_CONV, the lean form (round 5; profiles/r05_*): the production layers' case — whole 16-byte runs (N % 32 == 0, 16-byte
        // aligned rows) and an image wider than 16 pixels.  The generic form below spent 30-54k cycles per tile here, as long as an 18-K-tile
        // loop (in-kernel stamps, FLUX2MI_GEMM_TRACE): every problem field re-read from the kernel-argument segment at each use (hipcc
        // rematerialises such loads: ~100 exposed scalar-load waits per tile), 64-bit address arithmetic and a division per row and half,
        // the residual in four batches of exposed HBM latency.  Here: every field once into SGPRs (opaque, so they stay there), the rows'
        // pixel arithmetic once per tile and incrementally (rows of a lane are 16 apart: one division), 32-bit element offsets, and ALL
        // residual loads of the tile in flight before the first accumulator is touched.  Same arithmetic on the same values in the same
        // order as the generic form: bit-identical outputs and sums.
        {
            unsigned long long qC = reinterpret_cast<unsigned long long>(p.C), qR = reinterpret_cast<unsigned long long>(p.resid);
            unsigned long long qV = reinterpret_cast<unsigned long long>(p.vec), qG = reinterpret_cast<unsigned long long>(p.gn_part);
            uint32_t qN = (uint32_t)p.N, qM = (uint32_t)p.M, qldc = (uint32_t)p.ldc, qldr = (uint32_t)p.ldr, qwp = (uint32_t)p.conv_wp;
            uint32_t qwi = (uint32_t)p.conv_w, qup = (uint32_t)p.conv_up, qopad = (uint32_t)p.conv_out_padded;
            uint32_t qpy = (uint32_t)p.conv_py, qpx = (uint32_t)p.conv_px;
            asm volatile("" : "+s"(qC), "+s"(qR), "+s"(qV), "+s"(qG));
            asm volatile("" : "+s"(qN), "+s"(qM), "+s"(qldc), "+s"(qldr), "+s"(qwp), "+s"(qwi), "+s"(qup), "+s"(qopad), "+s"(qpy), "+s"(qpx));
            if constexpr (CFAST) {
                // global-memory pointers (from an integer hipcc would make them generic: flat_load / flat_store)
                typedef __attribute__((address_space(1))) uint16_t g_u16;
                typedef __attribute__((address_space(1))) float g_f32;
                typedef __attribute__((address_space(1))) u32x4 g_u4;
                typedef __attribute__((address_space(1))) f32x4 g_f4;
                typedef float gs_f32x2 __attribute__((ext_vector_type(2)));
                g_u16* const Cq = (g_u16*)qC;
                const g_u16* const Rq = (const g_u16*)qR;
                const g_f32* const Vq = (const g_f32*)qV;
                const bool want_gn = qG != 0;
                const uint32_t wo = qup ? 2u * qwi : qwi;
                const uint32_t cb = (uint32_t)(n0 + wn * 128);
                // ---- the TRANSPOSED domain.  A lane of the accumulator layout owns ONE row of a 16-row block (row = lane & 15): a 16-byte store
                // per lane is then 64 separate accesses, which a CU's vector-memory path takes at 16 B/clk whatever the addresses
                // (tools/store_pattern.hip, profiles/r05_a_store_pattern_probe.txt: 15.98 B/clk per CU for this pattern, 63 for lanes that
                // cover whole rows; the K loops of the other CUs leave the chip's 7.2 TB/s far from binding).  So every (row block, 64-column
                // half) goes through a private LDS tile as f32 — 4 ds_write_b128 in the accumulator layout, 4 ds_read_b128 in the
                // transposed one — and everything behind it (bias, residual, GroupNorm sums, rounding, stores) happens where lane
                // l owns 8 consecutive columns (l & 7) of rows (l >> 3) and (l >> 3) + 8: a store instruction is 8 whole 128-byte runs.
                const int tr = lane >> 3, tc = lane & 7;
                float* const tile = reinterpret_cast<float*>(smem + 36864 + wid * 8704);            // 2 x [16][64] f32, behind the sums' 33 KB
                // 16-byte chunk c of row r lives at chunk c ^ (r & 7) of a 256-byte row: conflict-free for BOTH access patterns (the padded
                // pitch of 68 floats left the transposed reads 2-way on a third of their lane groups: SQ_LDS_BANK_CONFLICT 8 % of the
                // 512 x 128 tile's LDS cycles, profiles/r05_m_vae_pmc_sq.txt).  The XOR only moves a lane's chunks among fixed slots: two
                // base addresses per side (even / odd block of the write, first / second chunk of the read) and immediate offsets as before.
                constexpr int TP = 64;                                                              // pitch in floats (256 B)
                const int wsw = frow & 7, rsw = tr & 7;
                float* const wr0 = tile + frow * TP + (((wsw >> 2) << 2) | (fq ^ (wsw & 3))) * 4;          // blocks jj = 0, 2: + jj * 16
                float* const wr1 = tile + frow * TP + ((((wsw >> 2) ^ 1) << 2) | (fq ^ (wsw & 3))) * 4;    // blocks jj = 1, 3: + (jj - 1) * 16
                const float* const rd0 = tile + tr * TP + ((2 * tc) ^ rsw) * 4;                     // + t * 8 * TP: rows tr, tr + 8 (same r & 7)
                const float* const rd1 = tile + tr * TP + ((2 * tc + 1) ^ rsw) * 4;
                // the lane's 2 NI rows (tr + 8 n of the wave's rows): pixel arithmetic once, incrementally (steps of 8 rows: one division)
                uint32_t offC[2 * NI], offR[2 * NI];
                uint32_t ibits = 0u, sbits = 0u;     // bit n = row n is an interior pixel / is stored
                {
                    uint32_t r0 = (uint32_t)(m0 + wm * WROWS + tr);
                    asm volatile("" : "+v"(r0));
                    uint32_t q = r0 + qwp + 1u;
                    uint32_t yy = q / qwp, xx = q - yy * qwp;          // padded coordinates of the first row
                    const uint32_t wcol = cb + (uint32_t)tc * 8u;
                    if (!qup && !qopad) {
                        // the plain case (same-size output, unpadded): branch-free, 24-bit multiplies — the general form below is ~45
                        // instructions per row with three branches, 16 rows per lane, in front of the tile's first load
                        uint32_t opl = r0 - 2u * (yy - 1u);            // plain pixel index of an interior row: q - (wp + 1) - 2 (yy - 1)
#pragma unroll
                        for (int n = 0; n < 2 * NI; ++n) {
                            const bool in = (r0 + 8u * n < qM) && (xx - 1u < qwi);
                            const uint32_t o = in ? opl : 0u;
                            offC[n] = __umul24(o, qldc) + wcol;
                            offR[n] = __umul24(o, qldr) + wcol;
                            ibits |= (in ? 1u : 0u) << n;
                            xx += 8u; opl += 8u;
                            const bool wrap = xx >= qwp;
                            xx -= wrap ? qwp : 0u;
                            opl -= wrap ? 2u : 0u;
                        }
                        sbits = ibits;
                    } else
#pragma unroll
                    for (int n = 0; n < 2 * NI; ++n) {
                        const bool live = r0 + 8u * n < qM;
                        const bool in = live && xx >= 1u && xx <= qwi;
                        const uint32_t oy = qup ? 2u * (yy - 1u) + qpy : yy - 1u;
                        const uint32_t ox = in ? (qup ? 2u * (xx - 1u) + qpx : xx - 1u) : 0u;
                        const uint32_t opl = oy * wo + ox;
                        const uint32_t oo = qopad ? (in ? (oy + 1u) * (wo + 2u) + (ox + 1u) : q) : opl;
                        const bool st = in || (live && qopad && !qup);   // a halo column of a padded output is written as zeros
                        offC[n] = st ? oo * qldc + wcol : wcol;
                        offR[n] = in ? opl * qldr + wcol : wcol;
                        ibits |= (in ? 1u : 0u) << n;
                        sbits |= (st ? 1u : 0u) << n;
                        q += 8u; xx += 8u;
                        if (xx >= qwp) { xx -= qwp; yy += 1u; }
                    }
                }
                // row blocks whose 16 rows are interior pixels in every lane (all but the blocks that touch a halo column: two pixels of an
                // image row) take a straight-line form: no exec masks, no branches — with one wave per SIMD every instruction is ~4 cycles
                uint32_t fullmask = 0u;
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const bool both = ((ibits >> (2 * i)) & 3u) == 3u;
                    fullmask |= (__builtin_amdgcn_ballot_w64(both) == ~0ull ? 1u : 0u) << i;
                }
                auto bias_load = [&](int h, f32x4 (&b)[2]) GS_INLINE {
                    const uint32_t col = cb + h * 64 + tc * 8;
                    b[0] = f32x4{0.f, 0.f, 0.f, 0.f}; b[1] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (Vq != nullptr && col < qN) { b[0] = *(const g_f4*)(Vq + col); b[1] = *(const g_f4*)(Vq + col + 4); }
                };
                // bias and residual of a half are pinned (asm operands) at the top of the half: hipcc waits for them ONCE there.  Left to its
                // own wait insertion every use behind a branch got a vmcnt(0) of its own, which also waits for every store issued so far (one
                // in-order counter): the stores went out one store latency apart.  The second half's residual is requested slot by slot as the
                // first half consumes its values.
                f32x4 bsv[2];
                bias_load(0, bsv);
                u32x4 rv[2 * NI];   // ext-vector slots: a slot is written by the load itself
                const bool colv0 = cb + tc * 8u < qN, colv1 = cb + 64u + tc * 8u < qN;
                const bool colv_all = cb + 128u <= qN;   // uniform: the wave's 128 columns all exist
#pragma unroll
                for (int n = 0; n < 2 * NI; ++n) {
                    rv[n] = u32x4{0u, 0u, 0u, 0u};
                    if (Rq && colv0) rv[n] = *(const g_u4*)(Rq + offR[n]);
                }
                asm volatile("" ::: "memory");   // the loads stay in front of the arithmetic
#ifdef FLUX2MI_DEV
                auto lean_stamp = [&](int k) GS_INLINE {
                    if (batch.trace) {
                        unsigned long long c;
                        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c)::"memory");
                        if (lane == 0) batch.trace[((size_t)bid * 4 + wid) * 8 + 4 + k] = c - tr_c1;
                    }
                };
#else
                auto lean_stamp = [&](int) GS_INLINE {};
#endif
                lean_stamp(0);
                __syncthreads();   // LDS holds the last K-tile until every wave has left the K loop
                asm volatile("" : "+v"(bsv[0]), "+v"(bsv[1]));
#pragma unroll
                for (int n = 0; n < 2 * NI; ++n) asm volatile("" : "+v"(rv[n]));
                lean_stamp(1);
                float* red = reinterpret_cast<float*>(smem);   // gn_part: [quantity][wave][tr][128 + 1] floats
                constexpr int RW = 129;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    gs_f32x2 gs[4], gq[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) { gs[k] = gs_f32x2{0.f, 0.f}; gq[k] = gs_f32x2{0.f, 0.f}; }
                    const bool colv = h == 0 ? colv0 : colv1;
                    if (h == 1) {
                        bias_load(1, bsv);
                        asm volatile("" : "+v"(bsv[0]), "+v"(bsv[1]));
                        // unconditional pins: hipcc's wait insertion does not correlate branches on the same condition
#pragma unroll
                        for (int n = 0; n < 2 * NI; ++n) asm volatile("" : "+v"(rv[n]));
                        lean_stamp(3);
                    }
                    // software pipeline over the row blocks (two LDS tiles per wave): block i + 2 is written and block i + 1 read back while
                    // block i is worked on
                    auto tile_w = [&](int i) GS_INLINE {
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj)
                            *reinterpret_cast<f32x4*>(((jj & 1) ? wr1 : wr0) + (i & 1) * (16 * TP) + (jj & 2) * 16) = acc[i][4 * h + jj];
                    };
                    auto tile_r = [&](int i, f32x4 (&v)[4]) GS_INLINE {
                        const float* r0 = rd0 + (i & 1) * (16 * TP);
                        const float* r1 = rd1 + (i & 1) * (16 * TP);
                        v[0] = *reinterpret_cast<const f32x4*>(r0); v[1] = *reinterpret_cast<const f32x4*>(r1);
                        v[2] = *reinterpret_cast<const f32x4*>(r0 + 8 * TP); v[3] = *reinterpret_cast<const f32x4*>(r1 + 8 * TP);
                    };
                    f32x4 cur[4], nxt[4];
                    tile_w(0);
                    if (NI > 1) tile_w(1);
                    tile_r(0, cur);
#pragma unroll
                    for (int i = 0; i < NI; ++i) {
                        if (i + 1 < NI) tile_r(i + 1, nxt);
                        if (i + 2 < NI) tile_w(i + 2);        // behind block i's reads in the LDS queue: its tile is free
                        if (((fullmask >> i) & 1u) && colv_all) {   // uniform: 16 interior rows in every lane, the straight-line form
#pragma unroll
                            for (int tt = 0; tt < 2; ++tt) {
                                const int n = 2 * i + tt;
                                f32x4 va = cur[2 * tt] + bsv[0], vb = cur[2 * tt + 1] + bsv[1];
                                if (Rq) {
                                    const u32x4 r4 = rv[n];
                                    va[0] += lo_bf16(r4[0]); va[1] += hi_bf16(r4[0]); va[2] += lo_bf16(r4[1]); va[3] += hi_bf16(r4[1]);
                                    vb[0] += lo_bf16(r4[2]); vb[1] += hi_bf16(r4[2]); vb[2] += lo_bf16(r4[3]); vb[3] += hi_bf16(r4[3]);
                                    if (h == 0) rv[n] = *(const g_u4*)(Rq + offR[n] + 64u);   // the second half's, into the slot just read
                                }
                                if (want_gn) {
                                    const gs_f32x2 a0 = {va[0], va[1]}, a1 = {va[2], va[3]}, b0 = {vb[0], vb[1]}, b1 = {vb[2], vb[3]};
                                    gs[0] += a0; gq[0] += a0 * a0; gs[1] += a1; gq[1] += a1 * a1;
                                    gs[2] += b0; gq[2] += b0 * b0; gs[3] += b1; gq[3] += b1 * b1;
                                }
                                *(g_u4*)(Cq + offC[n] + h * 64u) = u32x4{pack_bf16x2(va[0], va[1]), pack_bf16x2(va[2], va[3]), pack_bf16x2(vb[0], vb[1]), pack_bf16x2(vb[2], vb[3])};
                            }
                        } else
#pragma unroll
                        for (int tt = 0; tt < 2; ++tt) {
                            const int n = 2 * i + tt;
                            const bool in = (ibits >> n) & 1u, st = (sbits >> n) & 1u;
                            f32x4 va = cur[2 * tt], vb = cur[2 * tt + 1];
                            if (in && colv) {
                                va += bsv[0]; vb += bsv[1];
                                if (Rq) {
                                    const u32x4 r4 = rv[n];
                                    va[0] += lo_bf16(r4[0]); va[1] += hi_bf16(r4[0]); va[2] += lo_bf16(r4[1]); va[3] += hi_bf16(r4[1]);
                                    vb[0] += lo_bf16(r4[2]); vb[1] += hi_bf16(r4[2]); vb[2] += lo_bf16(r4[3]); vb[3] += hi_bf16(r4[3]);
                                    if (h == 0 && colv1) rv[n] = *(const g_u4*)(Rq + offR[n] + 64u);   // the second half's, into the slot just read
                                }
                                if (want_gn) {
                                    const gs_f32x2 a0 = {va[0], va[1]}, a1 = {va[2], va[3]}, b0 = {vb[0], vb[1]}, b1 = {vb[2], vb[3]};
                                    gs[0] += a0; gq[0] += a0 * a0; gs[1] += a1; gq[1] += a1 * a1;
                                    gs[2] += b0; gq[2] += b0 * b0; gs[3] += b1; gq[3] += b1 * b1;
                                }
                                *(g_u4*)(Cq + offC[n] + h * 64u) = u32x4{pack_bf16x2(va[0], va[1]), pack_bf16x2(va[2], va[3]), pack_bf16x2(vb[0], vb[1]), pack_bf16x2(vb[2], vb[3])};
                            } else if (st && colv) {   // a halo column of a padded output: zeros (rare: two pixels of an image row)
                                *(g_u4*)(Cq + offC[n] + h * 64u) = u32x4{0u, 0u, 0u, 0u};
                            }
                        }
#pragma unroll
                        for (int k = 0; k < 4; ++k) cur[k] = nxt[k];
                    }
                    if (h == 0) lean_stamp(2);
                    if (want_gn) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            const int cc = h * 64 + tc * 8 + k;
                            red[(wid * 8 + tr) * RW + cc] = gs[k >> 1][k & 1];
                            red[(32 + wid * 8 + tr) * RW + cc] = gq[k >> 1][k & 1];
                        }
                    }
                }
                if (want_gn) {
                    // ---- the tile's per-channel (sum, sum of squares): one thread per column adds the 8 row groups of every wave that holds the
                    // column, in a fixed order
                    __syncthreads();
                    constexpr int NWV = WIDE ? 4 : 2;                 // waves that hold a column
                    const int cidx = WIDE ? (t & 127) : t;            // column of the tile; WIDE: threads 128.. do the squares
                    const int cc = cidx & 127, w0 = WIDE ? 0 : (cidx >> 7);
                    const int col = n0 + cidx;
                    const int quant0 = WIDE ? (t >> 7) : 0;
                    float va[NWV][8], vb[WIDE ? 1 : NWV][8];
#pragma unroll
                    for (int w = 0; w < NWV; ++w) {
                        const int wv = WIDE ? w : w * 2 + w0;
#pragma unroll
                        for (int f = 0; f < 8; ++f) {
                            va[w][f] = red[(quant0 * 32 + wv * 8 + f) * RW + cc];
                            if constexpr (!WIDE) vb[w][f] = red[(32 + wv * 8 + f) * RW + cc];
                        }
                    }
                    float a0 = 0.f, a1 = 0.f;
#pragma unroll
                    for (int w = 0; w < NWV; ++w)
#pragma unroll
                        for (int f = 0; f < 8; ++f) {
                            a0 += va[w][f];
                            if constexpr (!WIDE) a1 += vb[w][f];
                        }
                    if ((uint32_t)col < qN) {
                        g_f32* dst = (g_f32*)qG + ((size_t)tm * qN + col) * 2;
                        if (WIDE) dst[quant0] = a0;
                        else { dst[0] = a0; dst[1] = a1; }
                    }
                }
                conv_trace_end();
                return;
            }
        }
        // ---- EPI_CONV, the generic form (any N % 4 == 0, any alignment, images down to 8 pixels wide): GEMM row -> pixel (kernels.h, GemmProblem::conv_*); bias, optional plain-layout residual, bf16 NHWC store.
        // 16-byte accesses: the 8-byte runs of accumulator blocks 2jp, 2jp + 1 are paired across the fq lanes of a row
        // (v_permlane16_swap: lane fq then owns the 8 consecutive columns starting at (fq & 1) 16 + (fq >> 1) 8 of the 32-column pair);
        // the residual is loaded in that arrangement and swapped back in front of the f32 add.  The exchange stays inside a GEMM row,
        // so the row's validity is uniform across the lanes that trade.
        uint16_t* C = reinterpret_cast<uint16_t*>(p.C);
        uint32_t wp = (uint32_t)p.conv_wp;
        // the per-row pixel arithmetic (a division per row) depends on this value only: tie it to a point BEHIND the K loop, or hipcc
        // hoists it in front of the loop and carries eight registers through it (spills: every VGPR of the loop is spoken for)
        asm volatile("" : "+s"(wp));
        const uint32_t wi = (uint32_t)p.conv_w;
        const int up = p.conv_up, opad = p.conv_out_padded;
        const uint32_t wo = up ? 2 * wi : wi;
        const int cbase = n0 + wn * 128;
        const bool vec16 = (p.N & 31) == 0 && (p.ldc & 7) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0 &&
                           (!p.resid || ((p.ldr & 7) == 0 && (reinterpret_cast<uintptr_t>(p.resid) & 15) == 0));
        const int wcol = cbase + (fq & 1) * 16 + (fq >> 1) * 8;
        const bool want_gn = p.gn_part != nullptr;
        float* red = reinterpret_cast<float*>(smem);   // gn_part: [quantity][wave][frow][128 + 1] floats
        constexpr int RW = 129;
        // the tile's 128 columns per wave in two halves of 64 (register budget: the accumulators fill the AGPRs, the epilogue lives in
        // the 256 VGPRs), rows in batches of four: the batch's residual loads (8 x 16 B per lane) are all issued before the first is
        // consumed (issued one by one in front of their use they ran at ~2 TB/s chip-wide: +115 us on a 1024^2 x 128 layer)
        constexpr int RBATCH = 4;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            typedef float gs_f32x2 __attribute__((ext_vector_type(2)));
            gs_f32x2 gs[4][2], gq[4][2];    // gn_part: this lane's sums over its valid rows, per (column block of the half, column pair):
                                            // two-wide so that hipcc emits v_pk_add_f32 / v_pk_fma_f32 (half the VALU issue of the scalar form)
            f32x4 bsv[4];                   // bias of the half's column blocks
            int rbase = row_base;           // opaque per half: the rows' pixel arithmetic is recomputed, not carried from one half to the
            asm volatile("" : "+v"(rbase)); // other (eight rows x several values would not fit next to the batch in flight: spills)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    gs[j][r >> 1] = gs_f32x2{0.f, 0.f}; gq[j][r >> 1] = gs_f32x2{0.f, 0.f};
                    const int col = cbase + (4 * h + j) * 16 + fq * 4 + r;
                    bsv[j][r] = (p.vec != nullptr && col < p.N) ? p.vec[col] : 0.f;
                }
#pragma unroll
            for (int i0 = 0; i0 < NI; i0 += RBATCH) {
                uint32_t o_plain[RBATCH], o_out[RBATCH];   // pixel indices
                bool interior[RBATCH], store[RBATCH];
                uint4 rv[RBATCH][2];
#pragma unroll
                for (int ii = 0; ii < RBATCH; ++ii) {
                    const int i = i0 + ii;
                    if (i >= NI) break;
                    const int row = rbase + i * 16;
                    const bool live = row < p.M;
                    const uint32_t q = (uint32_t)min(row, p.M - 1) + wp + 1u;
                    const uint32_t yy = q / wp, xx = q - yy * wp;          // padded coordinates; the row range keeps yy in [1, H]
                    interior[ii] = live && xx >= 1u && xx <= wi;
                    const uint32_t oy = up ? 2u * (yy - 1u) + (uint32_t)p.conv_py : yy - 1u;
                    const uint32_t ox = interior[ii] ? (up ? 2u * (xx - 1u) + (uint32_t)p.conv_px : xx - 1u) : 0u;
                    o_plain[ii] = oy * wo + ox;
                    o_out[ii] = opad ? (interior[ii] ? (oy + 1u) * (wo + 2u) + (ox + 1u) : q) : o_plain[ii];
                    store[ii] = interior[ii] || (live && opad && !up);   // a halo column of a padded output is written as zeros
                    if (vec16 && p.resid) {
#pragma unroll
                        for (int jq = 0; jq < 2; ++jq) {
                            const int col = wcol + (2 * h + jq) * 32;
                            rv[ii][jq] = make_uint4(0u, 0u, 0u, 0u);
                            if (interior[ii] && col < p.N) rv[ii][jq] = *reinterpret_cast<const uint4*>(p.resid + (size_t)o_plain[ii] * p.ldr + col);
                        }
                    }
                }
#pragma unroll
                for (int ii = 0; ii < RBATCH; ++ii) {
                    const int i = i0 + ii;
                    if (i >= NI) break;
                    if (vec16) {
#pragma unroll
                        for (int jq = 0; jq < 2; ++jq) {
                            const int jp = 2 * h + jq;
                            const int col = wcol + jp * 32;
                            f32x4 vx = acc[i][2 * jp] + bsv[2 * jq], vy = acc[i][2 * jp + 1] + bsv[2 * jq + 1];
                            if (p.resid) {
                                uint4 r4 = rv[ii][jq];
                                s_swap16(r4.x, r4.z);
                                s_swap16(r4.y, r4.w);
                                vx[0] += lo_bf16(r4.x); vx[1] += hi_bf16(r4.x); vx[2] += lo_bf16(r4.y); vx[3] += hi_bf16(r4.y);
                                vy[0] += lo_bf16(r4.z); vy[1] += hi_bf16(r4.z); vy[2] += lo_bf16(r4.w); vy[3] += hi_bf16(r4.w);
                            }
                            if (want_gn) {
                                const float m = interior[ii] ? 1.0f : 0.0f;
                                const gs_f32x2 m2 = {m, m};
#pragma unroll
                                for (int k = 0; k < 2; ++k) {
                                    const gs_f32x2 ax = gs_f32x2{vx[2 * k], vx[2 * k + 1]} * m2, ay = gs_f32x2{vy[2 * k], vy[2 * k + 1]} * m2;
                                    gs[2 * jq][k] += ax; gq[2 * jq][k] += ax * ax;
                                    gs[2 * jq + 1][k] += ay; gq[2 * jq + 1][k] += ay * ay;
                                }
                            }
                            uint32_t x0 = pack_bf16x2(vx[0], vx[1]), x1 = pack_bf16x2(vx[2], vx[3]);
                            uint32_t y0 = pack_bf16x2(vy[0], vy[1]), y1 = pack_bf16x2(vy[2], vy[3]);
                            s_swap16(x0, y0);
                            s_swap16(x1, y1);
                            if (store[ii] && col < p.N)
                                *reinterpret_cast<uint4*>(C + (size_t)o_out[ii] * p.ldc + col) = interior[ii] ? make_uint4(x0, x1, y0, y1) : make_uint4(0u, 0u, 0u, 0u);
                        }
                    } else if (store[ii]) {
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) {
                            const int j = 4 * h + jj;
                            const int col = cbase + j * 16 + fq * 4;
                            if (col >= p.N) continue;                      // N % 4 == 0 (launcher)
                            f32x4 v = acc[i][j] + bsv[jj];
                            if (p.resid && interior[ii]) {
                                const uint2 r2 = *reinterpret_cast<const uint2*>(p.resid + (size_t)o_plain[ii] * p.ldr + col);
                                v[0] += lo_bf16(r2.x); v[1] += hi_bf16(r2.x); v[2] += lo_bf16(r2.y); v[3] += hi_bf16(r2.y);
                            }
                            if (want_gn && interior[ii]) {
#pragma unroll
                                for (int k = 0; k < 2; ++k) {
                                    const gs_f32x2 a = {v[2 * k], v[2 * k + 1]};
                                    gs[jj][k] += a; gq[jj][k] += a * a;
                                }
                            }
                            *reinterpret_cast<uint2*>(C + (size_t)o_out[ii] * p.ldc + col) =
                                interior[ii] ? make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])) : make_uint2(0u, 0u);
                        }
                    }
                }
            }
            if (want_gn) {
                // the barrier in front of the first write: LDS holds the last K-tile until every wave has left the K loop
                if (h == 0) __syncthreads();
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int cc = (4 * h + jj) * 16 + fq * 4 + r;
                        red[(wid * 16 + frow) * RW + cc] = gs[jj][r >> 1][r & 1];
                        red[(64 + wid * 16 + frow) * RW + cc] = gq[jj][r >> 1][r & 1];
                    }
            }
        }
        if (want_gn) {
            // ---- the tile's per-channel (sum, sum of squares): one thread per column adds the 16 frow lanes of every wave that holds
            // the column, in a fixed order
            __syncthreads();
            constexpr int NWV = WIDE ? 4 : 2;                 // waves that hold a column
            const int cidx = WIDE ? (t & 127) : t;            // column of the tile; WIDE: threads 128.. do the squares
            const int cc = cidx & 127, w0 = WIDE ? 0 : (cidx >> 7);
            const int col = n0 + cidx;
            const int quant0 = WIDE ? (t >> 7) : 0;
            // all reads first (independent), then the sums in a fixed order
            float va[NWV][16], vb[WIDE ? 1 : NWV][16];
#pragma unroll
            for (int w = 0; w < NWV; ++w) {
                const int wv = WIDE ? w : w * 2 + w0;
#pragma unroll
                for (int f = 0; f < 16; ++f) {
                    va[w][f] = red[(quant0 * 64 + wv * 16 + f) * RW + cc];
                    if constexpr (!WIDE) vb[w][f] = red[(64 + wv * 16 + f) * RW + cc];
                }
            }
            float a0 = 0.f, a1 = 0.f;
#pragma unroll
            for (int w = 0; w < NWV; ++w)
#pragma unroll
                for (int f = 0; f < 16; ++f) {
                    a0 += va[w][f];
                    if constexpr (!WIDE) a1 += vb[w][f];
                }
            if (col < p.N) {
                float* dst = p.gn_part + ((size_t)tm * p.N + col) * 2;
                if (WIDE) dst[quant0] = a0;
                else { dst[0] = a0; dst[1] = a1; }
            }
        }
        conv_trace_end();
        return;
    }
#ifdef FLUX2MI_DEV
    auto epi_trace_end = [&](int tag = 0) GS_INLINE {
        if (batch.trace) {   // dev diagnostic: the epilogue, loop end -> everything issued -> every store retired (shader cycles)
            unsigned long long c2, c3;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c2)::"memory");
            asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c3)::"memory");
            if (lane == 0) batch.trace[((size_t)bid * 4 + wid) * 8 + 3] = ((c2 - tr_c1) & 0xffffffffull) | (((c3 - c2) & 0xfffffffull) << 32) | ((unsigned long long)tag << 60) | (1ull << 63);
        }
    };
#else
    auto epi_trace_end = [&](int = 0) GS_INLINE {};
#endif
    int epi = p.epi;
    if (epi == EPI_QKV_MLP) epi = (n0 >= p.n_split) ? EPI_SWIGLU : EPI_BF16;
    if (!GRT && !BT && p.qk_qw != nullptr && epi == EPI_BF16 && n0 < p.qk_cols) {
        // ---- fused QK RMSNorm + RoPE (Flux2Attention.swift:137-157).  A 256-column tile is two heads and a wave holds all
        // 128 columns of one: the sum of squares is the lane's 32 values + the 4 fq lanes of a row (two lane exchanges).
        int fqo = fq;
        asm volatile("" : "+v"(fqo));   // values derived from it belong to this path, not to the registers live across the K loop
        const bool is_k = n0 >= (p.qk_cols >> 1);
        const uint16_t* wv = (is_k ? p.qk_kw : p.qk_qw) + fqo * 4;
        const float qm = is_k ? 1.0f : p.qk_qmul;
        // the lane's columns inside the head: j*16 + fqo*4 + r = axis j >> 1, pair (j & 1)*8 + fqo*2 + (r >> 1) of that axis:
        // four frequencies per lane, in revolutions (v_sin_f32 / v_cos_f32 take x / 2pi)
        float wrev[2][2];
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
            for (int pr = 0; pr < 2; ++pr)
                wrev[jb][pr] = __builtin_amdgcn_exp2f(-(float)(jb * 8 + fqo * 2 + pr) * p.qk_l2t16) * 0.15915494309189535f;
        const int wcol = n0 + wn * 128 + (fqo & 1) * 16 + (fqo >> 1) * 8;
        int rbase = row_base;
        asm volatile("" : "+v"(rbase));   // the rows of this path are its own values: shared with the scale loads above, hipcc keeps them all live (spills)
        // one address form for both outputs (bf16 columns of C, or the e4m3 q | k rows of the attention): base + row * stride + jp * step
        const bool e4 = p.qk8 != nullptr;
        uint8_t* const obase = e4 ? p.qk8 + (size_t)p.qk_row0 * p.qk8_ld + wcol : reinterpret_cast<uint8_t*>(p.C) + (size_t)wcol * 2;
        const unsigned ostride = e4 ? (unsigned)p.qk8_ld : (unsigned)p.ldc * 2u, ostep = e4 ? 32u : 64u;
        const float m8 = is_k ? p.qk8_kmul : p.qk8_qmul;
        // the norm weights of the lane's columns (8 x 8 bytes) and the positions of its rows, ONCE, in front of the rows: read where they are
        // used (a weight pair per (row, block): 64 loads) every read got a vmcnt(0..2) of its own, which on this hardware also waits for the
        // stores issued so far (one in-order counter) — the rows went out one store latency apart: 32-33k cycles per tile, a quarter of a
        // q | k tile (in-kernel stamps, profiles/r05_e_dit_epilogue_trace_rope.txt)
        const int qM = p.M, qrow0 = p.qk_row0;
        uint2 uwv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) uwv[j] = *reinterpret_cast<const uint2*>(wv + j * 16);
        float4 psv[4];   // the positions of four rows at a time (all NI: spills): the second batch's wait drains the first rows' stores, once
        auto pos_load = [&](int i0) GS_INLINE {
#pragma unroll
            for (int ii = 0; ii < 4; ++ii)
                if (i0 + ii < NI) psv[ii] = *reinterpret_cast<const float4*>(p.qk_pos + (size_t)(unsigned)(qrow0 + min(rbase + (i0 + ii) * 16, qM - 1)) * 4);
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) asm volatile("" : "+v"(psv[ii].x), "+v"(psv[ii].y), "+v"(psv[ii].z), "+v"(psv[ii].w));
        };
        pos_load(0);
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(uwv[j].x), "+v"(uwv[j].y));
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int row = rbase + i * 16;
            uint8_t* const orow = obase + (size_t)((unsigned)row * ostride);
            if (i == 4) pos_load(4);
            const float4 ps = psv[i & 3];
            float ssq = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) ssq += accv(i, j)[r] * accv(i, j)[r];
            ssq += __shfl_xor(ssq, 16);
            ssq += __shfl_xor(ssq, 32);
            const float rinv = rsqrtf(ssq * (1.0f / 128.0f) + 1e-6f);
            auto rotated = [&](int j, float& r0, float& r1, float& r2, float& r3) GS_INLINE {
                const uint2 uw = uwv[j];
                const float w0 = lo_bf16(uw.x) * qm, w1 = hi_bf16(uw.x) * qm, w2 = lo_bf16(uw.y) * qm, w3 = hi_bf16(uw.y) * qm;
                const float pos = (j >> 1) == 0 ? ps.x : (j >> 1) == 1 ? ps.y : (j >> 1) == 2 ? ps.z : ps.w;
                const float f0 = __builtin_amdgcn_fractf(pos * wrev[j & 1][0]), f1 = __builtin_amdgcn_fractf(pos * wrev[j & 1][1]);
                const float c0 = __builtin_amdgcn_cosf(f0), s0 = __builtin_amdgcn_sinf(f0);
                const float c1 = __builtin_amdgcn_cosf(f1), s1 = __builtin_amdgcn_sinf(f1);
                const float a0 = accv(i, j)[0] * rinv * w0, a1 = accv(i, j)[1] * rinv * w1;
                const float a2 = accv(i, j)[2] * rinv * w2, a3 = accv(i, j)[3] * rinv * w3;
                r0 = a0 * c0 - a1 * s0, r1 = a1 * c0 + a0 * s0, r2 = a2 * c1 - a3 * s1, r3 = a3 * c1 + a2 * s1;
            };
#pragma unroll
            for (int jp = 0; jp < 4; ++jp) {
                float r0, r1, r2, r3, t0, t1, t2, t3;
                rotated(2 * jp, r0, r1, r2, r3);
                rotated(2 * jp + 1, t0, t1, t2, t3);
                if (e4) {   // 8-byte runs after the same lane exchange; the bf16 columns are not needed then
                    uint32_t e0 = pack_fp8x4(r0 * m8, r1 * m8, r2 * m8, r3 * m8), e1 = pack_fp8x4(t0 * m8, t1 * m8, t2 * m8, t3 * m8);
                    s_swap16(e0, e1);
                    if (row < qM) *reinterpret_cast<uint2*>(orow + jp * ostep) = make_uint2(e0, e1);
                } else {
                    uint32_t a0 = gs_pack(r0, r1), a1 = gs_pack(r2, r3), b0 = gs_pack(t0, t1), b1 = gs_pack(t2, t3);
                    s_swap16(a0, b0);
                    s_swap16(a1, b1);
                    if (row < qM) *reinterpret_cast<uint4*>(orow + jp * ostep) = make_uint4(a0, a1, b0, b1);
                }
            }
        }
        epi_trace_end(1);   // tag 1: the fused QK norm + RoPE path
        return;
    }
    // ---- SwiGLU hidden as block-scaled e4m3 (GemmProblem::c2_q8): a tile's 128 hidden columns x the 16 rows of an MFMA row block are
    // one block.  First pass: a bound of the block maxima — |silu(g) u| <= |g u|, no transcendental (evaluating SiLU twice cost 23 us
    // per linear1 launch); the bound is tight where the maximum sits (large positive gates) and a power-of-two scale that is one step
    // low changes no e4m3 value above the subnormals — (the lane's 16 products per row block, the wave by lane exchange, the two waves
    // of a row through LDS), exponents, scale bytes; the stores below compute the values and write bytes.
    float bs_mult[NI];
    const bool q8_out = !GRT && !BT && FP8 && !CONV && epi == EPI_SWIGLU && p.c2_q8 != nullptr;
    if constexpr (FP8 && !CONV) {
        if (q8_out) {
            // |g u| = |acc_g acc_u| x (column scale of g x column scale of u) x (row scale)^2: the column products once per lane
            f32x4 cgu[2][2];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) cgu[hh][jj] = fs_col[4 * hh + jj] * fs_col[4 * hh + jj + 2];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                float mx = 0.f;
#pragma unroll
                for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const f32x4 pr = acc[i][4 * hh + jj] * acc[i][4 * hh + jj + 2] * cgu[hh][jj];
#pragma unroll
                        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, fabsf(pr[r]));
                    }
                mx *= fs_row[i] * fs_row[i] * 1.0000005f;   // (a hair up: the factors are regrouped against the values' own rounding)
                // both waves of the row, all lanes: one LDS maximum per row block (non-negative floats order like their bit patterns;
                // a chain of six lane exchanges per row block cost 3 us per tile)
                mx = row16_max_nonneg(mx);   // 4 lanes per wave go to the LDS instead of 64
                if ((lane & 15) == 15) atomicMax(&bsred[wm][i], __float_as_uint(mx));
            }
            __syncthreads();
            const int start = (p.epi == EPI_QKV_MLP) ? p.n_split : 0;
            uint8_t* bsp = p.c2_bs + (size_t)(((n0 - start) >> 1) >> 7) * p.c2_bs_ld + ((m0 + wm * WROWS) >> 4);
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int e = fp8_block_exp(__uint_as_float(bsred[wm][i]));
                bs_mult[i] = ldexpf(1.0f, e);
                if (wn == 0 && lane == 0 && m0 + wm * WROWS + i * 16 < p.M) bsp[i] = (uint8_t)(127 - e);
            }
        }
    }
    // ---- EPI_GATE_RESID, whole column tiles on aligned rows (every production launch): x += gate * y in the TRANSPOSED domain.
    // A lane of the accumulator layout owns one row of a 16-row block, so its 16-byte read-modify-write is 64 separate accesses per
    // instruction, which a CU's vector-memory path takes at 16 B/clk whatever the addresses (tools/store_pattern.hip: 15.98 B/clk against
    // 42-63 for lanes that cover whole rows) — and row by row every access paid its own latency: 70-80k cycles per tile in the first
    // form, 42 % of a proj launch and 15 % of linear2 (profiles/r05_c_dit_epilogue_trace.txt).  Here every (row block, 64-column half) goes
    // through a private LDS tile (4 ds_write_b128 in the accumulator layout, 4 ds_read_b128 transposed, software-pipelined over the row
    // blocks) to lanes that own 4 consecutive columns (lane & 15) of rows (lane >> 4) + 4 t: a load / store instruction is 4 whole
    // 256-byte runs.  The residual of a batch of row blocks is requested before the batch's first value is used.  Same products and
    // sums per element as before: bit-identical results.
    if constexpr (GRT) {
        {
            unsigned long long qC = reinterpret_cast<unsigned long long>(p.C), qV = reinterpret_cast<unsigned long long>(p.vec);
            uint32_t qN = (uint32_t)p.N, qM = (uint32_t)p.M, qldc = (uint32_t)p.ldc;
            asm volatile("" : "+s"(qC), "+s"(qV), "+s"(qN), "+s"(qM), "+s"(qldc));
            {
                typedef __attribute__((address_space(1))) uint8_t g_u8;
                typedef __attribute__((address_space(1))) float g_f32;
                typedef __attribute__((address_space(1))) f32x4 g_f4;
                g_u8* const Cq = (g_u8*)qC;
                const g_f32* const Vq = (const g_f32*)qV;
                // every per-lane value of this path from opaque copies made HERE: derived from `lane` / `wid` directly hipcc computes them in
                // front of the K loop and carries them through it (spills inside the loop: every VGPR there is spoken for)
                int ln = lane, wd = wid;
                asm volatile("" : "+v"(ln), "+v"(wd));
                const int tr = ln >> 4, tc = ln & 15, wm_ = WIDE ? wd : wd >> 1, wn_ = WIDE ? 0 : wd & 1;
                constexpr int TP = 68;                                                       // tile pitch in floats (272 B)
                float* const tile = reinterpret_cast<float*>(smem + wd * 8704);             // 2 x [16][68] f32 per wave
                float* const wr = tile + (ln & 15) * TP + (ln >> 4) * 4;                     // + jj * 16: block jj of the half
                const float* const rd = tile + tr * TP + tc * 4;                             // + t * 4 * TP: rows tr + 4 t
                const uint32_t rowt = (uint32_t)(m0 + wm_ * WROWS + tr);
                const uint32_t lim = rowt < qM ? qM - rowt : 0u;                             // row tr + k of the wave is valid while k < lim
                const uint32_t ldcB = qldc * 4u;                                             // bytes per row
                const uint32_t off0 = (rowt * qldc + (uint32_t)(n0 + wn_ * 128 + tc * 4)) * 4u;
                __syncthreads();   // LDS holds the last K-tile until every wave has left the K loop
                constexpr int RB = FP8 ? 2 : 4;                                              // row blocks per residual batch (the e4m3 form keeps its scales in registers)
                if constexpr (!FP8)   // (the e4m3 form, whose de-quantisation scales fill the registers, keeps the one-batch-at-a-time form below)
                if ((uint32_t)(m0 + BMT) <= qM) {
                    // ---- the tile lies inside M (all but the last row of tiles): no per-row predicates, so hipcc's waits are COUNTED and the
                    // residual of batch s + 1 is in flight while batch s is worked on (two register sets; batches run through both halves)
                    constexpr int NB = (NI + RB - 1) / RB, NS = 2 * NB;                        // batches per half, in all
                    f32x4 rv[2][RB][4];
                    f32x4 g4[2];
                    g4[0] = *(const g_f4*)(Vq + (n0 + wn_ * 128 + tc * 4));
                    g4[1] = *(const g_f4*)(Vq + (n0 + wn_ * 128 + 64 + tc * 4));
                    auto issue = [&](int sq) GS_INLINE {
                        const int h = sq / NB, i0 = (sq % NB) * RB;
#pragma unroll
                        for (int ii = 0; ii < RB; ++ii)
#pragma unroll
                            for (int tt = 0; tt < 4; ++tt)
                                if (i0 + ii < NI)
                                    rv[sq & 1][ii][tt] = *(const g_f4*)(Cq + (off0 + (uint32_t)((i0 + ii) * 16 + tt * 4) * ldcB + (uint32_t)h * 256u));
                    };
                    issue(0);
                    if (NS > 1) issue(1);
                    f32x4 cur[4], nxt[4];
#pragma unroll
                    for (int sq = 0; sq < NS; ++sq) {
                        const int h = sq / NB, i0 = (sq % NB) * RB;
                        auto tile_w = [&](int i) GS_INLINE {
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj) *reinterpret_cast<f32x4*>(wr + (i & 1) * (16 * TP) + jj * 16) = accv(i, 4 * h + jj);
                        };
                        auto tile_r = [&](int i, f32x4 (&v)[4]) GS_INLINE {
#pragma unroll
                            for (int tt = 0; tt < 4; ++tt) v[tt] = *reinterpret_cast<const f32x4*>(rd + (i & 1) * (16 * TP) + tt * 4 * TP);
                        };
                        if (i0 == 0) {   // a half starts: its first two row blocks into the LDS tiles
                            tile_w(0);
                            if (NI > 1) tile_w(1);
                            tile_r(0, cur);
                        }
#pragma unroll
                        for (int ii = 0; ii < RB; ++ii) asm volatile("" : "+v"(rv[sq & 1][ii][0]), "+v"(rv[sq & 1][ii][1]), "+v"(rv[sq & 1][ii][2]), "+v"(rv[sq & 1][ii][3]));
#pragma unroll
                        for (int ii = 0; ii < RB; ++ii) {
                            const int i = i0 + ii;
                            if (i < NI) {
                                if (i + 1 < NI) tile_r(i + 1, nxt);
                                if (i + 2 < NI) tile_w(i + 2);        // behind block i's reads in the LDS queue: its tile is free
#pragma unroll
                                for (int tt = 0; tt < 4; ++tt)
                                    *(g_f4*)(Cq + (off0 + (uint32_t)(i * 16 + tt * 4) * ldcB + (uint32_t)h * 256u)) = rv[sq & 1][ii][tt] + g4[h] * cur[tt];
#pragma unroll
                                for (int tt = 0; tt < 4; ++tt) cur[tt] = nxt[tt];
                            }
                        }
                        if (sq + 2 < NS) issue(sq + 2);   // into the set just consumed
                    }
                    epi_trace_end();
                    return;
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f32x4 g4 = *(const g_f4*)(Vq + (n0 + wn_ * 128 + h * 64 + tc * 4));
                    auto tile_w = [&](int i) GS_INLINE {
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) *reinterpret_cast<f32x4*>(wr + (i & 1) * (16 * TP) + jj * 16) = accv(i, 4 * h + jj);
                    };
                    auto tile_r = [&](int i, f32x4 (&v)[4]) GS_INLINE {
#pragma unroll
                        for (int tt = 0; tt < 4; ++tt) v[tt] = *reinterpret_cast<const f32x4*>(rd + (i & 1) * (16 * TP) + tt * 4 * TP);
                    };
                    f32x4 cur[4], nxt[4];
                    tile_w(0);
                    if (NI > 1) tile_w(1);
                    tile_r(0, cur);
                    f32x4 rv[RB][4];
#pragma unroll
                    for (int i = 0; i < NI; ++i) {
                        if (i % RB == 0) {   // the residual of the next RB row blocks: requested together, one wait (unconditional pins: see the convolution's epilogue)
#pragma unroll
                            for (int ii = 0; ii < RB; ++ii)
#pragma unroll
                                for (int tt = 0; tt < 4; ++tt) {
                                    const uint32_t k = (uint32_t)((i + ii) * 16 + tt * 4);
                                    rv[ii][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
                                    if (i + ii < NI && k < lim) rv[ii][tt] = *(const g_f4*)(Cq + (off0 + k * ldcB + (uint32_t)h * 256u));
                                }
                            asm volatile("" ::: "memory");
#pragma unroll
                            for (int ii = 0; ii < RB; ++ii) asm volatile("" : "+v"(rv[ii][0]), "+v"(rv[ii][1]), "+v"(rv[ii][2]), "+v"(rv[ii][3]));
                        }
                        if (i + 1 < NI) tile_r(i + 1, nxt);
                        if (i + 2 < NI) tile_w(i + 2);        // behind block i's reads in the LDS queue: its tile is free
#pragma unroll
                        for (int tt = 0; tt < 4; ++tt) {
                            const uint32_t k = (uint32_t)(i * 16 + tt * 4);
                            const f32x4 v = rv[i % RB][tt] + g4 * cur[tt];
                            if (k < lim) *(g_f4*)(Cq + (off0 + k * ldcB + (uint32_t)h * 256u)) = v;
                        }
#pragma unroll
                        for (int tt = 0; tt < 4; ++tt) cur[tt] = nxt[tt];
                    }
                }
                epi_trace_end();
                return;
            }
        }
    }
    // ---- EPI_BF16 on whole, aligned column tiles (the V^T launches), in the transposed domain: C bf16 = alpha * acc, optionally key-permuted
    // inside every 16-column group (col_perm16: applied to the LDS column of the accumulator-layout write).  Lane l owns 8 consecutive
    // columns (l & 7) of rows (l >> 3) and (l >> 3) + 8 of a 16-row block: a store instruction is 8 whole 128-byte runs.
    if constexpr (BT) {
        unsigned long long qC = reinterpret_cast<unsigned long long>(p.C);
        uint32_t qM = (uint32_t)p.M, qldc = (uint32_t)p.ldc, qperm = (uint32_t)p.col_perm16;
        float alpha = p.alpha;
        asm volatile("" : "+s"(qC), "+s"(qM), "+s"(qldc), "+s"(qperm), "+s"(alpha));
        typedef __attribute__((address_space(1))) uint8_t g_u8;
        typedef __attribute__((address_space(1))) u32x4 g_u4;
        g_u8* const Cq = (g_u8*)qC;
        int ln = lane, wd = wid;
        asm volatile("" : "+v"(ln), "+v"(wd));   // per-lane values of this path from opaque copies made HERE (see the gate-residual form)
        const int tr = ln >> 3, tc = ln & 7, wm_ = wd >> 1, wn_ = wd & 1, fq_ = ln >> 4;
        constexpr int TP = 68;                                                       // tile pitch in floats (272 B)
        float* const tile = reinterpret_cast<float*>(smem + wd * 8704);             // 2 x [16][68] f32 per wave
        float* const wr = tile + (ln & 15) * TP + (qperm ? (fq_ & 1) * 8 + (fq_ >> 1) * 4 : fq_ * 4);   // + jj * 16: block jj of the half
        const float* const rd = tile + tr * TP + tc * 8;                             // + tt * 8 * TP: rows tr, tr + 8
        const uint32_t rowt = (uint32_t)(m0 + wm_ * WROWS + tr);
        const uint32_t lim = rowt < qM ? qM - rowt : 0u;                             // row tr + k of the wave is valid while k < lim
        const uint32_t ldcB = qldc * 2u;                                             // bytes per row
        const uint32_t off0 = (rowt * qldc + (uint32_t)(n0 + wn_ * 128 + tc * 8)) * 2u;
        __syncthreads();   // LDS holds the last K-tile until every wave has left the K loop
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            auto tile_w = [&](int i) GS_INLINE {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) *reinterpret_cast<f32x4*>(wr + (i & 1) * (16 * TP) + jj * 16) = accv(i, 4 * h + jj);
            };
            auto tile_r = [&](int i, f32x4 (&v)[4]) GS_INLINE {
                const float* r0 = rd + (i & 1) * (16 * TP);
                v[0] = *reinterpret_cast<const f32x4*>(r0); v[1] = *reinterpret_cast<const f32x4*>(r0 + 4);
                v[2] = *reinterpret_cast<const f32x4*>(r0 + 8 * TP); v[3] = *reinterpret_cast<const f32x4*>(r0 + 8 * TP + 4);
            };
            f32x4 cur[4], nxt[4];
            tile_w(0);
            if (NI > 1) tile_w(1);
            tile_r(0, cur);
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                if (i + 1 < NI) tile_r(i + 1, nxt);
                if (i + 2 < NI) tile_w(i + 2);        // behind block i's reads in the LDS queue: its tile is free
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const uint32_t k = (uint32_t)(i * 16 + tt * 8);
                    const f32x4 va = cur[2 * tt] * alpha, vb = cur[2 * tt + 1] * alpha;
                    if (k < lim)
                        *(g_u4*)(Cq + (off0 + k * ldcB + (uint32_t)h * 128u)) =
                            u32x4{pack_bf16x2(va[0], va[1]), pack_bf16x2(va[2], va[3]), pack_bf16x2(vb[0], vb[1]), pack_bf16x2(vb[2], vb[3])};
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) cur[k] = nxt[k];
            }
        }
        epi_trace_end();
        return;
    }
    if constexpr (!GRT && !BT)
    for_seq([&](auto hc) GS_INLINE {
        constexpr int h = decltype(hc)::value;
        (void)acc; (void)bs_mult;
        const int wnv = wn * 2 + h;               // the 64-column group of this half
        const int col_base = n0 + wnv * 64 + fq * 4;
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
                            *reinterpret_cast<f32x4*>(dst) = accv(i, 4 * h + j) * p.alpha;
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (col + r < p.N) dst[r] = accv(i, 4 * h + j)[r] * p.alpha;
                        }
                    }
                }
            }
        } else if (epi == EPI_BF16) {
            uint16_t* C = reinterpret_cast<uint16_t*>(p.C);
            const bool vec_ok = (p.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(C) & 7) == 0;
            if ((batch.flags & 1) && (p.ldc & 7) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0 && n0 + TN <= p.N) {
                // 16-byte stores: half the store instructions of the 8-byte form below.  col_perm16: the four-column
                // groups of a 16-group go fq 0,2,1,3, so the 16-byte runs pair fq with fq^2 (lanes l, l^32): after the
                // exchange lanes fq 0,1 hold block 2jp at (fq&1)*8 and lanes fq 2,3 block 2jp+1
                // the fields once, as opaque scalars: read at their uses hipcc re-loads them from the kernel-argument segment every time
                // (it treats such loads as free to rematerialise): 32 exposed scalar-load waits per half in the V^T launches
                float alpha = p.alpha;
                int perm = p.col_perm16, qM = p.M, qldc = p.ldc;
                asm volatile("" : "+s"(alpha), "+s"(perm), "+s"(qM), "+s"(qldc));
                const int wcol = perm ? n0 + wnv * 64 + (fq >> 1) * 16 + (fq & 1) * 8
                                      : n0 + wnv * 64 + (fq & 1) * 16 + (fq >> 1) * 8;
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int row = row_base + i * 16;
#pragma unroll
                    for (int jp = 0; jp < 2; ++jp) {
                        const f32x4 vx = accv(i, 4 * h + 2 * jp) * alpha, vy = accv(i, 4 * h + 2 * jp + 1) * alpha;
                        uint32_t x0 = gs_pack(vx[0], vx[1]), x1 = gs_pack(vx[2], vx[3]);
                        uint32_t y0 = gs_pack(vy[0], vy[1]), y1 = gs_pack(vy[2], vy[3]);
                        if (perm) {
                            s_swap32(x0, y0);
                            s_swap32(x1, y1);
                        } else {
                            s_swap16(x0, y0);
                            s_swap16(x1, y1);
                        }
                        if (row < qM)
                            *reinterpret_cast<uint4*>(C + (size_t)row * qldc + wcol + jp * 32) = make_uint4(x0, x1, y0, y1);
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
                        const int scol = p.col_perm16 ? n0 + wnv * 64 + j * 16 + ((fq & 1) * 2 + (fq >> 1)) * 4 : col;
                        uint16_t* dst = C + (size_t)row * p.ldc + scol;
                        const f32x4 v = accv(i, 4 * h + j) * p.alpha;
                        if (vec_ok && col + 3 < p.N) {
                            *reinterpret_cast<uint2*>(dst) = make_uint2(gs_pack(v[0], v[1]), gs_pack(v[2], v[3]));
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
                            *d4 = *d4 + g[j] * accv(i, 4 * h + j);
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (col + r < p.N) dst[r] = dst[r] + g[j][r] * accv(i, 4 * h + j)[r];
                        }
                    }
                }
            }
        } else if (epi == EPI_SWIGLU) {
            // weight rows are interleaved in blocks of 32 (gate | value) inside each 64-column group:
            // gate = accumulator blocks 4h + 0,1, value = blocks 4h + 2,3 of the same lane
            uint16_t* C = reinterpret_cast<uint16_t*>(p.C2);
            const int start = (p.epi == EPI_QKV_MLP) ? p.n_split : 0;
            const int hbase = ((n0 + wnv * 64 - start) >> 1) + fq * 4;
            const int hidden = (p.N - start) >> 1;
            const bool vec_ok = (p.ldc2 & 3) == 0 && (reinterpret_cast<uintptr_t>(C) & 7) == 0;
            if (q8_out) {   // 8 consecutive hidden bytes per lane after the same lane exchange (the launcher guarantees full tiles)
                const int wh = ((n0 + wnv * 64 - start) >> 1) + (fq & 1) * 16 + (fq >> 1) * 8;
                // the fields once, as opaque scalars (read at their uses hipcc re-loads them from the kernel-argument segment per store)
                unsigned long long q8p = reinterpret_cast<unsigned long long>(p.c2_q8);
                int q8ld = p.c2_q8_ld, q8M = p.M;
                asm volatile("" : "+s"(q8p), "+s"(q8ld), "+s"(q8M));
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int row = row_base + i * 16;
                    float ox[4], oy[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        ox[r] = silu_f(accv(i, 4 * h + 0)[r]) * accv(i, 4 * h + 2)[r] * bs_mult[i];
                        oy[r] = silu_f(accv(i, 4 * h + 1)[r]) * accv(i, 4 * h + 3)[r] * bs_mult[i];
                    }
                    uint32_t e0 = pack_fp8x4_inrange(ox[0], ox[1], ox[2], ox[3]), e1 = pack_fp8x4_inrange(oy[0], oy[1], oy[2], oy[3]);
                    s_swap16(e0, e1);
                    if (row < q8M) *reinterpret_cast<uint2*>(reinterpret_cast<uint8_t*>(q8p) + (size_t)row * q8ld + wh) = make_uint2(e0, e1);
                }
            } else
            if ((batch.flags & 1) && (p.ldc2 & 7) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0 && n0 + TN <= p.N) {
                const int wh = ((n0 + wnv * 64 - start) >> 1) + (fq & 1) * 16 + (fq >> 1) * 8;
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int row = row_base + i * 16;
                    float ox[4], oy[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        ox[r] = silu_f(accv(i, 4 * h + 0)[r]) * accv(i, 4 * h + 2)[r];
                        oy[r] = silu_f(accv(i, 4 * h + 1)[r]) * accv(i, 4 * h + 3)[r];
                    }
                    uint32_t x0 = gs_pack(ox[0], ox[1]), x1 = gs_pack(ox[2], ox[3]);
                    uint32_t y0 = gs_pack(oy[0], oy[1]), y1 = gs_pack(oy[2], oy[3]);
                    s_swap16(x0, y0);
                    s_swap16(x1, y1);
                    if (row < p.M) *reinterpret_cast<uint4*>(C + (size_t)row * p.ldc2 + wh) = make_uint4(x0, x1, y0, y1);
                }
            } else
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int row = row_base + i * 16;
                if (row < p.M) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int hh = hbase + j * 16;
                        uint16_t* dst = C + (size_t)row * p.ldc2 + hh;
                        float o[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = silu_f(accv(i, 4 * h + j)[r]) * accv(i, 4 * h + j + 2)[r];
                        if (vec_ok && hh + 3 < hidden) {
                            *reinterpret_cast<uint2*>(dst) = make_uint2(gs_pack(o[0], o[1]), gs_pack(o[2], o[3]));
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (hh + r < hidden) dst[r] = f32_to_bf16_bits(o[r]);
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
                        f32x4 v = accv(i, 4 * h + j) * p.alpha + bsv[j];
                        if (vec_ok && col + 3 < p.N) {
                            if (p.resid) {
                                const uint2 rv = *reinterpret_cast<const uint2*>(p.resid + (size_t)row * p.ldr + col);
                                v[0] += bf16_bits_to_f32((uint16_t)(rv.x & 0xffff)); v[1] += bf16_bits_to_f32((uint16_t)(rv.x >> 16));
                                v[2] += bf16_bits_to_f32((uint16_t)(rv.y & 0xffff)); v[3] += bf16_bits_to_f32((uint16_t)(rv.y >> 16));
                            }
                            *reinterpret_cast<uint2*>(dst) = make_uint2(gs_pack(v[0], v[1]), gs_pack(v[2], v[3]));
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
    }, std::make_integer_sequence<int, 2>{});
    epi_trace_end();
}

}  // namespace

// Tile height by wave quantisation (256x256 kernels): rounds of 256 workgroups x relative tile cost (+2 % per row block
// dropped for the lower arithmetic intensity); ties keep the taller tile.  FLUX2MI_GEMM_NI=8 forces 256 rows (dev A/B runs).
int gemm256_pick_ni(const GemmProblem* probs, int nprob) {
    static const int ni_force = [] {
        const char* e = F2_DEV_ENV("FLUX2MI_GEMM_NI");
        return e ? atoi(e) : 0;
    }();
    if (ni_force >= 5 && ni_force <= 8) return ni_force;
    int ni = 8;
    double best = 1e30;
    for (int c = 8; c >= 5; --c) {
        int tot = 0;
        for (int i = 0; i < nprob; ++i) tot += ((probs[i].M + 32 * c - 1) / (32 * c)) * ((probs[i].N + 255) / 256);
        const double cost = (double)((tot + 255) / 256) * (c / 8.0) * (1.0 + 0.02 * (8 - c));
        if (cost < best * 0.97) { best = cost; ni = c; }   // a shorter tile must win by 3 %
    }
    return ni;
}


// true when this kernel takes the launch: bf16 or e4m3 operands in plain row-major rows, 32-bit source offsets
bool gemm256s_eligible(const GemmProblem* probs, int nprob) {
    for (int i = 0; i < nprob; ++i) {
        const GemmProblem& q = probs[i];
        if (q.K % (q.fp8 ? 128 : 64) != 0 || (q.fp8 != 0) != (probs[0].fp8 != 0)) return false;
        if ((unsigned long long)q.M * (unsigned long long)q.lda * 2 >= (1ull << 32) || (unsigned long long)q.N * (unsigned long long)q.ldw * 2 >= (1ull << 32)) return false;
    }
    return true;
}

// GEMM rows per tile the launcher will use for these EPI_CONV problems (= the row count of GemmProblem::gn_part is ceil(M / this))
int gemm256s_conv_tile_rows(const GemmProblem* probs, int nprob) {
    if (nprob == 1 && probs[0].N <= 128) return 512;
    return 32 * gemm256_pick_ni(probs, nprob);
}

// returns 0 ok / 1 unsupported problem / 2 launch error; the caller has validated the problems
int launch_gemm256s(hipStream_t stream, const GemmProblem* probs, int nprob) {
    GemmBatch b;
    b.nprob = nprob;
    int ni = gemm256_pick_ni(probs, nprob);
    // convolutions with <= 128 output channels: the 512 x 128 tile (a 256-column tile would idle half of every MFMA)
    const bool wide = probs[0].epi == EPI_CONV && nprob == 1 && probs[0].N <= 128;
    if (wide) ni = 8;
    if (nprob < 1 || nprob > GEMM_MAX_PROBS) return 1;
    int tiles[GEMM_MAX_PROBS] = {};
    int total_tiles = 0;
    for (int i = 0; i < nprob; ++i) {
        b.p[i] = probs[i];
        const GemmProblem& q = probs[i];   // fused QK norm + RoPE: whole heads per tile, 16-byte row stores
        if (q.qk_qw && (!q.qk_kw || !q.qk_pos || q.qk_cols % 512 != 0 || q.qk_cols > q.N || (q.N & 255) || (q.ldc & 7) ||
                        (reinterpret_cast<uintptr_t>(q.C) & 15) || (q.epi != EPI_BF16 && q.epi != EPI_QKV_MLP) ||
                        (q.epi == EPI_QKV_MLP && q.qk_cols > q.n_split) ||
                        // the fused epilogue's row offsets are 32-bit; its e4m3 form stores 8-byte runs
                        (unsigned long long)q.M * (unsigned long long)q.ldc * 2ull >= (1ull << 32) ||
                        (q.qk8 && ((q.qk8_ld & 7) || (reinterpret_cast<uintptr_t>(q.qk8) & 7) ||
                                   (unsigned long long)q.M * (unsigned long long)q.qk8_ld >= (1ull << 32)))))
            return 1;
        tiles[i] = wide ? ((q.M + 511) / 512) * ((q.N + 127) / 128) : ((q.M + 32 * ni - 1) / (32 * ni)) * ((q.N + TN - 1) / TN);
    }
    for (int i = nprob; i < GEMM_MAX_PROBS; ++i) b.p[i] = probs[0];
    for (int i = 0; i < GEMM_MAX_PROBS; ++i) {
        b.tile_start[i] = total_tiles;
        total_tiles += tiles[i];
    }
    b.flags = 1;   // 16-byte bf16 / SwiGLU epilogue stores
    b.trace = nullptr;
    const dim3 grid(total_tiles), block(256);
    static const char* trace_env = F2_DEV_ENV("FLUX2MI_GEMM_TRACE");
    static int seen = 0;
    const size_t tn = (size_t)grid.x * 32;   // 8 words per wave
    static const char* trace_every = F2_DEV_ENV("FLUX2MI_GEMM_TRACE_EVERY");   // trace every n-th launch (default 16)
    static const int trace_n = trace_every ? (atoi(trace_every) > 0 ? atoi(trace_every) : 1) : 16;
    if (trace_env && ++seen % trace_n == 0) {
        // zeroed: words 4..7 are written by the lean convolution epilogue only, word 3 by dev builds only
        if (hipMalloc((void**)&b.trace, tn * 8) != hipSuccess) b.trace = nullptr;
        else if (hipMemsetAsync(b.trace, 0, tn * 8, stream) != hipSuccess) { (void)hipFree(b.trace); b.trace = nullptr; }
    }
    struct TraceOut {
        GemmBatch& b; hipStream_t stream; size_t tn; const GemmProblem* probs; int ni; int nprob;
        ~TraceOut() {
            if (!b.trace) return;
            std::vector<unsigned long long> h(tn);
            (void)hipStreamSynchronize(stream);
            (void)hipMemcpy(h.data(), b.trace, tn * 8, hipMemcpyDeviceToHost);
            (void)hipFree(b.trace);
            double cyc = 0, rt = 0, kt = 0, epi_issue = 0, epi_drain = 0;
            double st[4] = {0, 0, 0, 0};
            for (size_t w = 0; w < tn / 8; ++w) {
                cyc += (double)h[w * 8]; rt += (double)h[w * 8 + 1]; kt += (double)h[w * 8 + 2] + 1.5;   // + prologue ~1.5 K-tiles
                epi_issue += (double)(h[w * 8 + 3] & 0xffffffffull); epi_drain += (double)(h[w * 8 + 3] >> 32);
                for (int k = 0; k < 4; ++k) st[k] += (double)h[w * 8 + 4 + k];
            }
            const double nw = (double)(tn / 8);
            if (probs[0].epi != EPI_CONV && epi_issue > 0) {
                for (int tag = 0; tag < 2; ++tag) {
                    double ei = 0, ed = 0, n = 0;
                    for (size_t w = 0; w < tn / 8; ++w)
                        if ((h[w * 8 + 3] >> 63) && (int)((h[w * 8 + 3] >> 60) & 7) == tag) { ei += (double)(h[w * 8 + 3] & 0xffffffffull); ed += (double)((h[w * 8 + 3] >> 32) & 0xfffffffull); n += 1; }
                    if (n > 0)
                        fprintf(stderr, "[gemm256s trace] epilogue (epi %d%s): %.0f cycles to issue, %.0f more until its stores have retired (mean of %.0f waves); prologue + K loop %.0f cycles per tile, %d K-tiles\n",
                                probs[0].epi, tag ? ", the QK norm + RoPE path" : "", ei / n, ed / n, n, cyc / nw, probs[0].K / (probs[0].fp8 ? 128 : 64));
                }
            }
            if (probs[0].epi == EPI_CONV) {
                fprintf(stderr, "[gemm256s trace] conv epilogue: %.0f cycles to issue, %.0f more until its stores have retired (per wave, mean of %zu); prologue + K loop %.0f cycles per tile, %d K-tiles\n",
                        epi_issue / nw, epi_drain / nw, tn / 8, cyc / nw, probs[0].K / 64);
                if (st[3] > 0)   // the lean epilogue's own stamps (cycles from the loop's end)
                    fprintf(stderr, "[gemm256s trace]   lean epilogue: loads issued at %.0f, first half starts (bias + residual landed) %.0f, ends %.0f, second half starts %.0f (from the loop's end)\n",
                            st[0] / nw, st[1] / nw, st[2] / nw, st[3] / nw);
            }
            // a K-tile is 16 NI MFMAs of 16 cycles (bf16, 64 deep) or 8 NI of 32 (e4m3, 128 deep): 256 NI cycles of MFMA issue either way
            fprintf(stderr, "[gemm256s trace] M %d N %d K %d%s%s, %d-row tiles: %.0f shader cycles per K-tile (%d = MFMA issue alone; prologue counted as 1.5), clock %.0f MHz, %.3f us per K-tile\n",
                    probs[0].M, probs[0].N, probs[0].K, probs[0].fp8 ? " fp8" : "", nprob > 1 ? " (+ a second problem)" : "", 32 * ni, cyc / kt, 256 * ni,
                    cyc / rt * 100.0, rt / kt * 0.01);
        }
    } trace_out{b, stream, tn, probs, ni, nprob};
    if (probs[0].epi == EPI_CONV) {
        for (int i = 0; i < nprob; ++i) {
            const GemmProblem& q = probs[i];
            if (q.epi != EPI_CONV || q.fp8 || q.conv_seg < 2 || q.conv_wp < 3 || q.conv_w != q.conv_wp - 2 || (q.N & 3) || (q.ldc & 3) ||
                (reinterpret_cast<uintptr_t>(q.C) & 7) || (q.resid && ((q.ldr & 3) || (reinterpret_cast<uintptr_t>(q.resid) & 7))) ||
                (q.K / 64 + q.conv_seg - 1) / q.conv_seg > 3)
                return 1;
        }
        // the lean epilogue: whole 16-byte runs on aligned rows, an image wider than 16 pixels (every production layer)
        bool fast = true;
        for (int i = 0; i < nprob; ++i) {
            const GemmProblem& q = probs[i];
            fast = fast && (q.N & 31) == 0 && (q.ldc & 7) == 0 && (reinterpret_cast<uintptr_t>(q.C) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(q.vec) & 15) == 0 && q.conv_wp > 16 && q.M < (1 << 24) && q.ldc < (1 << 24) && q.ldr < (1 << 24) &&
                   (!q.resid || ((q.ldr & 7) == 0 && (reinterpret_cast<uintptr_t>(q.resid) & 15) == 0)) &&
                   // 32-bit ELEMENT offsets into the output (up to 4 M + a padded row of slack rows for the upsampling form) and the residual
                   (4ull * (unsigned long long)q.M + 4ull * (unsigned long long)q.conv_wp + 64ull) *
                       (unsigned long long)(q.ldc > q.ldr ? q.ldc : q.ldr) < (1ull << 32);
        }
        static const bool fast_on = [] {
            const char* e = F2_DEV_ENV("FLUX2MI_CONV_FAST");
            return e ? atoi(e) != 0 : true;
        }();
        fast = fast && fast_on;
        if (fast) {
            if (wide) hipLaunchKernelGGL((gemm256s_kernel<8, false, true, true, true>), grid, block, 0, stream, b);
            else switch (ni) {
                case 7: hipLaunchKernelGGL((gemm256s_kernel<7, false, true, false, true>), grid, block, 0, stream, b); break;
                case 6: hipLaunchKernelGGL((gemm256s_kernel<6, false, true, false, true>), grid, block, 0, stream, b); break;
                case 5: hipLaunchKernelGGL((gemm256s_kernel<5, false, true, false, true>), grid, block, 0, stream, b); break;
                default: hipLaunchKernelGGL((gemm256s_kernel<8, false, true, false, true>), grid, block, 0, stream, b); break;
            }
        } else if (wide) hipLaunchKernelGGL((gemm256s_kernel<8, false, true, true>), grid, block, 0, stream, b);
        else switch (ni) {
            case 7: hipLaunchKernelGGL((gemm256s_kernel<7, false, true>), grid, block, 0, stream, b); break;
            case 6: hipLaunchKernelGGL((gemm256s_kernel<6, false, true>), grid, block, 0, stream, b); break;
            case 5: hipLaunchKernelGGL((gemm256s_kernel<5, false, true>), grid, block, 0, stream, b); break;
            default: hipLaunchKernelGGL((gemm256s_kernel<8, false, true>), grid, block, 0, stream, b); break;
        }
        return hipGetLastError() == hipSuccess ? 0 : 2;
    }
    // the transposed gate-residual epilogue: every problem x += gate * y on whole, aligned 256-column tiles (proj, mlp_out, linear2 of every model)
    bool grt = true;
    for (int i = 0; i < nprob; ++i) {
        const GemmProblem& q = probs[i];
        grt = grt && q.epi == EPI_GATE_RESID && (q.N % TN) == 0 && (q.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(q.C) & 15) == 0 &&
              q.vec != nullptr && (reinterpret_cast<uintptr_t>(q.vec) & 15) == 0 &&
              // the transposed epilogue addresses X by 32-bit BYTE offsets: larger problems take the generic (64-bit) epilogue
              (unsigned long long)q.M * (unsigned long long)q.ldc * 4ull < (1ull << 32);
    }
    static const bool grt_on = [] {
        const char* e = F2_DEV_ENV("FLUX2MI_GEMM_GRT");
        return e ? atoi(e) != 0 : true;
    }();
    grt = grt && grt_on;
    // the transposed bf16 store: every problem a plain EPI_BF16 (no QK norm + RoPE columns, no e4m3 copy) on whole, aligned 256-column tiles (V^T)
    bool bt = !grt;
    for (int i = 0; i < nprob; ++i) {
        const GemmProblem& q = probs[i];
        bt = bt && q.epi == EPI_BF16 && q.qk_qw == nullptr && q.qk8 == nullptr && (q.N % TN) == 0 && (q.ldc & 7) == 0 &&
             (reinterpret_cast<uintptr_t>(q.C) & 15) == 0 && (size_t)q.M * (size_t)q.ldc * 2 < (1ull << 32);
    }
    static const bool bt_on = [] {
        const char* e = F2_DEV_ENV("FLUX2MI_GEMM_BT");
        return e ? atoi(e) != 0 : true;
    }();
    bt = bt && bt_on;
    if (bt) {
        if (probs[0].fp8) switch (ni) {
            case 7: hipLaunchKernelGGL((gemm256s_kernel<7, true, false, false, false, false, true>), grid, block, 0, stream, b); break;
            case 6: hipLaunchKernelGGL((gemm256s_kernel<6, true, false, false, false, false, true>), grid, block, 0, stream, b); break;
            case 5: hipLaunchKernelGGL((gemm256s_kernel<5, true, false, false, false, false, true>), grid, block, 0, stream, b); break;
            default: hipLaunchKernelGGL((gemm256s_kernel<8, true, false, false, false, false, true>), grid, block, 0, stream, b); break;
        }
        else switch (ni) {
            case 7: hipLaunchKernelGGL((gemm256s_kernel<7, false, false, false, false, false, true>), grid, block, 0, stream, b); break;
            case 6: hipLaunchKernelGGL((gemm256s_kernel<6, false, false, false, false, false, true>), grid, block, 0, stream, b); break;
            case 5: hipLaunchKernelGGL((gemm256s_kernel<5, false, false, false, false, false, true>), grid, block, 0, stream, b); break;
            default: hipLaunchKernelGGL((gemm256s_kernel<8, false, false, false, false, false, true>), grid, block, 0, stream, b); break;
        }
        return hipGetLastError() == hipSuccess ? 0 : 2;
    }
    if (grt) {
        if (probs[0].fp8) switch (ni) {
            case 7: hipLaunchKernelGGL((gemm256s_kernel<7, true, false, false, false, true>), grid, block, 0, stream, b); break;
            case 6: hipLaunchKernelGGL((gemm256s_kernel<6, true, false, false, false, true>), grid, block, 0, stream, b); break;
            case 5: hipLaunchKernelGGL((gemm256s_kernel<5, true, false, false, false, true>), grid, block, 0, stream, b); break;
            default: hipLaunchKernelGGL((gemm256s_kernel<8, true, false, false, false, true>), grid, block, 0, stream, b); break;
        }
        else switch (ni) {
            case 7: hipLaunchKernelGGL((gemm256s_kernel<7, false, false, false, false, true>), grid, block, 0, stream, b); break;
            case 6: hipLaunchKernelGGL((gemm256s_kernel<6, false, false, false, false, true>), grid, block, 0, stream, b); break;
            case 5: hipLaunchKernelGGL((gemm256s_kernel<5, false, false, false, false, true>), grid, block, 0, stream, b); break;
            default: hipLaunchKernelGGL((gemm256s_kernel<8, false, false, false, false, true>), grid, block, 0, stream, b); break;
        }
        return hipGetLastError() == hipSuccess ? 0 : 2;
    }
    if (probs[0].fp8) {
        switch (ni) {
            case 7: hipLaunchKernelGGL((gemm256s_kernel<7, true>), grid, block, 0, stream, b); break;
            case 6: hipLaunchKernelGGL((gemm256s_kernel<6, true>), grid, block, 0, stream, b); break;
            case 5: hipLaunchKernelGGL((gemm256s_kernel<5, true>), grid, block, 0, stream, b); break;
            default: hipLaunchKernelGGL((gemm256s_kernel<8, true>), grid, block, 0, stream, b); break;
        }
        return hipGetLastError() == hipSuccess ? 0 : 2;
    }
    switch (ni) {
        case 7: hipLaunchKernelGGL((gemm256s_kernel<7>), grid, block, 0, stream, b); break;
        case 6: hipLaunchKernelGGL((gemm256s_kernel<6>), grid, block, 0, stream, b); break;
        case 5: hipLaunchKernelGGL((gemm256s_kernel<5>), grid, block, 0, stream, b); break;
        default: hipLaunchKernelGGL((gemm256s_kernel<8>), grid, block, 0, stream, b); break;
    }
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
