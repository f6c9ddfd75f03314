// attention64.hip — (dev library only: FLUX2MI_ATTN64=1; the product form is attention64b.hip on MFMA 16x16x32)
// flash-attention forward with ONE WAVE PER SIMD on MFMA 32x32x16: workgroup = 4 waves = 256 query rows, 64 rows per wave,
// KV tiles of 64 keys by LDS-DMA into 3-deep rings, the whole tile loop hand-placed (inline asm in issue order).
//
// Same operation, operand layouts and rounding points as attention.hip's attn_fwd_kernel<false, *, 2> (non-causal, the
// score scale folded into Q, softmax against a zero offset: p = exp2(s)); launch_attention uses it for those calls when the
// key count is a multiple of 64.  Why a second structure (MI355X guide, "one wave per SIMD" rows): with 32 query rows per
// wave every K / V^T fragment read from LDS feeds ONE 32x32x16 MFMA and the two waves of a SIMD meet in their softmax
// segments; with 64 rows per wave a fragment feeds two MFMAs, and one instruction stream per SIMD can hold the softmax of
// tile t+1 in the gaps of the PV MFMAs of tile t and of the QK^T MFMAs of tile t+2 (<= 5 fillers per 32-cycle gap, one
// of them a v_exp_f32).
//
// Per wave: O^T 2 x 4 accumulators (128 AGPRs), Q 2 x 8 fragments (64 AGPRs), the K fragments of a whole tile (64 AGPRs);
// scores S 2 x 2 (64 VGPRs), P 2 x 4 bf16x8 (32), the V^T fragments of a tile (64).
// One iteration i (64 MFMAs = 64 gaps) = QK^T of tile i+1 (gaps 0-31: query block 0, then 1) + PV of tile i (32-63: block 0, then 1):
//   exp2 of S(i+1) block 0: gaps 17-48, its bf16 pack 48-63 (P block 0 is free once PV(i) block 0 is done at gap 47)
//   exp2 of S(i+1) block 1: gaps 49-63 and 0-12 of iteration i+1 (gaps 0-3 take two), packs 58-61 and 2-15 — all before the
//   QK^T of tile i+2 overwrites S block 1 at gap 16; row sums one gap behind each exp2
//   K fragments of tile i+2 -> AGPRs: gaps 33, 35, .. 63 (each register is free 17 gaps after its first use)
//   V^T fragments of tile i+1: gaps 50-63, 0, 1 (two gaps behind the last MFMA that reads the register)
//   LDS-DMA: 8 pieces per wave and tile pair, spread over the iteration; barrier(i) at gap 32 behind vmcnt(8): K(i+2) and
//   V(i+1), issued more than an iteration earlier, have landed in every wave; the 8 pieces of the current window stay in flight
#include "../dev.h"
#include "../kernels.h"
#include <stdio.h>
#include <utility>
#include <vector>

namespace {

constexpr int HD = 128, BKV = 64;
constexpr int KT = BKV * HD * 2;     // 16 KB K tile: [64 keys][256 B], 16-B chunk c of row r at c ^ (r & 15)
constexpr int VT = HD * BKV * 2;     // 16 KB V^T tile: [128 d][128 B], chunk c of row r at c ^ ((r >> 1) & 7)
constexpr int VRING = 3 * KT;        // LDS: K ring (3 slots), then V ring (3 slots) = 96 KB

template <int I> using ic = std::integral_constant<int, I>;
template <bool B> using bc = std::integral_constant<bool, B>;
template <class F, int... I>
__device__ __forceinline__ void for_seq(F&& f, std::integer_sequence<int, I...>) {
    (f(ic<I>{}), ...);
}
#define A64_INLINE __attribute__((always_inline))
// a generic lambda captures a local only if it is named outside asm operands / `if constexpr` branches
#define A64_USE (void)o; (void)s0; (void)s1; (void)p; (void)kf; (void)vf; (void)qf; (void)l0; (void)l1; (void)kfa; (void)vfa;

// ---- the schedule (gap = the slot behind MFMA number g of the iteration)
constexpr int x0_gap(int e) { return 17 + e; }                                  // exp #e of S block 0 (e = kb*16 + r)
constexpr int x1_gap(int e) { return e < 15 ? 49 + e : e < 23 ? 64 + (e - 15) / 2 : 64 + 4 + (e - 23); }   // >= 64: next iteration
constexpr int c0_gap(int c) { return 48 + c; }                                  // pack #c (elements 2c, 2c+1) of block 0
constexpr int c1_gap(int c) { return c < 4 ? 58 + c : c < 8 ? 64 + 2 + (c - 4) : c < 12 ? 64 + 6 + (c - 8) : 64 + 12 + (c - 12); }
constexpr int kr_gap(int n) { return 33 + 2 * n; }                              // K fragment n = 2 ks + kb of tile i+2
constexpr int vr_gap(int n) { return 50 + n; }                                  // V fragment n = 4 st + db of tile i+1 (>= 64: next iteration)
constexpr int dma_gap(int p) { return p < 5 ? 36 + 6 * p : 64 + 6 + 8 * (p - 5); }   // pieces 0-4 here, 5-7 in the next iteration
constexpr int BAR_GAP = 32;
static_assert(dma_gap(4) < 64 && dma_gap(5) >= 64 && x1_gap(31) == 64 + 12 && c1_gap(15) == 64 + 15 && vr_gap(15) == 65 && dma_gap(7) == 64 + 22, "schedule");

// HAS_QK: tile i+1 exists (QK^T, its softmax, the K / V fragment reads and the DMA run); HAS_PV: tile i exists (PV and the
// second part of its block-1 softmax, which the previous iteration started)
template <bool HAS_QK, bool HAS_PV>
struct Flags {};

__global__ __launch_bounds__(256, 1) void attn64_kernel(const uint16_t* __restrict__ q, const uint16_t* __restrict__ k, int ldqk,
                                                        const uint16_t* __restrict__ vt, int s_pad, uint16_t* __restrict__ out,
                                                        int ldo, int Sq, int S, unsigned long long* trace) {
    __shared__ __attribute__((aligned(16))) uint8_t smem[3 * KT + 3 * VT];
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int head = blockIdx.y, q0 = blockIdx.x * 256 + wid * 64;
    const int nt = S / BKV;

    // ---- Q fragments (B operand of S^T = K Q^T): lane holds Q[q = qb*32 + l31][d = 16 ks + 8 hi + j]
    // Loaded straight into AGPRs by asm (an "a" input fed from a compiler-visible load is re-copied VGPR -> AGPR in front of every
    // use); the loads are older than every DMA piece, so the prologue's counted vmcnt covers them.
    bf16x8 qf[2][8];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const uint16_t* qp = q + (size_t)min(q0 + qb * 32 + l31, Sq - 1) * ldqk + head * HD + hi * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {   // block 0 in AGPRs, block 1 in VGPRs: 224 AGPRs + ~215 VGPRs (all 64 in AGPRs fills the file exactly
                                           // and the allocator starts shuttling values through VGPRs inside the loop)
            if (qb == 0) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=a"(qf[qb][ks]) : "v"(qp), "i"(ks * 32) : "memory");
            else asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(qf[qb][ks]) : "v"(qp), "i"(ks * 32) : "memory");
        }
    }

    // ---- LDS-DMA sources.  K piece j of wave w: rows (4j + w) 4 + (lane >> 4), LDS chunk lane & 15 <- global chunk (lane & 15) ^ (row & 15);
    // V^T piece j: rows (4j + w) 8 + (lane >> 3), LDS chunk lane & 7 <- global chunk (lane & 7) ^ ((row >> 1) & 7).  Offsets are relative
    // to the head's first key row / first d row; the tile adds 64 rows (K) or 128 bytes (V^T) through the scalar base.
    uint32_t voffK[4], voffV[4];
    {
        const int kr = wid * 4 + (lane >> 4), vr = wid * 8 + (lane >> 3);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int krow = j * 16 + kr, vrow = j * 32 + vr;
            voffK[j] = (uint32_t)krow * (uint32_t)ldqk * 2u + (uint32_t)(((lane & 15) ^ (krow & 15)) << 4);
            voffV[j] = (uint32_t)vrow * (uint32_t)s_pad * 2u + (uint32_t)(((lane & 7) ^ ((vrow >> 1) & 7)) << 4);
        }
    }
    const uint8_t* kb8 = reinterpret_cast<const uint8_t*>(k + head * HD);
    const uint8_t* vb8 = reinterpret_cast<const uint8_t*>(vt + (size_t)head * HD * s_pad);
    const uint32_t k_tile_bytes = (uint32_t)BKV * (uint32_t)ldqk * 2u;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem;
    const uint32_t m0_wave = __builtin_amdgcn_readfirstlane(lds_base + wid * 1024);

    // ---- fragment read addresses (slot 0): K fragment (ks, kb): row kb*32 + l31, chunk (2 ks + hi) ^ (l31 & 15);
    // V^T fragment (st, db): row db*32 + l31, chunk (2 st + hi) ^ ((l31 >> 1) & 7)
    uint32_t kfa[8], vfa[4];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) kfa[ks] = lds_base + l31 * 256 + (((ks * 2 + hi) ^ (l31 & 15)) << 4);
#pragma unroll
    for (int st = 0; st < 4; ++st) vfa[st] = lds_base + VRING + l31 * 128 + (((st * 2 + hi) ^ ((l31 >> 1) & 7)) << 4);

    f32x16 o[2][4];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[qb][db][r] = 0.f;
    f32x16 s0[2], s1[2];         // scores of query block 0 / 1, key half kb
    u32x4 p[2][4];               // P^T operands: [query block][k-step of 16 keys]
    bf16x8 kf[16], vf[16];       // K fragments n = 2 ks + kb (AGPRs), V^T fragments n = 4 st + db
    float l0 = 0.f, l1 = 0.f;

    // piece pc (0-3: K piece j = pc, 4-7: V^T piece j = pc - 4): M0 <- LDS destination in ring slot `slot`, then the DMA from tile `tile`
    // (buffer form: SRD of the head's K rows / V^T rows, per-lane offset, tile offset in an SGPR)
    typedef int a64_i32x4 __attribute__((ext_vector_type(4)));
    auto mk_srd = [](const uint8_t* base) {   // raw buffer, stride 0, num_records = 4 GiB - 1, gfx9 word3
        const unsigned long long a = (unsigned long long)base;
        a64_i32x4 r;
        r[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
        r[1] = __builtin_amdgcn_readfirstlane((int)(uint32_t)((a >> 32) & 0xffff));
        r[2] = -1;
        r[3] = 0x00020000;
        return r;
    };
    const a64_i32x4 k_srd = mk_srd(kb8), v_srd = mk_srd(vb8);
    auto dma_m0 = [&](auto pc_c, auto slot_c) A64_INLINE {
        constexpr int pc = decltype(pc_c)::value, slot = decltype(slot_c)::value;
        constexpr int off = pc < 4 ? slot * KT + pc * 4096 : VRING + slot * VT + (pc - 4) * 4096;
        (void)m0_wave;
        asm volatile("s_add_u32 m0, %0, %1" ::"s"(m0_wave), "i"(off) : "memory", "scc");   // s_add writes SCC: hipcc may hold a loop condition there
    };
    auto dma_m0_rt = [&](auto pc_c, uint32_t slot) A64_INLINE {   // the same with a run-time slot (the last tiles of the loop)
        constexpr int pc = decltype(pc_c)::value;
        (void)m0_wave;
        const uint32_t mm = pc < 4 ? m0_wave + slot * KT + pc * 4096 : m0_wave + VRING + slot * VT + (pc - 4) * 4096;
        asm volatile("s_mov_b32 m0, %0" ::"s"(mm) : "memory");
    };
    auto dma_go = [&](auto pc_c, uint32_t tile_off) A64_INLINE {
        constexpr int pc = decltype(pc_c)::value;
        (void)voffK; (void)voffV; (void)k_srd; (void)v_srd;
        if constexpr (pc < 4) asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voffK[pc]), "s"(k_srd), "s"(tile_off) : "memory");
        else asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voffV[pc - 4]), "s"(v_srd), "s"(tile_off) : "memory");
    };
    auto koff = [&](int x) A64_INLINE { return (uint32_t)min(x, nt - 1) * k_tile_bytes; };   // byte offset of K tile x (clamped) from the head's first row
    auto voff = [&](int x) A64_INLINE { return (uint32_t)min(x, nt - 1) * (uint32_t)(BKV * 2); };

#define A64_DSR(dst_, cons_, addr_, off_) asm volatile("ds_read_b128 %0, %1 offset:%2" : cons_(dst_) : "v"(addr_), "i"(off_) : "memory")

    // ---- prologue: K(0) -> fragments; [K(1), V(0)] and pieces 0-4 of [K(2), V(1)] in flight (what iterations -3 and -2 would have issued)
    {
        // K(0) -> slot 0; [K(1) -> slot 1, V(0) -> slot 0]; pieces 0-4 of [K(2) -> slot 2, V(1) -> slot 1]
        for_seq([&](auto pc) A64_INLINE {
            dma_m0(pc, ic<0>{});
            asm volatile("s_nop 0" ::: "memory");
            dma_go(pc, koff(0));
        }, std::make_integer_sequence<int, 4>{});
        for_seq([&](auto pc) A64_INLINE {
            constexpr int pp = decltype(pc)::value;
            dma_m0(pc, ic<(pp < 4 ? 1 : 0)>{});
            asm volatile("s_nop 0" ::: "memory");
            dma_go(pc, pp < 4 ? koff(1) : voff(0));
        }, std::make_integer_sequence<int, 8>{});
        for_seq([&](auto pc) A64_INLINE {
            constexpr int pp = decltype(pc)::value;
            dma_m0(pc, ic<(pp < 4 ? 2 : 1)>{});
            asm volatile("s_nop 0" ::: "memory");
            dma_go(pc, pp < 4 ? koff(2) : voff(1));
        }, std::make_integer_sequence<int, 5>{});
        asm volatile("s_waitcnt vmcnt(13)\n\ts_barrier" ::: "memory");   // K(0) has landed everywhere
        for_seq([&](auto nc) A64_INLINE {
            constexpr int n = decltype(nc)::value;
            A64_USE
            A64_DSR(kf[n], "=a", kfa[n / 2], (n % 2) * 8192);
        }, std::make_integer_sequence<int, 16>{});
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }

    // ---- one iteration i: QK^T of tile i+1 (HAS_QK), PV of tile i (HAS_PV)
    // R = i mod 3 (2 for i = -1): the ring slots are compile-time, so every LDS address is a base register + immediate
    auto iter = [&](auto qk_c, auto pv_c, auto r_c, int i) A64_INLINE {
        constexpr bool HAS_QK = decltype(qk_c)::value, HAS_PV = decltype(pv_c)::value;
        constexpr int R = decltype(r_c)::value;         // -1: run-time slots from i
        constexpr bool CT = R >= 0;
        constexpr int kslot = CT ? ((R + 2) % 3) * KT : 0;       // K(i+2): fragment reads at gaps 33..63
        constexpr int vslot = CT ? ((R + 1) % 3) * VT : 0;       // V(i+1): gaps 50..63
        constexpr int vslot_prev = CT ? R * VT : 0;              // V(i): fragments 14, 15 at gaps 0, 1
        const uint32_t ri = (uint32_t)((i + 3) % 3);             // run-time form of R
        const uint32_t kslot_rt = CT ? 0u : ((ri + 2) % 3) * KT, vslot_rt = CT ? 0u : ((ri + 1) % 3) * VT, vslot_prev_rt = CT ? 0u : ri * VT;
        // pieces 5-7 (gaps 6, 14, 22) of [K(i+3) -> slot R, V(i+2) -> slot R+2]; pieces 0-4 (gaps 36..60) of [K(i+4) -> slot R+1, V(i+3) -> slot R]
        const uint32_t vA = voff(i + 2), kB = koff(i + 4), vB = voff(i + 3);
        auto gap = [&](auto gc) A64_INLINE {
            constexpr int g = decltype(gc)::value;
            A64_USE
            (void)vA; (void)kB; (void)vB; (void)ri; (void)kslot_rt; (void)vslot_rt; (void)vslot_prev_rt;
            // ---- waits in front of the MFMA
            if constexpr (g == 0 && HAS_QK) asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");
            if constexpr (g == 8) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // ---- the MFMA
            if constexpr (g < 32) {
                if constexpr (HAS_QK) {
                    constexpr int qb = g / 16, ks = (g % 16) / 2, kb = g % 2;
                    if constexpr (qb == 0) {
                        if constexpr (ks == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(s0[kb]) : "a"(kf[2 * ks + kb]), "a"(qf[0][ks]));
                        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s0[kb]) : "a"(kf[2 * ks + kb]), "a"(qf[0][ks]));
                    } else {
                        if constexpr (ks == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(s1[kb]) : "a"(kf[2 * ks + kb]), "v"(qf[1][ks]));
                        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s1[kb]) : "a"(kf[2 * ks + kb]), "v"(qf[1][ks]));
                    }
                }
            } else if constexpr (HAS_PV) {
                constexpr int qb = (g - 32) / 16, st = ((g - 32) % 16) / 4, db = (g - 32) % 4;
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(o[qb][db]) : "v"(vf[4 * st + db]), "v"(p[qb][st]));
            }
            // ---- fillers, in issue order
            if constexpr (g == BAR_GAP) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
            // exp2 of S block 0 (tile i+1): e -> s0[e / 16][e % 16]; its row-sum add one gap later
            // (one asm statement for the pair: hipcc puts an s_nop between two asm statements it takes for a producer / consumer pair)
            if constexpr (HAS_QK && g >= x0_gap(0) + 1 && g <= x0_gap(31)) {
                constexpr int e = g - x0_gap(0);
                asm volatile("v_exp_f32 %0, %0\n\tv_add_f32 %1, %1, %2" : "+v"(s0[e / 16][e % 16]), "+v"(l0) : "v"(s0[(e - 1) / 16][(e - 1) % 16]));
            } else if constexpr (HAS_QK && g == x0_gap(0)) {
                asm volatile("v_exp_f32 %0, %0" : "+v"(s0[0][0]));
            } else if constexpr (HAS_QK && g == x0_gap(31) + 1) {
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(l0) : "v"(s0[1][15]));
            }
            // exp2 of S block 1: first part (tile i+1) at gaps 49..63, second part (tile i, started by the previous iteration) at gaps 0..12
            for_seq([&](auto ec) A64_INLINE {
                constexpr int e = decltype(ec)::value;
                A64_USE
                if constexpr (HAS_QK && x1_gap(e) == g) asm volatile("v_exp_f32 %0, %0" : "+v"(s1[e / 16][e % 16]));
                if constexpr (HAS_PV && x1_gap(e) == g + 64) asm volatile("v_exp_f32 %0, %0" : "+v"(s1[e / 16][e % 16]));
                if constexpr (HAS_QK && x1_gap(e) + 1 == g) asm volatile("v_add_f32 %0, %0, %1" : "+v"(l1) : "v"(s1[e / 16][e % 16]));
                if constexpr (HAS_PV && x1_gap(e) + 1 == g + 64) asm volatile("v_add_f32 %0, %0, %1" : "+v"(l1) : "v"(s1[e / 16][e % 16]));
            }, std::make_integer_sequence<int, 32>{});
            // bf16 packs: c -> P[qb][st = c / 4][c % 4] <- elements 2c, 2c+1 of the block's 32 (kb = c / 8, r = 2c % 16)
            for_seq([&](auto cc) A64_INLINE {
                constexpr int c = decltype(cc)::value;
                A64_USE
                if constexpr (HAS_QK && c0_gap(c) == g)
                    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p[0][c / 4][c % 4]) : "v"(s0[c / 8][(2 * c) % 16]), "v"(s0[c / 8][(2 * c) % 16 + 1]));
                if constexpr (HAS_QK && c1_gap(c) == g)
                    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p[1][c / 4][c % 4]) : "v"(s1[c / 8][(2 * c) % 16]), "v"(s1[c / 8][(2 * c) % 16 + 1]));
                if constexpr (HAS_PV && c1_gap(c) == g + 64)
                    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p[1][c / 4][c % 4]) : "v"(s1[c / 8][(2 * c) % 16]), "v"(s1[c / 8][(2 * c) % 16 + 1]));
            }, std::make_integer_sequence<int, 16>{});
            // K fragments of tile i+2, V^T fragments of tile i+1 (14, 15: of tile i, at gaps 0, 1)
            for_seq([&](auto nc) A64_INLINE {
                constexpr int n = decltype(nc)::value;
                A64_USE (void)kslot_rt; (void)vslot_rt; (void)vslot_prev_rt;
                if constexpr (HAS_QK && kr_gap(n) == g) { const uint32_t a = kfa[n / 2] + kslot_rt; A64_DSR(kf[n], "=a", a, kslot + (n % 2) * 8192); }
                if constexpr (HAS_QK && vr_gap(n) == g) { const uint32_t a = vfa[n / 4] + vslot_rt; A64_DSR(vf[n], "=v", a, vslot + (n % 4) * 4096); }
                if constexpr (HAS_PV && vr_gap(n) == g + 64) { const uint32_t a = vfa[n / 4] + vslot_prev_rt; A64_DSR(vf[n], "=v", a, vslot_prev + (n % 4) * 4096); }
            }, std::make_integer_sequence<int, 16>{});
            // LDS-DMA: M0 one gap ahead of the piece
            for_seq([&](auto pc) A64_INLINE {
                constexpr int pp = decltype(pc)::value;
                (void)vA; (void)kB; (void)vB;
                (void)ri;
                if constexpr (HAS_QK && dma_gap(pp) - 1 == g) {
                    if constexpr (CT) dma_m0(pc, ic<(pp < 4 ? (R + 1) % 3 : (R + 3) % 3)>{});
                    else dma_m0_rt(pc, pp < 4 ? (ri + 1) % 3 : ri);
                }
                if constexpr (HAS_QK && dma_gap(pp) == g) dma_go(pc, pp < 4 ? kB : vB);
                if constexpr (HAS_QK && dma_gap(pp) - 1 == g + 64) {
                    if constexpr (CT) dma_m0(pc, ic<(pp < 4 ? (R + 3) % 3 : (R + 2) % 3)>{});
                    else dma_m0_rt(pc, pp < 4 ? ri : (ri + 2) % 3);
                }
                if constexpr (HAS_QK && dma_gap(pp) == g + 64) dma_go(pc, vA);   // pieces 5-7 are V^T pieces
            }, std::make_integer_sequence<int, 8>{});
        };
        for_seq(gap, std::make_integer_sequence<int, 64>{});
    };
    // Between the code regions below hipcc may re-assign registers (copies at the region's end): every fragment read and every
    // MFMA result must have landed before such a copy reads it.  Not needed inside the main loop (no copies there: checked in the ISA).
    auto drain = [&]() A64_INLINE { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory"); };
    // dev diagnostic (trace != nullptr, FLUX2MI_ATTN_TRACE with the dev library): shader-clock and 100 MHz stamps around the tile loop
    unsigned long long tr_c0 = 0, tr_r0 = 0;
    if (trace) asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(tr_c0), "=s"(tr_r0)::"memory");
    // iteration -1: QK^T of tile 0 only.  Its pieces 5-7 belong to [K(2), V(1)], 0-4 to [K(3), V(2)].
    iter(bc<true>{}, bc<false>{}, ic<2>{}, -1);
    drain();
    int i = 0;
    for (; i + 3 < nt; i += 3) {      // three tiles per trip: i mod 3 is compile-time inside
        iter(bc<true>{}, bc<true>{}, ic<0>{}, i);
        iter(bc<true>{}, bc<true>{}, ic<1>{}, i + 1);
        iter(bc<true>{}, bc<true>{}, ic<2>{}, i + 2);
        if (!(i + 6 < nt)) drain();   // last trip: in front of the exit's copies (hipcc places them before anything that follows the loop)
    }
    for (; i + 1 < nt; ++i) {         // at most three more with run-time slots
        iter(bc<true>{}, bc<true>{}, ic<-1>{}, i);
        drain();
    }
    iter(bc<false>{}, bc<true>{}, ic<-1>{}, i);
#undef A64_DSR
    // MFMA result -> first non-MFMA reader; every DMA piece has landed before the workgroup leaves
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");

    if (trace) {
        unsigned long long c1, r1;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1)::"memory");
        if (lane == 0) {
            unsigned long long* dst = trace + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wid) * 2;
            dst[0] = c1 - tr_c0;
            dst[1] = r1 - tr_r0;
        }
    }
    // ---- normalise and store.  o[qb][db][r]: d = db*32 + (r & 3) + 8 (r >> 2) + 4 hi, query = q0 + qb*32 + l31
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        float a = qb ? l1 : l0, b = a;
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
        const float inv = 1.0f / (a + b);
        const int qrow = q0 + qb * 32 + l31;
        if (qrow < Sq) {
            uint16_t* op = out + (size_t)qrow * ldo + head * HD + hi * 4;
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint2 v;
                    v.x = pack_bf16x2(o[qb][db][g * 4 + 0] * inv, o[qb][db][g * 4 + 1] * inv);
                    v.y = pack_bf16x2(o[qb][db][g * 4 + 2] * inv, o[qb][db][g * 4 + 3] * inv);
                    *reinterpret_cast<uint2*>(op + db * 32 + g * 8) = v;
                }
        }
    }
}

}  // namespace

// 0 launched / 1 not this kernel's case (the caller falls back to attention.hip) / 2 launch error.
// Requires: q pre-multiplied by scale*log2(e) and the fixed zero offset valid (the caller checks), S % 64 == 0.
int launch_attention64(hipStream_t stream, const uint16_t* q, const uint16_t* k, int ldqk, const uint16_t* vt, int s_pad, uint16_t* out,
                       int ldo, int Sq, int S, int heads) {
    if (S % BKV != 0 || S < BKV || Sq <= 0) return 1;
    if ((unsigned long long)S * (unsigned long long)ldqk * 2 >= (1ull << 31) || (unsigned long long)HD * (unsigned long long)s_pad * 2 >= (1ull << 31)) return 1;
    const dim3 grid((Sq + 255) / 256, heads), block(256);
    unsigned long long* trace = nullptr;
    static const char* trace_env = F2_DEV_ENV("FLUX2MI_ATTN_TRACE");
    static int seen = 0;
    const size_t tn = (size_t)grid.x * grid.y * 4 * 2;
    if (trace_env && ++seen % 8 == 0 && hipMalloc((void**)&trace, tn * 8) != hipSuccess) trace = nullptr;
    hipLaunchKernelGGL(attn64_kernel, grid, block, 0, stream, q, k, ldqk, vt, s_pad, out, ldo, Sq, S, trace);
    if (trace) {
        std::vector<unsigned long long> h(tn);
        (void)hipStreamSynchronize(stream);
        (void)hipMemcpy(h.data(), trace, tn * 8, hipMemcpyDeviceToHost);
        (void)hipFree(trace);
        double cyc = 0, rt = 0;
        for (size_t w = 0; w < tn / 2; ++w) { cyc += (double)h[2 * w]; rt += (double)h[2 * w + 1]; }
        const double tiles = (double)(tn / 2) * (S / BKV + 1);
        fprintf(stderr, "[attn64 trace] S %d heads %d: %.0f shader cycles per KV tile (64 MFMAs), in-loop clock %.0f MHz, %.3f us per tile\n", S, heads,
                cyc / tiles, cyc / rt * 100.0, rt / tiles * 0.01);
    }
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
