// attention64b.hip — the blocks' attention (one wave per SIMD, 4 waves x 64 query rows, LDS-DMA rings, hand-placed tile
// loop) on v_mfma_f32_16x16x32_bf16 instead of 32x32x16.
//
// Why: at 88 % MFMA issue efficiency attention64 is power-bound (in-loop clock 1.67 GHz), and register-only streams on these
// boxes sustain 1.94 PF with the 16x16x32 shape against 1.71 PF with 32x32x16 under the same softmax fillers
// (a register-only probe of round 2, profiles/r02_g_mfma_power_probe.log): the larger shape costs ~13 % more energy per FLOP.
//
// Operand geometry (lane = (c16 = lane & 15, g4 = lane >> 4)):
//   S^T tile = K fragment (A: 16 key rows x 32 d) x Q fragment (B: 32 d x 16 queries): lane holds S[q = c16][4 key rows 4 g4 + r].
//   The key rows of a K fragment are a free choice (any LDS row per lane): fragment kb = 2 st + u, row m <-> key
//   32 st + 16 (m >> 3) + 8 u + 4 ((m >> 2) & 1) + (m & 3).  With it the 8 values a lane owns per 32-key step st (rows 4 g4 + r of
//   fragments u = 0, 1) are exactly the keys of ONE 16-byte chunk of the V^T image as attention.hip's producers write it (keys of
//   every 16-group permuted to 8h + 4a + b): chunk 4 st + g4.  So P^T (B operand of O^T += V^T P^T) is built in the lane from its own
//   S registers, the V^T fragment (A: 16 d rows x 32 keys) is one ds_read_b128, and no producer changes.
//   K tile in LDS: [64 keys][256 B], chunk c of row r at c ^ ((r & 7) | ((r >> 4) & 1) << 3) (the 16 rows of a fragment get 16
//   different keys); V^T tile: [128 d][128 B], chunk c at c ^ ((r >> 1) & 7).
// Per wave: O^T 8 x 4 accumulators (128 AGPRs), K fragments of a tile (64 AGPRs), Q blocks 0, 1 (32 AGPRs), Q blocks 2, 3 (32 VGPRs),
// S 4 x 4 (64 VGPRs), P 2 x 4 (32), V^T fragments of a tile (64).
// One iteration i = 136 MFMAs (16 cycles each): QK^T of tile i+1 for query blocks 0, 1 (gaps 0-31) and 2, 3 (32-63), PV of tile i
// for blocks 0, 1 (64-99) and 2, 3 (100-135).  An MFMA holds the vector issue for 8 of its 16 cycles, so a gap takes ONE v_exp_f32
// or two 4-cycle instructions, and 64 row-sum adds per tile do not fit (measured: 2585 cycles per tile with them, every odd gap
// over budget).  The row sums therefore ride on the matrix pipe: per 32-key step and query block one more MFMA whose A operand is
// all ones gives sum_k P^T[k][q] in every row of a 16x16 accumulator — the sum of the SAME bf16 P that multiplies V, reduced across
// the lanes for free (8 MFMAs per tile instead of 64 v_add_f32 + the final cross-lane sums).  exp2 on the even gaps (blocks 0, 1:
// 34-96; blocks 2, 3: 98-134 and 0-24 of the next iteration), packs / fragment reads / DMA on the odd ones.
#include "dev.h"
#include "kernels.h"
#include <stdio.h>
#include <utility>
#include <vector>

namespace {

constexpr int HD = 128, BKV = 64;
// QK8: the output as block-scaled e4m3 instead of bf16 (the A operand of the fp8 models' proj / linear2): bytes q8[row * ld + head * 128 + d],
// scale bytes bs[head * bs_ld + row / 16] (one per 16 queries x this head's 128 columns); q8 = nullptr: the bf16 output
struct AttnBlocksOut { uint8_t* q8; int ld; uint8_t* bs; int bs_ld; };
// KV-split of the partial last round (fixed-offset softmax only: partial O and row sums of disjoint key ranges just add).  The
// (head, query block) items rarely fill the CUs a whole number of times — Klein-9B 32 heads x 18 = 576 items on 256 CUs = 2.25 rounds
// paid as 3, Dev 864 = 3.375 paid as 4.  With f > 1 the launch is 1-D: workgroups [0, n_whole) run whole items exactly as before,
// workgroup n_whole + r runs KV tiles [part nt / f, (part + 1) nt / f) of item n_whole + r / f, part = r % f, and leaves its raw
// accumulators in ws[r] (SPLIT_PIECE_FLOATS floats); attn64b_combine_kernel adds an item's f pieces in part order (bit-reproducible) and
// stores through the same epilogue.  f = 1: the plain 2-D launch, nothing else changes.
struct AttnSplit { int n_whole; int f; int qblocks; float* ws; };
constexpr int SPLIT_PIECE_FLOATS = 33 * 256 * 4;   // 32 accumulator quads + the four row sums, per lane
constexpr int KT = BKV * HD * 2;     // 16 KB
constexpr int VT = HD * BKV * 2;     // 16 KB
constexpr int VRING = 3 * KT;        // LDS: K ring (3 slots), then V ring (3 slots) = 96 KB

template <int I> using ic = std::integral_constant<int, I>;
template <bool B> using bc = std::integral_constant<bool, B>;
template <class F, int... I>
__device__ __forceinline__ void for_seq(F&& f, std::integer_sequence<int, I...>) {
    (f(ic<I>{}), ...);
}
#define B64_INLINE __attribute__((always_inline))
#define B64_USE (void)o; (void)s; (void)p; (void)kf; (void)vf; (void)qf; (void)lacc; (void)ones; (void)pmask; (void)kfa; (void)vfa; (void)negm; (void)kf8; (void)qf8; (void)sc_q; (void)sc_k;

// ---- the schedule (gap = the slot behind MFMA number g of the iteration; >= T: the same slot of the NEXT iteration)
// softmax element e of a half (query blocks 2H, 2H+1): e = st*16 + qh*8 + u*4 + r -> s[2 st + u][2H + qh][r]; pack c = e / 2
constexpr int T = 136;                                   // 64 QK^T + 2 x (2 steps x (16 PV + 2 row-sum)) MFMAs
constexpr int x0_gap(int e) { return 34 + 2 * e; }       // 34..96
constexpr int c0_gap(int c) { return 101 + 2 * c; }      // 101..131: P blocks 0, 1 are free once PV(i) blocks 0, 1 are done (gap 99)
constexpr int x1_gap(int e) { return e < 19 ? 98 + 2 * e : T + 2 * (e - 19); }          // 98..134, then 0..24
constexpr int c1_gap(int c) { return c < 9 ? T + 1 + 2 * c : T + 2 * c + 1; }          // 1..31 of the next iteration: before QK^T(i+2) reaches S blocks 2, 3 at gap 32
constexpr int kr_gap(int n) { return 65 + 2 * n; }       // K fragment n = 4 ks + kb of tile i+2 (its register was last read at gap 33 + 2n)
constexpr int vr_gap(int n) { return 103 + 18 * (n / 8) + 2 * (n % 8); }               // V^T fragment n = 8 st + db of tile i+1 (last read at 101 + 18 st + 2 db)
constexpr int dma_gap(int pc) { return pc < 4 ? 71 + 16 * pc : T + 7 + 16 * (pc - 4); }   // K pieces of tile i+4 here, V^T pieces of tile i+2 in the next window
constexpr int BAR_GAP = 64;
// immediate offsets of the fragments inside a tile image
constexpr int kimm(int n) { return ((n % 4) / 2) * 8192 + ((n % 4) % 2) * 2048; }   // K fragment kb = n % 4 = 2 st + u: rows 32 st + 8 u
constexpr int vimm(int n) { return (n % 8) * 2048; }                                // d block n % 8
constexpr int kimm8(int kb) { return (kb / 2) * 4096 + (kb % 2) * 1024; }           // QK8: K operand kb = 2 st + u of the [64][128 B] tile: rows 32 st + 8 u
static_assert(x1_gap(31) == T + 24 && c1_gap(15) == T + 31 && vr_gap(15) == 135 && kr_gap(15) == 95 && c0_gap(15) == 131, "schedule");

// DOFF = false: p = 2^s (the caller guarantees |s| <= 80: attention_uses_fixed_offset).  DOFF = true: any scores.  p = 2^(s - m_q) with a
// per-query offset m_q = the query's largest score against KV tile 0, subtracted for free as the C operand of the first QK^T MFMA of
// every S block; nothing else in the tile loop changes.  A softmax only needs ONE offset per row that keeps every p inside the f32
// range, not the running maximum: m_q <= the row maximum, so nothing that matters underflows, and a row whose later scores exceed
// m_q by more than 90 (62 in natural units) shows up in its row sum (>= 2^90, inf or NaN), which the ones-MFMA computes anyway: the
// workgroup then repeats its pass with that query's offset raised (by log2 of the sum, or by 100 when it overflowed).  Exact for any
// input; the repeat is a safety net that trained attention does not reach.
// QK8 (fp8 models, plain forwards): Q and K arrive as OCP e4m3 bytes (q8 = q x 2^qe, k8 = k x 2^ke, written by the QK norm + RoPE
// kernel; `q`, `k` then point at BYTES and ldqk is a byte stride), and S^T = K Q^T is ONE v_mfma_scale_f32_16x16x128_f8f6f4 per 16 x 16
// block (K = 128 = the head dim; 32 cycles) instead of four 16x16x32 bf16 MFMAs (64): the power-of-two scales are undone for free by
// the instruction's e8m0 scale operands.  Everything behind the scores (exp2, P, PV in bf16, row sums) is unchanged, and so is the
// slot structure of the iteration: the QK^T part keeps its 64 slots, of which every fourth carries an MFMA.
template <bool DOFF, bool QK8 = false>
__global__ __launch_bounds__(256, 1) void attn64b_kernel(const uint16_t* __restrict__ q, const uint16_t* __restrict__ k, int ldqk,
                                                         const uint16_t* __restrict__ vt, int s_pad, uint16_t* __restrict__ out,
                                                         int ldo, int Sq, int S_total, unsigned long long* trace, int qk8_exps, AttnBlocksOut bo,
                                                         AttnSplit sp) {
    __shared__ __attribute__((aligned(16))) uint8_t smem[3 * KT + 3 * VT];
    __shared__ unsigned bo_max[4][4];     // AttnBlocksOut: block maxima [wave][query block], as float bits
    __shared__ int s_again;               // DOFF: some wave of the workgroup asks for another pass
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const int c16 = lane & 15, g4 = lane >> 4;
    // Workgroup -> (head, query block).  Workgroups are dealt to the 8 XCDs round-robin in launch order (id % 8); with the head count a
    // multiple of 8, XCD x takes heads x, x + 8, ... and runs the query blocks of one head back to back, so that the workgroups of a head
    // stream its K / V^T through ONE L2 instead of eight.
    int head = blockIdx.y, qblk = blockIdx.x;
    int S = S_total, kv_tile0 = 0;          // this workgroup's keys: [kv_tile0 * 64, kv_tile0 * 64 + S)
    float* piece_ws = nullptr;              // a piece of a KV-split item: raw accumulators go here
    if (sp.f <= 1) {
        if ((gridDim.y & 7) == 0) {
            const int id = blockIdx.y * gridDim.x + blockIdx.x, xcd = id & 7, k = id >> 3;
            head = (k / (int)gridDim.x) * 8 + xcd;
            qblk = k % (int)gridDim.x;
        }
    } else {   // 1-D launch (the launcher guarantees heads % 8 == 0 and !DOFF): items in the same XCD-aware order, then the pieces
        int item = blockIdx.x;
        if (item >= sp.n_whole) {
            const int r = item - sp.n_whole, part = r % sp.f, ntf = (S_total + BKV - 1) / BKV;
            item = sp.n_whole + r / sp.f;
            kv_tile0 = part * ntf / sp.f;
            const int kv_tile1 = (part + 1) * ntf / sp.f;
            S = min(S_total, kv_tile1 * BKV) - kv_tile0 * BKV;
            piece_ws = sp.ws + (size_t)r * SPLIT_PIECE_FLOATS;
        }
        const int xcd = item & 7, k = item >> 3;
        head = (k / sp.qblocks) * 8 + xcd;
        qblk = k % sp.qblocks;
    }
    const int q0 = qblk * 256 + wid * 64;
    const int nt = (S + BKV - 1) / BKV;   // the last tile may be partial: its missing keys score against key S - 1 again (clamped rows: finite
                                          // scores) and their P^T slots are cleared in front of the last tile's PV (pmask), so that neither the
                                          // row sums nor O see them — whatever V^T holds behind key S - 1 (the KV-extract pass runs the reference
                                          // queries against a key PREFIX of the shared V^T image: real values follow).  V^T must be finite there.

    // ---- Q fragments (B operand of S^T = K Q^T): lane holds Q[q = qb*16 + c16][d = 32 ks + 8 g4 + j]; loaded by asm straight into
    // their registers (blocks 0, 1 in AGPRs, 2, 3 in VGPRs); the loads are older than every DMA piece: the prologue's vmcnt covers them
    typedef uint32_t b64_u32x8 __attribute__((ext_vector_type(8)));
    bf16x8 qf[QK8 ? 1 : 4][4];
    b64_u32x8 qf8[4];            // QK8: lane holds the 32 bytes d = 32 g4 .. of query qb*16 + c16 (B operand of the 16x16x128 MFMA)
    (void)qf8;
    if constexpr (!QK8) {
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) {
            const uint16_t* qp = q + (size_t)min(q0 + qb * 16 + c16, Sq - 1) * ldqk + head * HD + g4 * 8;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (qb < 2) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=a"(qf[qb][ks]) : "v"(qp), "i"(ks * 64) : "memory");
                else asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(qf[qb][ks]) : "v"(qp), "i"(ks * 64) : "memory");
            }
        }
    } else {
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) {
            const uint8_t* qp = reinterpret_cast<const uint8_t*>(q) + (size_t)min(q0 + qb * 16 + c16, Sq - 1) * ldqk + head * HD + g4 * 32;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(qf8[qb].lo) : "v"(qp) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(qf8[qb].hi) : "v"(qp) : "memory");
        }
    }
    // QK8: the e8m0 scale operands that undo q8 = q 2^qe, k8 = k 2^ke (qk8_exps = the two e8m0 bytes 127 - qe | (127 - ke) << 8), in every byte
    uint32_t sc_q = 0x01010101u * (uint32_t)(qk8_exps & 0xff), sc_k = 0x01010101u * (uint32_t)((qk8_exps >> 8) & 0xff);
    asm volatile("" : "+v"(sc_q), "+v"(sc_k));

    // ---- LDS-DMA sources.  K piece j of wave w: rows 16 j + 4 w + (lane >> 4), LDS chunk lane & 15 <- global chunk (lane & 15) ^ f(row),
    // f(row) = (row & 7) | ((row >> 4) & 1) << 3; V^T piece j: rows 32 j + 8 w + (lane >> 3), LDS chunk lane & 7 <- global chunk
    // (lane & 7) ^ ((row >> 1) & 7)
    uint32_t voffK[4], voffV[4], kmax[4];   // kmax: the same chunk of the last key row (a partial last tile re-reads row S - 1: finite scores, masked later)
    {
        const int kr = wid * 4 + (lane >> 4), vr = wid * 8 + (lane >> 3);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int krow = j * 16 + kr, vrow = j * 32 + vr;
            const int fk = (krow & 7) | (((krow >> 4) & 1) << 3);
            if constexpr (!QK8) {
                voffK[j] = (uint32_t)krow * (uint32_t)ldqk * 2u + (uint32_t)(((lane & 15) ^ fk) << 4);
                kmax[j] = (uint32_t)(S - 1) * (uint32_t)ldqk * 2u + (uint32_t)(((lane & 15) ^ fk) << 4);
            } else {
                // e4m3 K tile [64 keys][128 B]: piece j (of 2) of wave w = rows 32 j + 8 w + (lane >> 3) — the V^T piece geometry; chunk c
                // of row r at c ^ f8(r), f8(r) = ((r >> 1) & 3) | ((r >> 4) & 1) << 2 (the 16 rows of a fragment hit 16 different banks)
                const int r8 = (j & 1) * 32 + vr;
                const int f8 = ((r8 >> 1) & 3) | (((r8 >> 4) & 1) << 2);
                voffK[j] = (uint32_t)r8 * (uint32_t)ldqk + (uint32_t)(((lane & 7) ^ f8) << 4);
                kmax[j] = (uint32_t)(S - 1) * (uint32_t)ldqk + (uint32_t)(((lane & 7) ^ f8) << 4);
            }
            voffV[j] = (uint32_t)vrow * (uint32_t)s_pad * 2u + (uint32_t)(((lane & 7) ^ ((vrow >> 1) & 7)) << 4);
        }
    }
    const uint32_t k_tile_bytes = (uint32_t)BKV * (uint32_t)ldqk * (QK8 ? 1u : 2u);
    const uint8_t* kb8 = (QK8 ? reinterpret_cast<const uint8_t*>(k) + head * HD : reinterpret_cast<const uint8_t*>(k + head * HD)) + (size_t)kv_tile0 * k_tile_bytes;
    const uint8_t* vb8 = reinterpret_cast<const uint8_t*>(vt + (size_t)head * HD * s_pad) + (size_t)kv_tile0 * (BKV * 2);
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem;
    const uint32_t m0_wave = __builtin_amdgcn_readfirstlane(lds_base + wid * 1024);

    // ---- fragment read addresses (slot 0).  K fragment (kb = 2 st + u, ks): row 32 st + 8 u + [16 (c16 >> 3) + 4 ((c16 >> 2) & 1) + (c16 & 3)],
    // chunk (4 ks + g4) ^ c16 (f(row) of that row IS c16).  V^T fragment (db, st): row db*16 + c16, chunk (4 st + g4) ^ ((c16 >> 1) & 7).
    uint32_t kfa[4], vfa[2];
    {
        const int krow = 16 * (c16 >> 3) + 4 * ((c16 >> 2) & 1) + (c16 & 3);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if constexpr (!QK8) kfa[ks] = lds_base + krow * 256 + (((ks * 4 + g4) ^ c16) << 4);
            else {   // QK8: kfa[h], h = 0, 1: the two 16-byte halves of the lane's 32 bytes (d = 32 g4 ..) of key row krow
                const int f8 = ((krow >> 1) & 3) | (((krow >> 4) & 1) << 2);
                kfa[ks] = lds_base + krow * 128 + ((((2 * g4 + (ks & 1)) ^ f8)) << 4);
            }
        }
#pragma unroll
        for (int st = 0; st < 2; ++st) vfa[st] = lds_base + VRING + c16 * 128 + (((st * 4 + g4) ^ ((c16 >> 1) & 7)) << 4);
    }

    f32x4 o[8][4];               // O^T accumulators [d block][query block]
    f32x4 s[4][4];               // scores [K fragment kb = 2 st + u][query block]
    u32x4 p[2][4];               // P^T operands [k-step of 32 keys][query block]
    bf16x8 kf[QK8 ? 1 : 16], vf[16];   // K fragments n = 4 ks + kb (AGPRs), V^T fragments n = 8 st + db
    b64_u32x8 kf8[4];            // QK8: K operand kb = 2 st + u (16 key rows x 128 d bytes), halves written by their own ds_read_b128
    (void)kf8;
    f32x4 lacc[4];               // row sums: every row of the accumulator holds sum_k P[q = c16][k] (A operand = ones)
    f32x4 negm[4];               // DOFF: -m_q of query (qb, c16) in all four components (the C operand that starts an S block)
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) negm[qb] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (DOFF && t == 0) s_again = 0;
    bf16x8 ones;
    u32x4 pmask[2];              // pmask[st]: all-ones bits for the k-slots of step st whose key exists in the last tile (slot (g4, j) <-> key
                                 // 32 st + 16 (g4 >> 1) + 8 (j >> 2) + 4 (g4 & 1) + (j & 3)), 0 for the others; word w = slots 2w, 2w + 1
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;
    {
        const int valid = S - (nt - 1) * BKV;
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                uint32_t m = 0;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int j = 2 * w + h;
                    if ((32 * st + 16 * (g4 >> 1) + 8 * (j >> 2) + 4 * (g4 & 1) + (j & 3)) < valid) m |= 0xffffu << (16 * h);
                }
                pmask[st][w] = m;
            }
    }
    asm volatile("" : "+v"(ones));   // one register tuple, not re-materialised in the loop

    typedef int b64_i32x4 __attribute__((ext_vector_type(4)));
    auto mk_srd = [](const uint8_t* base, uint32_t num_records) {   // raw buffer, stride 0, gfx9 word3
        const unsigned long long a = (unsigned long long)base;
        b64_i32x4 r;
        r[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
        r[1] = __builtin_amdgcn_readfirstlane((int)(uint32_t)((a >> 32) & 0xffff));
        r[2] = (int)num_records;
        r[3] = 0x00020000;
        return r;
    };
    // K: the tile offset is added to the lane offset and clamped to the last key row (kmax)
    const b64_i32x4 k_srd = mk_srd(kb8, 0xffffffffu), v_srd = mk_srd(vb8, 0xffffffffu);
    // piece pc (0-3: K piece j = pc, 4-7: V^T piece j = pc - 4): M0 <- LDS destination in ring slot `slot`, then the DMA
    auto dma_m0 = [&](auto pc_c, auto slot_c) B64_INLINE {
        constexpr int pc = decltype(pc_c)::value, slot = decltype(slot_c)::value;
        constexpr int off = pc < 4 ? slot * KT + pc * 4096 : VRING + slot * VT + (pc - 4) * 4096;
        (void)m0_wave;
        asm volatile("s_add_u32 m0, %0, %1" ::"s"(m0_wave), "i"(off) : "memory", "scc");
    };
    auto dma_m0_rt = [&](auto pc_c, uint32_t slot) B64_INLINE {
        constexpr int pc = decltype(pc_c)::value;
        (void)m0_wave;
        const uint32_t mm = pc < 4 ? m0_wave + slot * KT + pc * 4096 : m0_wave + VRING + slot * VT + (pc - 4) * 4096;
        asm volatile("s_mov_b32 m0, %0" ::"s"(mm) : "memory");
    };
    constexpr int NKP = QK8 ? 2 : 4;   // K pieces per wave and tile (pieces NKP..3 do not exist)
    auto dma_go = [&](auto pc_c, uint32_t tile_off) B64_INLINE {
        constexpr int pc = decltype(pc_c)::value;
        (void)voffK; (void)voffV; (void)kmax; (void)k_srd; (void)v_srd;
        if constexpr (pc < 4) {
            const uint32_t vo = min(voffK[pc] + tile_off, kmax[pc]);
            asm volatile("buffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(vo), "s"(k_srd) : "memory");
        } else asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voffV[pc - 4]), "s"(v_srd), "s"(tile_off) : "memory");
    };
    auto koff = [&](int x) B64_INLINE { return (uint32_t)min(x, nt - 1) * k_tile_bytes; };
    auto voff = [&](int x) B64_INLINE { return (uint32_t)min(x, nt - 1) * (uint32_t)(BKV * 2); };

#define B64_DSR(dst_, cons_, addr_, off_) asm volatile("ds_read_b128 %0, %1 offset:%2" : cons_(dst_) : "v"(addr_), "i"(off_) : "memory")

    // ---- one iteration i: QK^T of tile i+1 (HAS_QK), PV of tile i (HAS_PV); R = i mod 3 (2 for i = -1), -1 = run-time ring slots
    // (defined in front of the pass loop; the MFMA of S block (kb, qb), k-step ks, is also what the offset estimate issues)
    // QK8: the whole S block (kb, qb) in one MFMA
    auto qk8_mfma = [&](auto kb_c, auto qb_c, auto off_c) B64_INLINE {
        constexpr int kb = decltype(kb_c)::value, qb = decltype(qb_c)::value;
        constexpr bool off = decltype(off_c)::value;
        B64_USE
        // (operands in VGPRs: a half-written 8-register AGPR tuple makes hipcc assemble it through v_accvgpr copies)
        if constexpr (off) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %3, %4, %5 op_sel_hi:[0,0,0]" : "=&v"(s[kb][qb]) : "v"(kf8[kb]), "v"(qf8[qb]), "v"(negm[qb]), "v"(sc_k), "v"(sc_q));
        else asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, 0, %3, %4 op_sel_hi:[0,0,0]" : "=&v"(s[kb][qb]) : "v"(kf8[kb]), "v"(qf8[qb]), "v"(sc_k), "v"(sc_q));
    };
    auto qk_mfma = [&](auto kb_c, auto qb_c, auto ks_c, auto off_c) B64_INLINE {
        constexpr int kb = decltype(kb_c)::value, qb = decltype(qb_c)::value, ks = decltype(ks_c)::value;
        constexpr bool off = decltype(off_c)::value;   // start the block from -m_q instead of 0
        B64_USE
        if constexpr (qb < 2) {
            if constexpr (ks == 0 && off) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(s[kb][qb]) : "a"(kf[4 * ks + kb]), "a"(qf[qb][ks]), "v"(negm[qb]));
            else if constexpr (ks == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(s[kb][qb]) : "a"(kf[4 * ks + kb]), "a"(qf[qb][ks]));
            else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(s[kb][qb]) : "a"(kf[4 * ks + kb]), "a"(qf[qb][ks]));
        } else {
            if constexpr (ks == 0 && off) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(s[kb][qb]) : "a"(kf[4 * ks + kb]), "v"(qf[qb][ks]), "v"(negm[qb]));
            else if constexpr (ks == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(s[kb][qb]) : "a"(kf[4 * ks + kb]), "v"(qf[qb][ks]));
            else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(s[kb][qb]) : "a"(kf[4 * ks + kb]), "v"(qf[qb][ks]));
        }
    };
    // the QK^T slot g (0..63) of an iteration: bf16 = MFMA (ks, kb, qb) of the slot; QK8 = every fourth slot, one whole block
    auto qk_slot = [&](auto gc, auto off_c) B64_INLINE {
        constexpr int g = decltype(gc)::value;
        if constexpr (!QK8) {
            constexpr int H = g / 32, ks = (g % 32) / 8, kb = (g % 8) / 2, qb = 2 * H + g % 2;
            qk_mfma(ic<kb>{}, ic<qb>{}, ic<ks>{}, off_c);
        } else if constexpr (g % 4 == 0) {
            constexpr int idx = g / 4, H = idx / 8, kb = (idx % 8) / 2, qb = 2 * H + idx % 2;
            qk8_mfma(ic<kb>{}, ic<qb>{}, off_c);
        }
    };
    // Between the code regions below hipcc may re-assign registers (copies at the region's end): every fragment read and every
    // MFMA result must have landed before such a copy reads it.  Not needed inside the main loop (no copies there: checked in the ISA).
    auto drain = [&]() B64_INLINE { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory"); };
    unsigned long long tr_c0 = 0, tr_r0 = 0;
    if (trace) asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(tr_c0), "=s"(tr_r0)::"memory");

  for (int pass = 0;; ++pass) {   // one pass unless DOFF finds a row sum out of range (see the kernel's header)
#pragma unroll
    for (int db = 0; db < 8; ++db)
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) o[db][qb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) lacc[qb] = f32x4{0.f, 0.f, 0.f, 0.f};
    // ---- prologue: K(0) -> fragments; [K(1), V(0)] and the K pieces of K(2) in flight (what iterations -3 and -2 would have issued)
    {
        for_seq([&](auto pc) B64_INLINE {
            dma_m0(pc, ic<0>{});
            asm volatile("s_nop 0" ::: "memory");
            dma_go(pc, koff(0));
        }, std::make_integer_sequence<int, NKP>{});
        for_seq([&](auto pc) B64_INLINE {
            constexpr int pp = decltype(pc)::value;
            if constexpr (pp < NKP || pp >= 4) {
                dma_m0(pc, ic<(pp < 4 ? 1 : 0)>{});
                asm volatile("s_nop 0" ::: "memory");
                dma_go(pc, pp < 4 ? koff(1) : voff(0));
            }
        }, std::make_integer_sequence<int, 8>{});
        for_seq([&](auto pc) B64_INLINE {
            dma_m0(pc, ic<2>{});
            asm volatile("s_nop 0" ::: "memory");
            dma_go(pc, koff(2));
        }, std::make_integer_sequence<int, NKP>{});
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"i"(2 * NKP + 4) : "memory");   // the Q fragments and K(0) have landed everywhere
        if constexpr (!QK8) {
            for_seq([&](auto nc) B64_INLINE {
                constexpr int n = decltype(nc)::value;
                B64_USE
                B64_DSR(kf[n], "=a", kfa[n / 4], kimm(n));
            }, std::make_integer_sequence<int, 16>{});
        } else {
            for_seq([&](auto nc) B64_INLINE {
                constexpr int n = decltype(nc)::value, kb = n / 2;
                B64_USE
                if constexpr (n % 2 == 0) B64_DSR(kf8[kb].lo, "=v", kfa[0], kimm8(kb));
                else B64_DSR(kf8[kb].hi, "=v", kfa[1], kimm8(kb));
            }, std::make_integer_sequence<int, 8>{});
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if constexpr (DOFF) {
        if (pass == 0) {   // m_q = max over KV tile 0 of the query's scores (S(0) from zero; iteration -1 computes it again from -m_q)
            for_seq([&](auto gc) B64_INLINE { qk_slot(gc, bc<false>{}); }, std::make_integer_sequence<int, 64>{});
            drain();
            // the compiler's own instructions read S only behind the drain: they hang on these (empty) statements' outputs
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) asm volatile("" : "+v"(s[kb][0]), "+v"(s[kb][1]), "+v"(s[kb][2]), "+v"(s[kb][3]));
#pragma unroll
            for (int qb = 0; qb < 4; ++qb) {
                float mx = s[0][qb][0];
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kb][qb][r]);
                // the query's 64 keys sit in the four lanes c16 + 16 g4: row swaps instead of LDS shuffles
                float a = mx, b = mx;
                asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
                mx = fmaxf(a, b);
                a = mx; b = mx;
                asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
                mx = fmaxf(a, b);
                negm[qb] = f32x4{-mx, -mx, -mx, -mx};
            }
        }
    }

    auto iter = [&](auto qk_c, auto pv_c, auto r_c, int i) B64_INLINE {
        constexpr bool HAS_QK = decltype(qk_c)::value, HAS_PV = decltype(pv_c)::value;
        constexpr int R = decltype(r_c)::value;
        constexpr bool CT = R >= 0;
        constexpr int kslot = CT ? ((R + 2) % 3) * KT : 0;       // K(i+2): fragment reads at gaps 65..95
        constexpr int vslot = CT ? ((R + 1) % 3) * VT : 0;       // V(i+1): gaps 103..135
        constexpr int vslot_prev = CT ? ((R + 3) % 3) * VT : 0;  // (no V^T fragment read crosses the iteration in this schedule)
        const uint32_t ri = (uint32_t)((i + 3) % 3);
        const uint32_t kslot_rt = CT ? 0u : ((ri + 2) % 3) * KT, vslot_rt = CT ? 0u : ((ri + 1) % 3) * VT, vslot_prev_rt = CT ? 0u : ri * VT;
        // V^T pieces (gaps 7..55) of V(i+2) -> slot R+2; K pieces (gaps 71..119) of K(i+4) -> slot R+1
        const uint32_t vA = voff(i + 2), kB = koff(i + 4);
        auto gap = [&](auto gc) B64_INLINE {
            constexpr int g = decltype(gc)::value;
            B64_USE
            (void)vA; (void)kB; (void)ri; (void)kslot_rt; (void)vslot_rt; (void)vslot_prev_rt;
            // ---- waits in front of the MFMA
            if constexpr (g == 0 && HAS_QK) asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");
            if constexpr (g == 60) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // ---- the MFMA
            if constexpr (g < 64) {
                if constexpr (HAS_QK) qk_slot(ic<g>{}, bc<DOFF>{});
            } else if constexpr (HAS_PV) {
                constexpr int H = (g - 64) / 36, st = ((g - 64) % 36) / 18, m = ((g - 64) % 36) % 18;
                if constexpr (m < 16) {
                    constexpr int db = m / 2, qb = 2 * H + m % 2;
                    // (QK8: the e4m3 K / Q operands live in VGPRs — see qk8_mfma — so the V^T fragments take the AGPRs they left)
                    if constexpr (QK8) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(o[db][qb]) : "a"(vf[8 * st + db]), "v"(p[st][qb]));
                    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(o[db][qb]) : "v"(vf[8 * st + db]), "v"(p[st][qb]));
                } else {
                    constexpr int qb = 2 * H + (m - 16);
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(lacc[qb]) : "v"(ones), "v"(p[st][qb]));
                }
            }
            // ---- the last tile (the only iteration without a QK^T part; its gaps 0..63 carry no MFMA): behind the last pack (gap 31) the
            // P^T slots of keys that do not exist are cleared — one v_and per P register, two per gap
            if constexpr (!HAS_QK && HAS_PV && g >= 36 && g < 52) {
                constexpr int n0 = 2 * (g - 36);
                asm volatile("v_and_b32 %0, %0, %1" : "+v"(p[n0 / 16][(n0 / 4) % 4][n0 % 4]) : "v"(pmask[n0 / 16][n0 % 4]));
                asm volatile("v_and_b32 %0, %0, %1" : "+v"(p[(n0 + 1) / 16][((n0 + 1) / 4) % 4][(n0 + 1) % 4]) : "v"(pmask[(n0 + 1) / 16][(n0 + 1) % 4]));
            }
            // ---- fillers, in issue order
            if constexpr (g == BAR_GAP) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"i"(NKP + 4) : "memory");
            for_seq([&](auto ec) B64_INLINE {
                constexpr int e = decltype(ec)::value;
                B64_USE
                constexpr int st = e / 16, qh = (e / 8) % 2, u = (e / 4) % 2, r = e % 4;
                // blocks 0, 1 (tile i+1)
                if constexpr (HAS_QK && x0_gap(e) == g) asm volatile("v_exp_f32 %0, %0" : "+v"(s[2 * st + u][qh][r]));
                // blocks 2, 3: first part (tile i+1), second part (tile i, started by the previous iteration)
                if constexpr (HAS_QK && x1_gap(e) == g) asm volatile("v_exp_f32 %0, %0" : "+v"(s[2 * st + u][2 + qh][r]));
                if constexpr (HAS_PV && x1_gap(e) == g + T) asm volatile("v_exp_f32 %0, %0" : "+v"(s[2 * st + u][2 + qh][r]));
            }, std::make_integer_sequence<int, 32>{});
            // bf16 packs: c = st*8 + qh*4 + u*2 + rp -> p[st][qb][u*2 + rp] <- s[2 st + u][qb][2 rp], [2 rp + 1]
            for_seq([&](auto cc) B64_INLINE {
                constexpr int c = decltype(cc)::value;
                B64_USE
                constexpr int st = c / 8, qh = (c / 4) % 2, u = (c / 2) % 2, rp = c % 2;
                if constexpr (HAS_QK && c0_gap(c) == g)
                    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p[st][qh][u * 2 + rp]) : "v"(s[2 * st + u][qh][2 * rp]), "v"(s[2 * st + u][qh][2 * rp + 1]));
                if constexpr (HAS_PV && c1_gap(c) == g + T)
                    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p[st][2 + qh][u * 2 + rp]) : "v"(s[2 * st + u][2 + qh][2 * rp]), "v"(s[2 * st + u][2 + qh][2 * rp + 1]));
            }, std::make_integer_sequence<int, 16>{});
            // K fragments of tile i+2, V^T fragments of tile i+1
            for_seq([&](auto nc) B64_INLINE {
                constexpr int n = decltype(nc)::value;
                B64_USE (void)kslot_rt; (void)vslot_rt; (void)vslot_prev_rt;
                if constexpr (!QK8 && HAS_QK && kr_gap(n) == g) { const uint32_t a = kfa[n / 4] + kslot_rt; B64_DSR(kf[n], "=a", a, kslot + kimm(n)); }
                if constexpr (QK8 && HAS_QK && n < 8 && kr_gap(n) == g) {   // operand n / 2 (last read at slot 36 + 8 (n / 2) <= 60), half n % 2
                    const uint32_t a = kfa[n % 2] + kslot_rt;
                    if constexpr (n % 2 == 0) B64_DSR(kf8[n / 2].lo, "=v", a, kslot + kimm8(n / 2));
                    else B64_DSR(kf8[n / 2].hi, "=v", a, kslot + kimm8(n / 2));
                }
                if constexpr (HAS_QK && vr_gap(n) == g) {
                    const uint32_t a = vfa[n / 8] + vslot_rt;
                    if constexpr (QK8) B64_DSR(vf[n], "=a", a, vslot + vimm(n)); else B64_DSR(vf[n], "=v", a, vslot + vimm(n));
                }
                if constexpr (HAS_PV && vr_gap(n) == g + T) {
                    const uint32_t a = vfa[n / 8] + vslot_prev_rt;
                    if constexpr (QK8) B64_DSR(vf[n], "=a", a, vslot_prev + vimm(n)); else B64_DSR(vf[n], "=v", a, vslot_prev + vimm(n));
                }
            }, std::make_integer_sequence<int, 16>{});
            // LDS-DMA: M0 two gaps ahead of the piece
            for_seq([&](auto pc) B64_INLINE {
                constexpr int pp = decltype(pc)::value;
                (void)vA; (void)kB; (void)ri;
                if constexpr (HAS_QK && pp < NKP && dma_gap(pp) - 2 == g) {
                    if constexpr (CT) dma_m0(pc, ic<(R + 4) % 3>{});
                    else dma_m0_rt(pc, (ri + 1) % 3);
                }
                if constexpr (HAS_QK && pp < NKP && dma_gap(pp) == g) dma_go(pc, kB);
                if constexpr (HAS_QK && pp >= 4 && dma_gap(pp) - 2 == g + T) {
                    if constexpr (CT) dma_m0(pc, ic<(R + 5) % 3>{});
                    else dma_m0_rt(pc, (ri + 2) % 3);
                }
                if constexpr (HAS_QK && pp >= 4 && dma_gap(pp) == g + T) dma_go(pc, vA);
            }, std::make_integer_sequence<int, 8>{});
        };
        for_seq(gap, std::make_integer_sequence<int, T>{});
    };
    iter(bc<true>{}, bc<false>{}, ic<2>{}, -1);
    drain();
    int i = 0;
    for (; i + 3 < nt; i += 3) {      // three tiles per trip: i mod 3 is compile-time inside
        iter(bc<true>{}, bc<true>{}, ic<0>{}, i);
        iter(bc<true>{}, bc<true>{}, ic<1>{}, i + 1);
        iter(bc<true>{}, bc<true>{}, ic<2>{}, i + 2);
        if (!(i + 6 < nt)) drain();   // last trip: in front of the exit's copies
    }
    for (; i + 1 < nt; ++i) {         // at most three more with run-time slots
        iter(bc<true>{}, bc<true>{}, ic<-1>{}, i);
        drain();
    }
    iter(bc<false>{}, bc<true>{}, ic<-1>{}, i);
    // MFMA result -> first non-MFMA reader; every DMA piece has landed before the workgroup leaves (or starts over)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    // (the accumulators are read by the compiler's instructions only behind that wait: they hang on these empty statements' outputs)
    asm volatile("" : "+a"(lacc[0]), "+a"(lacc[1]), "+a"(lacc[2]), "+a"(lacc[3]));
#pragma unroll
    for (int db = 0; db < 8; ++db) asm volatile("" : "+a"(o[db][0]), "+a"(o[db][1]), "+a"(o[db][2]), "+a"(o[db][3]));
    if constexpr (!DOFF) break;
    else {
        bool bad = false;
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) bad |= !(lacc[qb][0] < 0x1p90f);   // also true for inf / NaN
        if (__any(bad) && lane == 0) s_again = 1;
        __syncthreads();
        const int again = s_again;
        if (!again || pass >= 12) break;   // 12 repeats raise an offset by >= 1200: beyond that the inputs are not finite numbers
        __syncthreads();
        if (t == 0) s_again = 0;
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) {
            const float l = lacc[qb][0];
            if (!(l < 0x1p90f)) {
                const float up = (l < 0x1p127f) ? ceilf(__builtin_amdgcn_logf(l)) : 100.0f;   // v_log_f32 = log2
                negm[qb] = negm[qb] - f32x4{up, up, up, up};
            }
        }
        __syncthreads();
    }
  }
#undef B64_DSR
    if (trace) {
        unsigned long long c1, r1;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1)::"memory");
        if (lane == 0) {
            unsigned long long* dst = trace + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wid) * 2;
            dst[0] = c1 - tr_c0;
            dst[1] = r1 - tr_r0;
        }
    }

    // ---- a piece of a KV-split item: the raw accumulators, [quad index][thread] as 16-byte stores (4 KB per instruction), for the combine kernel
    if (!DOFF && piece_ws != nullptr) {
        f32x4* w4 = reinterpret_cast<f32x4*>(piece_ws) + t;
#pragma unroll
        for (int db = 0; db < 8; ++db)
#pragma unroll
            for (int qb = 0; qb < 4; ++qb) w4[(db * 4 + qb) * 256] = o[db][qb];
        w4[32 * 256] = f32x4{lacc[0][0], lacc[1][0], lacc[2][0], lacc[3][0]};
        return;
    }
    // ---- normalise and store.  o[db][qb][r]: d = db*16 + 4 g4 + r, query = q0 + qb*16 + c16; lacc[qb][*] = the row sum of that query
    if (QK8 && bo.q8 != nullptr) {
        // fp8 models: O as block-scaled e4m3 (kernels.h, GemmProblem::a_bscale) — the 16 queries of a query block x this head's 128
        // columns are one block: maximum over the wave (an LDS maximum per query block: own entries, the wave's LDS operations are
        // ordered), exact power-of-two exponent, bytes + one scale byte; no bf16 copy, no quantisation pass in front of proj / linear2.
        if (lane < 4) bo_max[wid][lane] = 0u;
        float inv4[4];
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) {
            inv4[qb] = 1.0f / lacc[qb][0];
            float mx = 0.f;
#pragma unroll
            for (int db = 0; db < 8; ++db)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, fabsf(o[db][qb][r] * inv4[qb]));
            mx = row16_max_nonneg(mx);
            if ((lane & 15) == 15) atomicMax(&bo_max[wid][qb], __float_as_uint(mx));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) {
            const int e = fp8_block_exp(__uint_as_float(bo_max[wid][qb]));
            const float mult = ldexpf(1.0f, e) * inv4[qb];
            const int qrow = q0 + qb * 16 + c16;
            if (lane == 0 && q0 + qb * 16 < Sq) bo.bs[(size_t)head * bo.bs_ld + ((q0 + qb * 16) >> 4)] = (uint8_t)(127 - e);
            if (qrow < Sq) {
                uint8_t* op = bo.q8 + (size_t)qrow * bo.ld + head * HD + g4 * 4;
#pragma unroll
                for (int db = 0; db < 8; ++db)
                    *reinterpret_cast<uint32_t*>(op + db * 16) =
                        pack_fp8x4_inrange(o[db][qb][0] * mult, o[db][qb][1] * mult, o[db][qb][2] * mult, o[db][qb][3] * mult);
            }
        }
        return;
    }
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) {
        const float inv = 1.0f / lacc[qb][0];
        const int qrow = q0 + qb * 16 + c16;
        if (qrow < Sq) {
            uint16_t* op = out + (size_t)qrow * ldo + head * HD + g4 * 4;
#pragma unroll
            for (int db = 0; db < 8; ++db) {
                uint2 v;
                v.x = pack_bf16x2(o[db][qb][0] * inv, o[db][qb][1] * inv);
                v.y = pack_bf16x2(o[db][qb][2] * inv, o[db][qb][3] * inv);
                *reinterpret_cast<uint2*>(op + db * 16) = v;
            }
        }
    }
}

// One 256-thread workgroup per (split item, wave of the attention's workgroup); its four waves take two of the eight d blocks each, so a
// lane adds 8 accumulator quads + the row sums over the item's f pieces (in part order: bit-reproducible) — sixteen times the waves of one
// workgroup per item, which read 64 items' 35 MB of partials from a quarter of the chip, one dependent accumulation chain per lane
// (20 us per launch measured, 1.9 % of a Klein-9B fp8 step).  Lane (c16, g4) = the attention's; then the attention kernel's own epilogue:
// bf16, or the block-scaled e4m3 of the fp8 models (a block = 16 queries x the head's 128 columns: maximum over the four waves through LDS).
template <bool QK8>
__global__ __launch_bounds__(256) void attn64b_combine_kernel(uint16_t* __restrict__ out, int ldo, int Sq, AttnBlocksOut bo, AttnSplit sp) {
    __shared__ unsigned bo_max[4];        // per query block of this (item, wave)
    const int lane = threadIdx.x & 63, sub = threadIdx.x >> 6, c16 = lane & 15, g4 = lane >> 4;
    const int wid = blockIdx.x & 3, t = wid * 64 + lane;       // wave / thread of the attention's workgroup whose accumulators these are
    const int it = blockIdx.x >> 2, item = sp.n_whole + it;
    const int xcd = item & 7, kk = item >> 3;
    const int head = (kk / sp.qblocks) * 8 + xcd, qblk = kk % sp.qblocks;
    const int q0 = qblk * 256 + wid * 64;
    f32x4 o[2][4], l4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) o[j][qb] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int part = 0; part < sp.f; ++part) {
        const f32x4* w4 = reinterpret_cast<const f32x4*>(sp.ws + (size_t)(it * sp.f + part) * SPLIT_PIECE_FLOATS) + t;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int qb = 0; qb < 4; ++qb) o[j][qb] += w4[((2 * sub + j) * 4 + qb) * 256];
        l4 += w4[32 * 256];
    }
    if (QK8 && bo.q8 != nullptr) {
        if (threadIdx.x < 4) bo_max[threadIdx.x] = 0u;
        __syncthreads();
        float inv4[4];
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) {
            inv4[qb] = 1.0f / l4[qb];
            float mx = 0.f;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, fabsf(o[j][qb][r] * inv4[qb]));
            mx = row16_max_nonneg(mx);
            if ((lane & 15) == 15) atomicMax(&bo_max[qb], __float_as_uint(mx));
        }
        __syncthreads();
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) {
            const int e = fp8_block_exp(__uint_as_float(bo_max[qb]));
            const float mult = ldexpf(1.0f, e) * inv4[qb];
            const int qrow = q0 + qb * 16 + c16;
            if (threadIdx.x == 0 && q0 + qb * 16 < Sq) bo.bs[(size_t)head * bo.bs_ld + ((q0 + qb * 16) >> 4)] = (uint8_t)(127 - e);
            if (qrow < Sq) {
                uint8_t* op = bo.q8 + (size_t)qrow * bo.ld + head * HD + g4 * 4;
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    *reinterpret_cast<uint32_t*>(op + (2 * sub + j) * 16) =
                        pack_fp8x4_inrange(o[j][qb][0] * mult, o[j][qb][1] * mult, o[j][qb][2] * mult, o[j][qb][3] * mult);
            }
        }
        return;
    }
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) {
        const float inv = 1.0f / l4[qb];
        const int qrow = q0 + qb * 16 + c16;
        if (qrow < Sq) {
            uint16_t* op = out + (size_t)qrow * ldo + head * HD + g4 * 4;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                uint2 v;
                v.x = pack_bf16x2(o[j][qb][0] * inv, o[j][qb][1] * inv);
                v.y = pack_bf16x2(o[j][qb][2] * inv, o[j][qb][3] * inv);
                *reinterpret_cast<uint2*>(op + (2 * sub + j) * 16) = v;
            }
        }
    }
}

}  // namespace

// 0 launched / 1 not this kernel's case (the caller falls back) / 2 launch error.
// Requires: q pre-multiplied by scale*log2(e); data_offset = false: the fixed zero offset valid (the caller checks), true: any scores;
// vt finite beyond key S - 1 (up to s_pad): those P^T slots are zero, the values themselves are not used.
// The four instantiations (softmax offset form x bf16 / e4m3 Q, K) are separate translation units — attention64b_doff.hip,
// attention64b_qk8.hip, attention64b_qk8_doff.hip include this file with ATTN64B_DOFF / ATTN64B_QK8 set: each takes hipcc ~2.5 minutes,
// side by side in a parallel build.
#ifndef ATTN64B_DOFF
#define ATTN64B_DOFF 0
#endif
#ifndef ATTN64B_QK8
#define ATTN64B_QK8 0
#endif
#define A64B_CAT2(a, b, c) a##b##c
#define A64B_CAT(a, b, c) A64B_CAT2(a, b, c)
#define A64B_VARIANT A64B_CAT(launch_attention64b_v, ATTN64B_DOFF, ATTN64B_QK8)
// q, k: bf16 rows of ldqk elements, or (QK8) e4m3 rows of ldqk bytes with qk8_exps = (127 - qe) | (127 - ke) << 8 (q8 = q 2^qe, k8 = k 2^ke)
int attention64b_split_factor(int Sq, int S, int heads, int n_cu);
int launch_attention64b_v00(hipStream_t, const void*, const void*, int, const uint16_t*, int, uint16_t*, int, int, int, int, int, uint8_t*, int, uint8_t*, int, int, float*);
int launch_attention64b_v10(hipStream_t, const void*, const void*, int, const uint16_t*, int, uint16_t*, int, int, int, int, int, uint8_t*, int, uint8_t*, int, int, float*);
int launch_attention64b_v01(hipStream_t, const void*, const void*, int, const uint16_t*, int, uint16_t*, int, int, int, int, int, uint8_t*, int, uint8_t*, int, int, float*);
int launch_attention64b_v11(hipStream_t, const void*, const void*, int, const uint16_t*, int, uint16_t*, int, int, int, int, int, uint8_t*, int, uint8_t*, int, int, float*);

int A64B_VARIANT(hipStream_t stream, const void* q, const void* k, int ldqk, const uint16_t* vt, int s_pad, uint16_t* out, int ldo, int Sq, int S,
                 int heads, int qk8_exps, uint8_t* out8, int ld8o, uint8_t* bs, int bs_ld, int n_cu, float* split_ws) {
    if (out8 && (!ATTN64B_QK8 || !bs || (ld8o & 3) || (reinterpret_cast<uintptr_t>(out8) & 3) || bs_ld < (Sq + 15) / 16)) return 1;
    const AttnBlocksOut bo{out8, ld8o, bs, bs_ld};
    if (S < 1 || Sq <= 0 || s_pad < ((S + BKV - 1) / BKV) * BKV) return 1;
    const unsigned long long esz = ATTN64B_QK8 ? 1 : 2;
    if ((unsigned long long)S * (unsigned long long)ldqk * esz >= (1ull << 31) || (unsigned long long)HD * (unsigned long long)s_pad * 2 >= (1ull << 31)) return 1;
    if (ATTN64B_QK8 && ((ldqk & 15) || (reinterpret_cast<uintptr_t>(q) & 15) || (reinterpret_cast<uintptr_t>(k) & 15))) return 1;
    dim3 grid((Sq + 255) / 256, heads);
    const dim3 block(256);
    // KV-split of the partial last round (AttnSplit): only the fixed-offset form, heads % 8 == 0 (the 1-D launch keeps the XCD-aware item
    // order), a tail of at most half the CUs (each tail item then splits into f >= 2 equal pieces that still fit ONE round) and pieces of
    // at least 8 KV tiles (shorter ones are all prologue).  split_ws: attention64b_split_ws_bytes(n_cu) bytes, or nullptr = never split.
    AttnSplit sp{0, 1, (int)grid.x, nullptr};
    static const bool split_on = [] {
        const char* e = F2_DEV_ENV("FLUX2MI_ATTN_SPLIT");
        return e ? atoi(e) != 0 : true;
    }();
    if (!ATTN64B_DOFF && split_on && split_ws != nullptr) {
        const int f = attention64b_split_factor(Sq, S, heads, n_cu);
        if (f >= 2) {
            const int items = (int)grid.x * heads, tail = items % n_cu;
            sp.n_whole = items - tail;
            sp.f = f;
            sp.ws = split_ws;
            grid = dim3(sp.n_whole + tail * f, 1);
        }
    }
    unsigned long long* trace = nullptr;
    static const char* trace_env = F2_DEV_ENV("FLUX2MI_ATTN_TRACE");
    static int seen = 0;
    const size_t tn = (size_t)grid.x * grid.y * 4 * 2;
    if (trace_env && ++seen % 8 == 0 && hipMalloc((void**)&trace, tn * 8) != hipSuccess) trace = nullptr;
    hipLaunchKernelGGL((attn64b_kernel<(ATTN64B_DOFF != 0), (ATTN64B_QK8 != 0)>), grid, block, 0, stream, reinterpret_cast<const uint16_t*>(q),
                       reinterpret_cast<const uint16_t*>(k), ldqk, vt, s_pad, out, ldo, Sq, S, trace, qk8_exps, bo, sp);
    if (sp.f > 1) {
        const int tail = ((int)grid.x - sp.n_whole) / sp.f;
        hipLaunchKernelGGL((attn64b_combine_kernel<(ATTN64B_QK8 != 0)>), dim3(tail * 4), dim3(256), 0, stream, out, ldo, Sq, bo, sp);
    }
    if (trace) {
        std::vector<unsigned long long> h(tn);
        (void)hipStreamSynchronize(stream);
        (void)hipMemcpy(h.data(), trace, tn * 8, hipMemcpyDeviceToHost);
        (void)hipFree(trace);
        double cyc = 0, rt = 0;
        for (size_t w = 0; w < tn / 2; ++w) { cyc += (double)h[2 * w]; rt += (double)h[2 * w + 1]; }
        const double tiles = (double)(tn / 2) * ((S + BKV - 1) / BKV + 1);
        fprintf(stderr, "[attn64b trace%s%s] S %d heads %d: %.0f shader cycles per KV tile (136 MFMAs = 2176; e4m3 QK^T: 88 MFMAs = 1664), in-loop clock %.0f MHz, %.3f us per tile\n",
                ATTN64B_DOFF ? ", data offset" : "", ATTN64B_QK8 ? ", e4m3 Q K" : "", S, heads, cyc / tiles, cyc / rt * 100.0, rt / tiles * 0.01);
    }
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

#if !ATTN64B_DOFF && !ATTN64B_QK8
size_t attention64b_split_ws_bytes(int n_cu) { return (size_t)(n_cu > 0 ? n_cu : 0) * SPLIT_PIECE_FLOATS * 4; }
// Pieces per tail item the launcher cuts (1 = no split): heads % 8 == 0 (the 1-D launch keeps the XCD-aware item order), more than one
// round of items, a tail of at most half the CUs (f = n_cu / tail >= 2 pieces per item still fit ONE round), pieces of >= 8 KV tiles
// (shorter ones are all prologue), at most 8 per item.
int attention64b_split_factor(int Sq, int S, int heads, int n_cu) {
    if (n_cu < 16 || (heads & 7) != 0 || Sq <= 0 || S <= 0) return 1;
    const int items = ((Sq + 255) / 256) * heads, tail = items % n_cu, nt = (S + BKV - 1) / BKV;
    if (items <= n_cu || tail == 0) return 1;
    int f = n_cu / tail;
    if (f > nt / 8) f = nt / 8;
    if (f > 8) f = 8;
    return f >= 2 ? f : 1;
}

int launch_attention64b(hipStream_t stream, const uint16_t* q, const uint16_t* k, int ldqk, const uint16_t* vt, int s_pad, uint16_t* out,
                        int ldo, int Sq, int S, int heads, bool data_offset, int n_cu, float* split_ws) {
    return data_offset ? launch_attention64b_v10(stream, q, k, ldqk, vt, s_pad, out, ldo, Sq, S, heads, 0, nullptr, 0, nullptr, 0, n_cu, split_ws)
                       : launch_attention64b_v00(stream, q, k, ldqk, vt, s_pad, out, ldo, Sq, S, heads, 0, nullptr, 0, nullptr, 0, n_cu, split_ws);
}
int launch_attention64b_qk8(hipStream_t stream, const uint8_t* q8, const uint8_t* k8, int ld8, int q_exp, int k_exp, const uint16_t* vt, int s_pad,
                            uint16_t* out, int ldo, int Sq, int S, int heads, bool data_offset, uint8_t* out8, int ld8o, uint8_t* bs, int bs_ld,
                            int n_cu, float* split_ws) {
    if (q_exp < -60 || q_exp > 60 || k_exp < -60 || k_exp > 60) return 1;   // negative: operands larger than e4m3's 448 were scaled DOWN
    const int exps = (127 - q_exp) | ((127 - k_exp) << 8);
    return data_offset ? launch_attention64b_v11(stream, q8, k8, ld8, vt, s_pad, out, ldo, Sq, S, heads, exps, out8, ld8o, bs, bs_ld, n_cu, split_ws)
                       : launch_attention64b_v01(stream, q8, k8, ld8, vt, s_pad, out, ldo, Sq, S, heads, exps, out8, ld8o, bs, bs_ld, n_cu, split_ws);
}
#endif
