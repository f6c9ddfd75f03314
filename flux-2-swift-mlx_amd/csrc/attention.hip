// attention.hip — flash-attention forward for the DiT's joint [text ; image] sequence.
//
// Replaces MLXFast.scaledDotProductAttention(queries:keys:values:scale:mask: nil)
// (Transformer/Flux2Attention.swift:168-174, Flux2ParallelAttention.swift:104-110):
// softmax(Q K^T / sqrt(128)) V, no mask (pad text tokens are attended), one head per
// blockIdx.y, B = 1.  bf16 operands, f32 scores / softmax statistics / accumulators.
//
// Structure (v1): workgroup = 4 waves = 128 query rows (32 per wave), KV tiles of 64 keys.
//   * swapped product S^T = K Q^T with v_mfma_f32_32x32x16_bf16: the accumulator puts the
//     query on the lane and 16 keys in registers, so the row max / row sum are in-register
//     (+ one cross-half exchange), cdna guide T12.
//   * P^T (bf16) is fed straight back as the B operand of O^T += V^T P^T: a 32x32
//     accumulator's registers 8s..8s+7 are exactly one k-step of the next MFMA
//     (guide §3 "An accumulator tile as the next MFMA's operand").  The k order inside a
//     step is key = 16s + 8(j>>2) + 4h + (j&3); V is therefore stored transposed
//     ([head][d][key]) with the keys of every group of 16 permuted to position 8h+4a+b
//     (done once per block by launch_qk_norm_rope_vt), which makes the A fragment one
//     contiguous ds_read_b128.
//   * K tile [64][128] and V^T tile [128][64] are double-buffered in LDS (64 KB per
//     workgroup, 2 workgroups per CU), XOR-swizzled for conflict-free ds_read_b128;
//     global->register loads of tile t+1 are issued before the MFMAs of tile t and written
//     to LDS after them (one barrier per tile).
#include "dev.h"
#include "kernels.h"
#include <stdlib.h>
#include <algorithm>

#define ATTN_STAMP(acc_)
namespace {

constexpr int BKV = 64, HD = 128;   // BQ = 32 query rows per wave x NW waves
constexpr int K_TILE_BYTES = BKV * HD * 2;   // 16 KB
constexpr int V_TILE_BYTES = HD * BKV * 2;   // 16 KB
constexpr int STAGE_BYTES = K_TILE_BYTES + V_TILE_BYTES;

// CAUSAL (text encoders, Qwen3Model.swift:176-212 / MistralModel.swift:150-195): real tokens are rows
// [pad_lo, valid_len).  A real query row i sees keys pad_lo <= j <= i, j < valid_len.  Rows after
// valid_len (right padding, Klein) see the real keys only.  Rows before pad_lo (left padding, Dev)
// see keys 0..i, all of them pads carrying the reference's finite -1e9 mask, which in f32 swallows
// the scores: uniform attention over those keys — reproduced by scoring them 0.
// kv_rep query heads share one K/V head (GQA, Qwen3Attention.swift:135-147).
// The two lanes l and l ^ 32 hold the two halves of a query's keys: exchange through
// v_permlane32_swap (one VALU op) instead of ds_bpermute (an LDS round trip in the softmax's
// dependent chain).  swap(a, b) with a == b == v leaves a = [lo, lo], b = [hi, hi].  Inline asm: on this
// ROCm the builtin's second result comes back as a copy of the first (checked with a one-wave
// kernel); the s_nop cover the VALU-write -> permlane-read and permlane-write -> VALU-read hazards
// the compiler would otherwise handle.
__device__ __forceinline__ void xhalf_swap(float& a, float& b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float xhalf_max(float v) {
    float a = v, b = v;
    xhalf_swap(a, b);
    return fmaxf(a, b);
}
__device__ __forceinline__ float xhalf_sum(float v) {
    float a = v, b = v;
    xhalf_swap(a, b);
    return a + b;
}

// NW waves per workgroup (4: two workgroups per CU; 8: one workgroup per CU, the K / V^T tile staged
// once for 256 query rows, i.e. half the global loads and LDS stores per wave).
// FIXM: softmax against a FIXED offset instead of a running row maximum.  Valid when the caller can
// bound the scaled scores: q and k come out of the per-head RMSNorm (|q| <= sqrt(128) max|w_q|, RoPE
// is a rotation), so s * scale * log2(e) <= fixed_m by Cauchy-Schwarz; then p = exp2(s c - fixed_m) is
// in (0, 1] for every key, the row maximum, its cross-lane exchange, the rescale branch and their
// serial dependency in front of the first exp2 all disappear (same softmax: any offset cancels in
// O / l).  The caller only asks for it when fixed_m is small enough that p cannot underflow.
// FIXM == 2: Q arrives pre-multiplied by scale * log2(e) and the offset is zero: p = exp2(s), no fma.
// One (head, q-block) item over the KV tiles [kt_begin, kt_end) (clamped to the tiles the item can see).  `part` == nullptr:
// the whole item, output normalised and stored as bf16.  Otherwise a PIECE of a KV-split item: the un-normalised f32
// accumulators and the per-lane (l, m) go to `part` in register order ([64 values + l + m][NW*64 threads], 2 KB per
// store instruction) for attn_combine_kernel.
template <bool CAUSAL, int NW, int FIXM>
__device__ __forceinline__ void attn_item(uint8_t* smem, const uint16_t* __restrict__ q,
                                                          const uint16_t* __restrict__ k, int ldqk,
                                                          const uint16_t* __restrict__ vt, int s_pad,
                                                          uint16_t* __restrict__ out, int ldo, int Sq, int S,
                                                          float scale_log2e, int valid_len, int kv_rep,
                                                          float fixed_m, int pad_lo, const int head, const int q0,
                                                          const int kt_begin, const int kt_end, float* __restrict__ part) {
    // Sq query rows (q / out already point at the first of them), S keys (k / vt from key 0)
    constexpr int BQ = NW * 32;
    constexpr int NT = NW * 64;          // threads
    constexpr int PP = 1024 / NT;        // 16-B pieces of the K tile (and of the V^T tile) per thread
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const int l31 = lane & 31, hi = lane >> 5;

    // ---- Q fragments (B operand of S^T = K Q^T): lane holds Q[q = l31][d = 16ks + 8hi + j]
    const int qrow = q0 + wid * 32 + l31;
    const uint16_t* qp = q + (size_t)min(qrow, Sq - 1) * ldqk + head * HD + hi * 8;
    bf16x8 qf[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16);

    // ---- staging assignment
    const int kv_head = CAUSAL ? head / kv_rep : head;
    const uint16_t* kbase = k + kv_head * HD;
    const uint16_t* vbase = vt + (size_t)kv_head * HD * s_pad;
    // K piece i of a tile = rows idx>>4, 16-B chunk idx&15 (idx = t + 256 i); V^T piece: rows idx>>3, chunk idx&7.
    // Sources are 32-bit byte offsets from the wave-uniform bases (scalar base + vector offset loads):
    // K: tile*64 rows + own row, clamped to the last valid row (a per-lane min of monotonic offsets);
    // V^T: columns of the zero-padded [128][s_pad] image, always in range.
    int k_lds[PP], v_lds[PP];
    uint32_t k_rowoff[PP], k_maxoff[PP], v_off[PP];
    const uint32_t k_row_bytes = (uint32_t)ldqk * 2u;
#pragma unroll
    for (int i = 0; i < PP; ++i) {
        const int idx = t + NT * i;
        const int kr = idx >> 4, kc = idx & 15;
        k_rowoff[i] = (uint32_t)kr * k_row_bytes + kc * 16;
        k_maxoff[i] = (uint32_t)(S - 1) * k_row_bytes + kc * 16;
        k_lds[i] = kr * 256 + ((kc ^ (kr & 15)) << 4);
        const int vr = idx >> 3, vc = idx & 7;
        v_off[i] = ((uint32_t)vr * s_pad + vc * 8) * 2u;
        v_lds[i] = K_TILE_BYTES + vr * 128 + ((vc ^ ((vr >> 1) & 7)) << 4);
    }
    const char* kb8 = reinterpret_cast<const char*>(kbase);
    const char* vb8 = reinterpret_cast<const char*>(vbase);
    u32x4 kreg[PP], vreg[PP];
    // One piece (of 8: 4 K, 4 V^T) of a tile's global->register prefetch.  In the loop the pieces are
    // issued one per K-step BETWEEN the QK^T MFMAs: a wave's global_load waits its turn in the CU's
    // address path (8 waves x 8 loads queue up when they all issue at the top of a tile: s_memtime
    // stamps showed 740-1060 cycles per tile just to ISSUE them), and spread out like this the wait
    // runs under MFMAs that are already in flight.  No branch around them (the tile index is clamped
    // instead), so hipcc's waitcnt pass sees straight-line code and inserts no conservative vmcnt(0).
#define ATTN_LOAD_PIECE(tile_, i_)                                                                       \
    if ((i_) < PP) {                                                                                     \
        const uint32_t off_ = min((uint32_t)(tile_) * (BKV * k_row_bytes) + k_rowoff[(i_) % PP], k_maxoff[(i_) % PP]); \
        kreg[(i_) % PP] = *reinterpret_cast<const u32x4*>(kb8 + off_);                                   \
    } else {                                                                                             \
        vreg[(i_) % PP] = *reinterpret_cast<const u32x4*>(vb8 + (v_off[(i_) % PP] + (uint32_t)(tile_) * (BKV * 2))); \
    }
#define ATTN_STORE_TILE(buf_)                                                                            \
    {                                                                                                    \
        uint8_t* base_ = smem + (buf_) * STAGE_BYTES;                                                    \
        _Pragma("unroll") for (int i_ = 0; i_ < PP; ++i_) {                                              \
            *reinterpret_cast<u32x4*>(base_ + k_lds[i_]) = kreg[i_];                                     \
            *reinterpret_cast<u32x4*>(base_ + v_lds[i_]) = vreg[i_];                                     \
        }                                                                                                \
    }

    f32x16 o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    // CAUSAL: a finite start value, so that a tile whose keys are all masked for a row (left padding
    // inside a straddling workgroup) gives exp2(-inf) = 0 and not exp2(-inf + inf)
    float m_run = FIXM == 2 ? 0.f : FIXM ? fixed_m : (CAUSAL ? -1.0e30f : -INFINITY), l_run = 0.f;

    // keys this block can see at all, and the per-lane limit of the causal form
    const int key_lim_blk = CAUSAL ? min(min(valid_len, S), q0 + BQ) : S;
    const int key_lim = CAUSAL ? min(min(valid_len, S), qrow + 1) : S;
    const int wave_lim = CAUSAL ? min(min(valid_len, S), q0 + wid * 32 + 1) : S;   // smallest limit in the wave
    const int nt = min((key_lim_blk + BKV - 1) / BKV, kt_end);
    // left padding: lower key bound of real rows, "uniform" pad rows, first tile of the workgroup
    const bool pad_row = CAUSAL && qrow < pad_lo;
    const int key_lo = (CAUSAL && !pad_row) ? pad_lo : 0;
    const int key_hi = pad_row ? min(qrow + 1, S) : key_lim;
    const bool wave_lo_mask = CAUSAL && pad_lo > 0;    // some lane of some wave needs the lower bound / the zero scores
    const int kt0 = max((CAUSAL && q0 >= pad_lo) ? pad_lo / BKV : 0, kt_begin);
#pragma unroll
    for (int i = 0; i < 2 * PP; ++i) { ATTN_LOAD_PIECE(kt0, i) }
    ATTN_STORE_TILE(0)
    // The Q fragments come from global loads issued before the loop.  Without this, hipcc's
    // waitcnt pass carries "qf may still be in flight" around the back edge and puts
    // s_waitcnt vmcnt(7..0) in front of the eight QK^T MFMAs of EVERY tile — which also drains the
    // tile t+1 prefetch issued at the top of the iteration, so its latency was never hidden.
    // Passing qf through an empty asm makes the values asm-defined: the wait lands here, once.
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) asm volatile("" : "+v"(qf[ks]));
    __syncthreads();

    for (int kt = kt0; kt < nt; ++kt) {
        const int buf = (kt - kt0) & 1;
        const int ktn = min(kt + 1, nt - 1);   // the last tile re-fetches itself: harmless, keeps the loop branch-free
        const uint8_t* sk = smem + buf * STAGE_BYTES;
        const uint8_t* sv = sk + K_TILE_BYTES;

        // ---- S^T = K Q^T  (A = K rows, B = Q^T): both 32-key halves per K-step (independent
        // accumulators back to back), the K fragments of step ks+1 read before the MFMAs of step ks,
        // one prefetch piece after them.  sched_barrier pins that order.
        f32x16 sa[2];
        const int krow0 = l31 * 256, krow1 = (32 + l31) * 256, ksw = l31 & 15;
        // K fragments run two K-steps ahead of their MFMAs (LDS latency > one MFMA pair)
        bf16x8 kq0[3], kq1[3];
#pragma unroll
        for (int pre = 0; pre < 2; ++pre) {
            kq0[pre] = *reinterpret_cast<const bf16x8*>(sk + krow0 + (((pre * 2 + hi) ^ ksw) << 4));
            kq1[pre] = *reinterpret_cast<const bf16x8*>(sk + krow1 + (((pre * 2 + hi) ^ ksw) << 4));
        }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            if (ks < 6) {
                kq0[(ks + 2) % 3] = *reinterpret_cast<const bf16x8*>(sk + krow0 + ((((ks + 2) * 2 + hi) ^ ksw) << 4));
                kq1[(ks + 2) % 3] = *reinterpret_cast<const bf16x8*>(sk + krow1 + ((((ks + 2) * 2 + hi) ^ ksw) << 4));
            }
            if (ks == 0) {   // the chain starts from an inline-constant zero accumulator (no 32 v_mov)
                sa[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kq0[0], qf[0], f32x16{}, 0, 0, 0);
                sa[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kq1[0], qf[0], f32x16{}, 0, 0, 0);
            } else {
                sa[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kq0[ks % 3], qf[ks], sa[0], 0, 0, 0);
                sa[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kq1[ks % 3], qf[ks], sa[1], 0, 0, 0);
            }
            if (PP == 4) { ATTN_LOAD_PIECE(ktn, ks) }
            else if ((ks & 1) == 0) { ATTN_LOAD_PIECE(ktn, ks >> 1) }   // 4 pieces: every other K-step
            __builtin_amdgcn_sched_barrier(0);
        }

        ATTN_STAMP(dg_qk)
        // ---- online softmax (base-2), per query = per lane pair (lane, lane^32).
        // VALU budget matters as much as MFMA here (the wave issues both in order): the raw
        // scores are only max-reduced; scale and running max go into one fma per element in
        // front of exp2; O and l are rescaled only when some lane's running max grew (exact:
        // alpha == 1 otherwise), which after the first tiles is rare.
        const bool tail = (kt + 1) * BKV > wave_lim || wave_lo_mask;
        if (tail) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kt * BKV + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (key >= key_hi || key < key_lo) sa[kb][r] = -INFINITY;
                    else if (pad_row) sa[kb][r] = 0.f;
                }
        }
        if (!FIXM) {
        // 32 scores -> 16 x v_max3_f32 (asm: hipcc would put a canonicalising v_max in front of
        // every fmaxf that reads an MFMA result)
        float mx;
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mx) : "v"(sa[0][0]), "v"(sa[0][1]), "v"(sa[1][0]));
        asm("v_max3_f32 %0, %0, %1, %2" : "+v"(mx) : "v"(sa[1][1]), "v"(sa[0][2]));
#pragma unroll
        for (int r = 3; r < 16; r += 2) asm("v_max3_f32 %0, %0, %1, %2" : "+v"(mx) : "v"(sa[0][r]), "v"(sa[0][r + (r < 15 ? 1 : 0)]));
#pragma unroll
        for (int r = 2; r < 16; r += 2) asm("v_max3_f32 %0, %0, %1, %2" : "+v"(mx) : "v"(sa[1][r]), "v"(sa[1][r + 1]));
        mx = xhalf_max(mx) * scale_log2e;                     // scale > 0 commutes with max
        if (__builtin_amdgcn_ballot_w64(mx > m_run) != 0) {   // wave-uniform branch
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
        }
        }
        float rs = 0.f;
        bf16x8 pf8[2][2];
        const float neg_m = -m_run;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float pv = FIXM == 2 ? __builtin_amdgcn_exp2f(sa[kb][s2 * 8 + j])
                                               : __builtin_amdgcn_exp2f(__builtin_fmaf(sa[kb][s2 * 8 + j], scale_log2e, neg_m));
                    rs += pv;
                    pf8[kb][s2][j] = (__bf16)pv;
                }
        l_run += rs;
        ATTN_STAMP(dg_sm)

        // ---- O^T += V^T P^T  (A = V^T rows d, B = P^T).  Step order: the four d-blocks of one
        // key sub-block back to back (independent accumulators); the V^T fragments of the next step
        // are read first, and the register->LDS stores of the prefetched tile ride between the
        // steps (buffer buf^1 was last read before the previous tile's barrier).
        uint8_t* stg = smem + (buf ^ 1) * STAGE_BYTES;
        const int vsw = (l31 >> 1) & 7;                 // (db*32 + l31) >> 1 & 7 == (l31 >> 1) & 7
        bf16x8 vc[4], vn[4];
#pragma unroll
        for (int db = 0; db < 4; ++db)
            vc[db] = *reinterpret_cast<const bf16x8*>(sv + (db * 32 + l31) * 128 + (((0 + hi) ^ vsw) << 4));
#pragma unroll
        for (int st4 = 0; st4 < 4; ++st4) {             // st4 = kb*2 + s2: k-step of 16 keys
            if (st4 < 3) {
#pragma unroll
                for (int db = 0; db < 4; ++db)
                    vn[db] = *reinterpret_cast<const bf16x8*>(sv + (db * 32 + l31) * 128 + ((((st4 + 1) * 2 + hi) ^ vsw) << 4));
            }
#pragma unroll
            for (int db = 0; db < 4; ++db)
                o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vc[db], pf8[st4 >> 1][st4 & 1], o[db], 0, 0, 0);
            if (PP == 4) {
                *reinterpret_cast<u32x4*>(stg + k_lds[st4]) = kreg[st4];
                *reinterpret_cast<u32x4*>(stg + v_lds[st4]) = vreg[st4];
            } else if (st4 < 2) {
                *reinterpret_cast<u32x4*>(stg + k_lds[st4 % PP]) = kreg[st4 % PP];
            } else {
                *reinterpret_cast<u32x4*>(stg + v_lds[st4 % PP]) = vreg[st4 % PP];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int db = 0; db < 4; ++db) vc[db] = vn[db];
        }
        ATTN_STAMP(dg_pv)
        __syncthreads();
        ATTN_STAMP(dg_st)
    }
#undef ATTN_LOAD_PIECE

    // ---- a piece of a KV-split item: raw state for attn_combine_kernel
    if (part != nullptr) {
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) part[(size_t)(db * 16 + r) * NT + t] = o[db][r];
        part[(size_t)64 * NT + t] = l_run;
        part[(size_t)65 * NT + t] = m_run;
        return;
    }
    // ---- normalise and store.  o[db][r]: d = db*32 + (r&3) + 8*(r>>2) + 4*hi, query = l31
    const float l_tot = xhalf_sum(l_run);
    const float inv = 1.0f / l_tot;
    if (qrow < Sq) {
        uint16_t* op = out + (size_t)qrow * ldo + head * HD + hi * 4;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 v;
                v.x = pack_bf16x2(o[db][g * 4 + 0] * inv, o[db][g * 4 + 1] * inv);
                v.y = pack_bf16x2(o[db][g * 4 + 2] * inv, o[db][g * 4 + 3] * inv);
                *reinterpret_cast<uint2*>(op + db * 32 + g * 8) = v;
            }
    }
}

template <bool CAUSAL, int NW, int FIXM = 0>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void attn_fwd_kernel(const uint16_t* __restrict__ q,
                                                          const uint16_t* __restrict__ k, int ldqk,
                                                          const uint16_t* __restrict__ vt, int s_pad,
                                                          uint16_t* __restrict__ out, int ldo, int Sq, int S,
                                                          float scale_log2e, int valid_len, int kv_rep,
                                                          float fixed_m, int pad_lo) {
    __shared__ __attribute__((aligned(16))) uint8_t smem[2 * STAGE_BYTES];
    attn_item<CAUSAL, NW, FIXM>(smem, q, k, ldqk, vt, s_pad, out, ldo, Sq, S, scale_log2e, valid_len, kv_rep, fixed_m, pad_lo,
                                blockIdx.y, blockIdx.x * (NW * 32), 0, 1 << 30, nullptr);
}


}  // namespace

int launch_attention64b(hipStream_t stream, const uint16_t* q, const uint16_t* k, int ldqk, const uint16_t* vt, int s_pad, uint16_t* out,
                        int ldo, int Sq, int S, int heads, bool data_offset, int n_cu, float* split_ws);  // attention64b.hip (MFMA 16x16x32)

// Offset = the bound (scores not pre-scaled): p in [2^(-2 bound), 1], bound <= 40 keeps 2^-80 far inside the f32 / bf16 exponent range.
// Offset = 0 on pre-scaled q (the model's form): p in [2^-bound, 2^bound]; with bound <= 80 a row sum stays below 2^80 S and an
// accumulator below 2^80 S max|v| (< 2^100 for S < 2^13, |v| < 2^7), a row of all-minimal scores still sums to a normal 2^-80 S.
bool attention_uses_fixed_offset(float score_bound_log2, bool q_prescaled) {
    static const int fixm_on = [] {   // FLUX2MI_ATTN_FIXM=0 keeps the running maximum everywhere (dev A/B runs)
        const char* e = F2_DEV_ENV("FLUX2MI_ATTN_FIXM");
        return e ? atoi(e) : 1;
    }();
    return fixm_on && score_bound_log2 > 0.f && score_bound_log2 <= (q_prescaled ? 80.f : 40.f);
}

int launch_attention(hipStream_t stream, const uint16_t* q, const uint16_t* k, int ldqk, const uint16_t* vt,
                     int s_pad, uint16_t* out, int ldo, int Sq, int S, int heads, float score_bound_log2, bool q_prescaled,
                     float* split_ws, int n_cu) {
    if (Sq <= 0 || S <= 0 || heads <= 0 || (ldqk & 7) || (ldo & 3) || s_pad < attn_s_pad(S) || (s_pad & 7)) return 1;
    const float scale_log2e = q_prescaled ? 1.0f : 1.4426950408889634f / sqrtf((float)HD);
    static const int nw_env = [] {   // FLUX2MI_ATTN_NW=4|8 forces a workgroup size (A/B runs)
        const char* e = F2_DEV_ENV("FLUX2MI_ATTN_NW");
        return e ? atoi(e) : 0;
    }();
    // 8-wave workgroups (one per CU) when the 4-wave grid would be under two full rounds of the 512
    // workgroup slots (Klein-4B: 864): +0.6 % on the step; 4-wave ones are faster on larger grids
    const int nw = nw_env ? nw_env : (((long)((Sq + 127) / 128) * heads < 1024 && Sq >= 1024) ? 8 : 4);
    const bool fixm = attention_uses_fixed_offset(score_bound_log2, q_prescaled);
    const int fx = fixm ? (q_prescaled ? 2 : 1) : 0;
    const int qblocks = (Sq + nw * 32 - 1) / (nw * 32);
    // pre-scaled Q (the model's form): the one-wave-per-SIMD kernel (attention64b.hip), with the fixed zero offset when the score
    // bound allows it and with its per-query data offset otherwise (the "online" count of flux2mi_dit_attention_softmax_info).
    // FLUX2MI_ATTN64=0 with the dev library keeps the kernel below (A/B runs)
    if (q_prescaled && fx != 1 && !nw_env) {
        static const int a64 = [] { const char* e = F2_DEV_ENV("FLUX2MI_ATTN64"); return e ? atoi(e) : 2; }();   // dev: 0 = the kernel below
        if (a64 != 0) {
            const int r64 = launch_attention64b(stream, q, k, ldqk, vt, s_pad, out, ldo, Sq, S, heads, fx != 2, n_cu, split_ws);
            if (r64 != 1) return r64;
        }
    }
    const dim3 grid(qblocks, heads), block(nw * 64);
#define ATTN_GO(NW_, FX_) hipLaunchKernelGGL((attn_fwd_kernel<false, NW_, FX_>), grid, block, 0, stream, q, k, ldqk, vt, s_pad, out, ldo, Sq, S, scale_log2e, S, 1, score_bound_log2, 0)
    if (nw == 8) { if (fx == 2) ATTN_GO(8, 2); else if (fx) ATTN_GO(8, 1); else ATTN_GO(8, 0); }
    else { if (fx == 2) ATTN_GO(4, 2); else if (fx) ATTN_GO(4, 1); else ATTN_GO(4, 0); }
#undef ATTN_GO
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

int launch_attention_causal(hipStream_t stream, const uint16_t* q, const uint16_t* k, int ldqk, const uint16_t* vt,
                            int s_pad, uint16_t* out, int ldo, int S, int pad_lo, int valid_len, int heads, int kv_heads) {
    if (S <= 0 || heads <= 0 || kv_heads <= 0 || heads % kv_heads || pad_lo < 0 || valid_len <= pad_lo || valid_len > S || (ldqk & 7) ||
        (ldo & 3) || s_pad < attn_s_pad(S) || (s_pad & 7))
        return 1;
    const float scale_log2e = 1.4426950408889634f / sqrtf((float)HD);
    dim3 grid((S + 127) / 128, heads);
    hipLaunchKernelGGL((attn_fwd_kernel<true, 4>), grid, dim3(256), 0, stream, q, k, ldqk, vt, s_pad, out, ldo, S, S,
                       scale_log2e, valid_len, heads / kv_heads, 0.f, pad_lo);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

