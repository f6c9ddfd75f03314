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

namespace {

constexpr int BQ = 128, BKV = 64, HD = 128;
constexpr int K_TILE_BYTES = BKV * HD * 2;   // 16 KB
constexpr int V_TILE_BYTES = HD * BKV * 2;   // 16 KB
constexpr int STAGE_BYTES = K_TILE_BYTES + V_TILE_BYTES;

// CAUSAL (text encoder, Qwen3Model.swift:176-212): query row i sees keys j <= i and j < valid_len
// (right padding); kv_rep query heads share one K/V head (GQA, Qwen3Attention.swift:135-147).
template <bool CAUSAL>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const uint16_t* __restrict__ q,
                                                          const uint16_t* __restrict__ k, int ldqk,
                                                          const uint16_t* __restrict__ vt, int s_pad,
                                                          uint16_t* __restrict__ out, int ldo, int Sq, int S,
                                                          float scale_log2e, int valid_len, int kv_rep) {
    // Sq query rows (q / out already point at the first of them), S keys (k / vt from key 0)
    __shared__ __attribute__((aligned(16))) uint8_t smem[2 * STAGE_BYTES];

    const int head = blockIdx.y;
    const int q0 = blockIdx.x * BQ;
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
    int k_row[4], k_chunk[4], k_lds[4], v_lds[4];
    const uint16_t* v_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = t + 256 * i;
        k_row[i] = idx >> 4;
        k_chunk[i] = idx & 15;
        k_lds[i] = k_row[i] * 256 + ((k_chunk[i] ^ (k_row[i] & 15)) << 4);
        const int vr = idx >> 3, vc = idx & 7;
        v_src[i] = vbase + (size_t)vr * s_pad + vc * 8;
        v_lds[i] = K_TILE_BYTES + vr * 128 + ((vc ^ ((vr >> 1) & 7)) << 4);
    }
    u32x4 kreg[4], vreg[4];
    // K rows are addressed by a per-thread pointer that advances one tile per iteration; only a
    // tile that reaches past S (the last one) pays for the row clamp.
    const uint16_t* k_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) k_src[i] = kbase + (size_t)min(k_row[i], S - 1) * ldqk + k_chunk[i] * 8;
    const size_t k_step = (size_t)BKV * ldqk;
#define ATTN_LOAD_TILE(kt_)                                                                              \
    if (((kt_) + 1) * BKV <= S) {                                                                        \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                               \
            kreg[i_] = *reinterpret_cast<const u32x4*>(k_src[i_] + (size_t)(kt_) * k_step);              \
            vreg[i_] = *reinterpret_cast<const u32x4*>(v_src[i_] + (kt_) * BKV);                         \
        }                                                                                                \
    } else {                                                                                             \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                               \
            const int kr_ = min((kt_) * BKV + k_row[i_], S - 1);                                         \
            kreg[i_] = *reinterpret_cast<const u32x4*>(kbase + (size_t)kr_ * ldqk + k_chunk[i_] * 8);    \
            vreg[i_] = *reinterpret_cast<const u32x4*>(v_src[i_] + (kt_) * BKV);                         \
        }                                                                                                \
    }
#define ATTN_STORE_TILE(buf_)                                                                            \
    {                                                                                                    \
        uint8_t* base_ = smem + (buf_) * STAGE_BYTES;                                                    \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                               \
            *reinterpret_cast<u32x4*>(base_ + k_lds[i_]) = kreg[i_];                                     \
            *reinterpret_cast<u32x4*>(base_ + v_lds[i_]) = vreg[i_];                                     \
        }                                                                                                \
    }

    f32x16 o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // keys this block can see at all, and the per-lane limit of the causal form
    const int key_lim_blk = CAUSAL ? min(min(valid_len, S), q0 + BQ) : S;
    const int key_lim = CAUSAL ? min(min(valid_len, S), qrow + 1) : S;
    const int wave_lim = CAUSAL ? min(min(valid_len, S), q0 + wid * 32 + 1) : S;   // smallest limit in the wave
    const int nt = (key_lim_blk + BKV - 1) / BKV;
    ATTN_LOAD_TILE(0)
    ATTN_STORE_TILE(0)
    // The Q fragments come from global loads issued before the loop.  Without this, hipcc's
    // waitcnt pass carries "qf may still be in flight" around the back edge and puts
    // s_waitcnt vmcnt(7..0) in front of the eight QK^T MFMAs of EVERY tile — which also drains the
    // tile t+1 prefetch issued at the top of the iteration, so its latency was never hidden.
    // Passing qf through an empty asm makes the values asm-defined: the wait lands here, once.
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) asm volatile("" : "+v"(qf[ks]));
    __syncthreads();

    for (int kt = 0; kt < nt; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nt) { ATTN_LOAD_TILE(kt + 1) }
        const uint8_t* sk = smem + buf * STAGE_BYTES;
        const uint8_t* sv = sk + K_TILE_BYTES;

        // ---- S^T = K Q^T  (A = K rows, B = Q^T)
        f32x16 sa[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const int row = kb * 32 + l31;
            const uint8_t* rp = sk + row * 256;
            {   // k-step 0 starts the chain from an inline-constant zero accumulator (no 16 v_mov)
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(rp + (((0 * 2 + hi) ^ (row & 15)) << 4));
                sa[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[0], f32x16{}, 0, 0, 0);
            }
#pragma unroll
            for (int ks = 1; ks < 8; ++ks) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(rp + (((ks * 2 + hi) ^ (row & 15)) << 4));
                sa[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sa[kb], 0, 0, 0);
            }
        }

        // ---- online softmax (base-2), per query = per lane pair (lane, lane^32).
        // VALU budget matters as much as MFMA here (the wave issues both in order): the raw
        // scores are only max-reduced; scale and running max go into one fma per element in
        // front of exp2; O and l are rescaled only when some lane's running max grew (exact:
        // alpha == 1 otherwise), which after the first tiles is rare.
        const bool tail = (kt + 1) * BKV > wave_lim;
        if (tail) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kt * BKV + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (key >= key_lim) sa[kb][r] = -INFINITY;
                }
        }
        // 32 scores -> 16 x v_max3_f32 (asm: hipcc would put a canonicalising v_max in front of
        // every fmaxf that reads an MFMA result)
        float mx;
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mx) : "v"(sa[0][0]), "v"(sa[0][1]), "v"(sa[1][0]));
        asm("v_max3_f32 %0, %0, %1, %2" : "+v"(mx) : "v"(sa[1][1]), "v"(sa[0][2]));
#pragma unroll
        for (int r = 3; r < 16; r += 2) asm("v_max3_f32 %0, %0, %1, %2" : "+v"(mx) : "v"(sa[0][r]), "v"(sa[0][r + (r < 15 ? 1 : 0)]));
#pragma unroll
        for (int r = 2; r < 16; r += 2) asm("v_max3_f32 %0, %0, %1, %2" : "+v"(mx) : "v"(sa[1][r]), "v"(sa[1][r + 1]));
        mx = fmaxf(mx, __shfl_xor(mx, 32)) * scale_log2e;     // scale > 0 commutes with max
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
        float rs = 0.f;
        bf16x8 pf[2][2];
        const float neg_m = -m_run;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(sa[kb][s2 * 8 + j], scale_log2e, neg_m));
                    rs += pv;
                    pf[kb][s2][j] = (__bf16)pv;
                }
        l_run += rs;

        // ---- O^T += V^T P^T  (A = V^T rows d, B = P^T)
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            const int row = db * 32 + l31;
            const uint8_t* rp = sv + row * 128;
            const int sw = (row >> 1) & 7;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const bf16x8 vf = *reinterpret_cast<const bf16x8*>(rp + (((kb * 4 + s2 * 2 + hi) ^ sw) << 4));
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[kb][s2], o[db], 0, 0, 0);
                }
        }

        if (kt + 1 < nt) ATTN_STORE_TILE(buf ^ 1)
        __syncthreads();
    }

    // ---- normalise and store.  o[db][r]: d = db*32 + (r&3) + 8*(r>>2) + 4*hi, query = l31
    const float l_tot = l_run + __shfl_xor(l_run, 32);
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

}  // namespace

int launch_attention(hipStream_t stream, const uint16_t* q, const uint16_t* k, int ldqk, const uint16_t* vt,
                     int s_pad, uint16_t* out, int ldo, int Sq, int S, int heads) {
    if (Sq <= 0 || S <= 0 || heads <= 0 || (ldqk & 7) || (ldo & 3) || s_pad < attn_s_pad(S) || (s_pad & 7)) return 1;
    const float scale_log2e = 1.4426950408889634f / sqrtf((float)HD);
    dim3 grid((Sq + BQ - 1) / BQ, heads);
    hipLaunchKernelGGL(attn_fwd_kernel<false>, grid, dim3(256), 0, stream, q, k, ldqk, vt, s_pad, out, ldo, Sq, S,
                       scale_log2e, S, 1);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

int launch_attention_causal(hipStream_t stream, const uint16_t* q, const uint16_t* k, int ldqk, const uint16_t* vt,
                            int s_pad, uint16_t* out, int ldo, int S, int valid_len, int heads, int kv_heads) {
    if (S <= 0 || heads <= 0 || kv_heads <= 0 || heads % kv_heads || valid_len < 1 || valid_len > S || (ldqk & 7) ||
        (ldo & 3) || s_pad < attn_s_pad(S) || (s_pad & 7))
        return 1;
    const float scale_log2e = 1.4426950408889634f / sqrtf((float)HD);
    dim3 grid((S + BQ - 1) / BQ, heads);
    hipLaunchKernelGGL(attn_fwd_kernel<true>, grid, dim3(256), 0, stream, q, k, ldqk, vt, s_pad, out, ldo, S, S,
                       scale_log2e, valid_len, heads / kv_heads);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
