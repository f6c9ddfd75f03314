// vae_attention.hip — streaming (flash-style) single-head attention of the VAE mid block: softmax(Q K^T / sqrt(C)) V with C = 512
// (standard decoder / encoder) or 384 (small decoder) over P = h*w latent pixels, the P x P scores never in HBM.
// Reference: AttentionBlock.callAsFunction, /root/reference/Sources/Flux2Core/VAE/ResnetBlock.swift:281-313 (softmax(matmul(q, k) * scale)
// then matmul(attnWeights, v), one head of width C).  The three-launch form it replaces (scores GEMM -> row softmax -> P V GEMM, plus a
// V transpose) wrote P*P*4 + P*P*2 bytes: 1.5 GiB at 1024^2, 25.8 GiB at 2048^2.
//
// Shape of the kernel (gfx950, one wave per SIMD):
//   * workgroup = 4 waves x 16 queries; a wave keeps its 16 queries' Q as MFMA B fragments (HD/32 x 4 VGPRs) and O^T for them as
//     HD/16 accumulator quads (HD 512: 128 registers) — head dim 512 leaves no room for more than 16 queries per wave;
//   * keys in tiles of 32: K [32 keys][HD] and V^T [HD][32 keys] of a tile are staged ONCE per workgroup into LDS by LDS-DMA
//     (global_load_lds_dwordx4, 1 KB per wave instruction, no VGPR round trip).  K and V^T have their own two-slot rings and their own
//     hand-over (barrier) per tile, so that each operand is requested a tile and a half before its first read: the loop is bound by
//     the latency of the DMA, not by any rate (profiles/NOTEBOOK.md, round 6);
//   * both products swapped so that a probability never leaves its lane: S^T = K Q^T (v_mfma_f32_16x16x32_bf16, A = K fragment from
//     LDS, B = Q), lane (g, q) of the result holds keys of query q only; the A rows of key block kb are ASSIGNED keys
//     8 (row >> 2) + 4 kb + (row & 3), so the lane's 4 + 4 results are keys 8g .. 8g + 7 in order — exactly the B fragment (k = 8g + j)
//     of O^T += V^T P^T, whose A fragment is a plain 16-byte LDS read of V^T;
//   * LDS images are conflict-free for ds_read_b128 (MI355X_MICROARCH.md "LDS": 16-lane groups, bank = 16-byte chunk mod 16):
//     K row-major with chunk' = chunk ^ swz(key), swz = (key & 3) | ((key >> 3) & 3) << 2 (= the A row of that key), realised by the
//     SOURCE address of each DMA lane; V^T fragment-major (block cb = 16 columns: chunk 16 g + r = column 16 cb + r, keys 8g .. 8g+7);
//   * softmax in the log2 domain with a FIXED per-query offset m = the maximum of the query's scores against the first key tile: the
//     tile loop then has no rescale and no cross-lane traffic at all (numerators 2^(s - m) are floating point: a later, larger score
//     only makes them larger than 1, relative precision unchanged), each lane just tracks its running maximum.  Only if some query of
//     the workgroup ends more than 2^60 above its offset (f32 would run out of exponent) the workgroup runs its loop a second time
//     with the exact maxima as offsets — exact for any scores, paid only by inputs that need it.  Row sums are the sums of the
//     bf16-ROUNDED numerators, i.e. of what P V multiplies;
//   * the value bias is added at the end (softmax rows sum to 1): V^T comes from a role-swapped GEMM without a bias epilogue.
// MFMA work 4 P^2 HD.  Per tile and wave: 65 MFMAs (1040 cycles), 64 ds_read_b128 (all four waves read the whole K and V^T tile: 256 KB
// per tile and CU = 1024 cycles of the LDS), 64 KB of L2 -> LDS delivery, ~290 instructions of a wave that is alone on its SIMD: four
// limits of the same size, each reached at about a third — the phases of a tile run one after the other in ONE wave per SIMD (364
// registers: no second wave).  Measured 0.27-0.31 of the 2.5 PF peak.
#include "dev.h"
#include "kernels.h"
#include <utility>

namespace {

constexpr int VA_BN = 32;                         // keys per tile
constexpr int VA_NW = 4;                          // waves per workgroup
constexpr int VA_BM = 16 * VA_NW;                 // queries per workgroup
constexpr float VA_HEADROOM = 60.0f;              // log2 of the growth the numerators may take over the sampled offset before the loop is redone

template <int I> using va_ic = std::integral_constant<int, I>;
template <class F, int... I>
__device__ __forceinline__ void va_for(F&& f, std::integer_sequence<int, I...>) {
    (f(va_ic<I>{}), ...);
}
#define VA_INLINE __attribute__((always_inline))
// MFMAs as inline asm with their accumulators tied in place: S^T in VGPRs (read by the softmax right away), O^T and the row sums in
// AGPRs.  (As builtins hipcc selects untied forms around this loop's control flow and copies all 128 accumulators on every tile; the
// "a" constraints also make it allocate AGPRs at all.)  Not volatile: hipcc places the LDS reads and the vector work around them.
// hipcc does not know these statements are MFMAs, so the read-after-MFMA wait states are explicit below (s_nop).
// all eight fragments of a batch "used" by one empty statement: hipcc puts ONE counted wait in front of it (for the youngest of them, the
// next batch's reads stay in flight) and none in front of the MFMAs behind it — left alone it waits for each fragment in front of its own
// MFMA, an issue slot per MFMA for a wave that is alone on its SIMD
#define VA_FENCE8(f_) asm volatile("" : "+v"(f_[0]), "+v"(f_[1]), "+v"(f_[2]), "+v"(f_[3]), "+v"(f_[4]), "+v"(f_[5]), "+v"(f_[6]), "+v"(f_[7]))
#define VA_MFMA_NEW(d_, a_, b_) asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(d_) : "v"(a_), "v"(b_))
#define VA_MFMA_V(d_, a_, b_) asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(d_) : "v"(a_), "v"(b_))
#define VA_MFMA_A(d_, a_, b_) asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(d_) : "v"(a_), "v"(b_))

template <int HD>
__global__ __launch_bounds__(64 * VA_NW, 1) void vae_attention_kernel(const uint16_t* __restrict__ Q, const uint16_t* __restrict__ K,
                                                                      const uint16_t* __restrict__ Vt, int ldv,
                                                                      const float* __restrict__ vbias, uint16_t* __restrict__ O,
                                                                      int P, float scale_log2e) {
    constexpr int KS = HD / 32;                   // k-steps of Q K^T
    constexpr int CB = HD / 16;                   // column blocks of O^T
    constexpr int KROW = 1024;                    // LDS pitch of a key row (HD 384: 768 bytes used)
    constexpr int KT = VA_BN * KROW;              // 32 KB
    constexpr int VT = CB * 1024;                 // 32 KB (HD 384: 24 KB)
    constexpr int STG = KT + VT;
    constexpr int KPW = VA_BN / VA_NW;            // key rows a wave brings per tile (8)
    constexpr int CPW = CB / VA_NW;               // V^T blocks a wave brings per tile (8 / 6)
    constexpr int NKB = KS / 4;                   // fragment batches of Q K^T: 4 k-steps x 2 key blocks = 8 reads each
    constexpr int NVB = CB / 8;                   // fragment batches of P V: 8 column blocks each
    static_assert(HD % 128 == 0 && CB % VA_NW == 0 && VA_BN % VA_NW == 0 && CB % 8 == 0, "tile shape");
    __shared__ __attribute__((aligned(1024))) uint8_t smem[2 * STG];   // K slots 0, 1 (2 x KT), then V^T slots 0, 1 (2 x VT)

    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, r = lane & 15;
    const int q0 = blockIdx.x * VA_BM + wid * 16;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem;
    const int nt = (P + VA_BN - 1) / VA_BN;

    // ---- Q fragments: B operand, lane (g, r) holds Q[q0 + r][32 ks + 8 g .. + 7]
    bf16x8 qf[KS];
    {
        const int qr = min(q0 + r, P - 1);
        const uint16_t* src = Q + (size_t)qr * HD + 8 * g;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(src + 32 * ks);
    }

    // ---- DMA of a tile's K / V^T into a ring slot: wave w brings key rows KPW w .. and V^T blocks CPW w ..; three instructions per piece
    // (M0 <- LDS destination, the hazard nop, the DMA): per-lane offsets fixed, one scalar base per operand and tile.
    // K row k of the tile: LDS chunk `lane` <- global chunk lane ^ swz(k); HD 384: lanes 48..63 fill the unused quarter with a valid chunk
    uint32_t vo_k[KPW], vo_v[CPW];
#pragma unroll
    for (int j = 0; j < KPW; ++j) {
        const int k = wid * KPW + j;
        uint32_t ch = (uint32_t)lane ^ (uint32_t)((k & 3) | (((k >> 3) & 3) << 2));
        if (HD < 512 && ch >= (uint32_t)(HD / 8)) ch -= 16;
        vo_k[j] = ch * 16u + (uint32_t)j * (HD * 2);
    }
#pragma unroll
    for (int j = 0; j < CPW; ++j) vo_v[j] = ((uint32_t)(16 * j + r) * (uint32_t)ldv + 8u * (uint32_t)g) * 2u;
    const uint8_t* k_src = reinterpret_cast<const uint8_t*>(K) + (size_t)(wid * KPW) * (HD * 2);               // + tile * 32 rows
    const uint8_t* v_src = reinterpret_cast<const uint8_t*>(Vt) + (size_t)(16 * wid * CPW) * (size_t)ldv * 2;   // + tile * 64 bytes
    const uint32_t m0_k = lds_base + (uint32_t)(wid * KPW) * KROW, m0_v = lds_base + 2 * KT + (uint32_t)(wid * CPW) * 1024u;
#define VA_DMA(vo_, sb_, m0_, off_) \
    asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(vo_), "s"(sb_), "s"(m0_), "i"(off_) : "memory", "scc")
    // K has ceil32(P) readable rows (launch_vae_attention's contract): every tile's rows exist.  K slots [0, 2 KT), V^T slots behind them.
    auto issue_k = [&](int tile, int slot) VA_INLINE {
        const uint8_t* ks_ = k_src + (size_t)tile * (VA_BN * HD * 2);
        const uint32_t mk = m0_k + (uint32_t)slot * KT;
        va_for([&](auto jc) VA_INLINE { constexpr int j = decltype(jc)::value; (void)vo_k; (void)ks_; (void)mk; VA_DMA(vo_k[j], ks_, mk, j * KROW); }, std::make_integer_sequence<int, KPW>{});
    };
    auto issue_v = [&](int tile, int slot) VA_INLINE {
        const uint8_t* vs_ = v_src + (size_t)tile * (VA_BN * 2);
        const uint32_t mv = m0_v + (uint32_t)slot * VT;
        va_for([&](auto jc) VA_INLINE { constexpr int j = decltype(jc)::value; (void)vo_v; (void)vs_; (void)mv; VA_DMA(vo_v[j], vs_, mv, j * 1024); }, std::make_integer_sequence<int, CPW>{});
    };

    // ---- fragment addresses.  K: A row r of key block kb is key 8 (r >> 2) + 4 kb + (r & 3); chunk (4 ks + g) ^ r
    const uint32_t k_row = (uint32_t)(8 * (r >> 2) + (r & 3)) * KROW;
    uint32_t k_x[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) k_x[j] = k_row + ((((uint32_t)(4 * j + g)) ^ (uint32_t)r) << 4);
    const uint32_t v_x = (uint32_t)lane * 16u;

    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;
    asm volatile("" : "+v"(ones));

    f32x4 s0, s1;                // S^T of the tile: lane (g, r): s0[i] = key 8g + i, s1[i] = key 8g + 4 + i, query r (RAW scores)
    bf16x8 fa[8], fb[8];         // two fragment batches in flight
    // reads of batch b of the K tile (4 k-steps x 2 key blocks) / of the V^T tile (8 column blocks)
    auto read_k = [&](const uint8_t* sp, int b, bf16x8 (&f)[8]) VA_INLINE {
#pragma unroll
        for (int e = 0; e < 8; ++e)                             // k-step 4 b + (e >> 1): chunk 16 b + (4 (e >> 1) + g) ^ r; e & 1: key block
            f[e] = *reinterpret_cast<const bf16x8*>(sp + k_x[e >> 1] + b * 256 + (e & 1) * 4 * KROW);
    };
    auto read_v = [&](const uint8_t* sp, int b, bf16x8 (&f)[8]) VA_INLINE {
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = *reinterpret_cast<const bf16x8*>(sp + v_x + (8 * b + i) * 1024);
    };
    // S^T = K Q^T of the K tile staged at `sp`
    auto scores = [&](const uint8_t* sp, auto&& after_first_reads) VA_INLINE {
        read_k(sp, 0, fa);
        __builtin_amdgcn_sched_barrier(0);
        after_first_reads();          // the DMA requests of this hand-over go out under the latency of the first fragment reads
        va_for([&](auto bc_) VA_INLINE {
            constexpr int b = decltype(bc_)::value;
            (void)fa; (void)fb; (void)qf; (void)s0; (void)s1; (void)sp;
            bf16x8 (&cur)[8] = (b & 1) ? fb : fa;
            bf16x8 (&nxt)[8] = (b & 1) ? fa : fb;
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (b + 1 < NKB) read_k(sp, b + 1, nxt);
            __builtin_amdgcn_sched_barrier(0);
            VA_FENCE8(cur);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (b == 0 && i == 0) { VA_MFMA_NEW(s0, cur[0], qf[0]); VA_MFMA_NEW(s1, cur[1], qf[0]); }
                else { VA_MFMA_V(s0, cur[2 * i], qf[4 * b + i]); VA_MFMA_V(s1, cur[2 * i + 1], qf[4 * b + i]); }
            }
        }, std::make_integer_sequence<int, NKB>{});
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 15" : "+v"(s0), "+v"(s1));   // the last two MFMAs' results are in s0 / s1 (the operands tie every reader behind the wait states)
    };
    auto max8 = [&]() VA_INLINE -> float {
        return fmaxf(fmaxf(fmaxf(s0[0], s0[1]), fmaxf(s0[2], s0[3])), fmaxf(fmaxf(s1[0], s1[1]), fmaxf(s1[2], s1[3])));
    };
    auto query_max = [&](float v) VA_INLINE -> float {   // over the 4 lanes (g) of a query
        v = fmaxf(v, __shfl_xor(v, 16));
        return fmaxf(v, __shfl_xor(v, 32));
    };
    // keys of the last tile that do not exist get a score far below everything (their numerators are exactly 0)
    auto mask_last = [&]() VA_INLINE {
        const int kbase = (nt - 1) * VA_BN + 8 * g;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (kbase + i >= P) s0[i] = -3.0e38f;
            if (kbase + 4 + i >= P) s1[i] = -3.0e38f;
        }
    };

    // ---- the offset: the query's maximum over the first key tile (scaled: log2 domain)
    issue_k(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    scores(smem, [] {});
    if (nt == 1) mask_last();
    float m_off = query_max(max8()) * scale_log2e;

    // ---- the tile loop.  K and V^T have their own two-slot rings and their own hand-over per tile, so that each operand is requested
    // a tile and a half before its first read (the loop is bound by the latency of the LDS-DMA, not its rate: with one hand-over per tile —
    // K and V^T requested together, one tile ahead — it ran 6-15 % slower when the requests were issued half a tile later):
    //   top of tile t:  K(t) has landed everywhere, P V(t-1) is done everywhere  -> request V^T(t+1) into V slot (t+1)&1;  Q K^T(t)
    //   mid of tile t:  V^T(t) has landed everywhere, Q K^T(t) is done everywhere -> request K(t+2) into K slot t&1;  softmax, P V(t)
    // Requests in program order: K0 V0 K1 | V1 | K2 | V2 | ...: behind K(t) there are V(t) and K(t+1), behind V(t) there are K(t+1), V(t+1).
    f32x4 acc[CB];
    f32x4 lsum;                  // row sums on the matrix pipe: A = ones gives sum_k P^T[k][q] in every row (the sum of the bf16 numerators)
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
        lsum = f32x4{0.f, 0.f, 0.f, 0.f};
        float mx = -3.0e38f;     // this lane's running maximum (raw scores)
        __syncthreads();         // every slot is free (the sample above / the previous pass)
        issue_k(0, 0);
        issue_v(0, 0);
        if (nt > 1) issue_k(1, 1);
        asm volatile("s_nop 7" ::: "memory");
        for (int t = 0; t < nt; ++t) {
            const int sl = t & 1;
            const bool more = t + 1 < nt;
            // K(t): everything but the V^T(t) and K(t+1) requests behind it
            if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(KPW + CPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"i"(CPW) : "memory");
            __syncthreads();
            scores(smem + sl * KT, [&]() VA_INLINE { if (more) issue_v(t + 1, sl ^ 1); });
            if (t == nt - 1) mask_last();
            mx = fmaxf(mx, max8());
            // numerators 2^(s c - m) as bf16: the B operand (k = 8 g + j) of O^T += V^T P^T
            const float nm = -m_off;
            const uint32_t p0 = pack_bf16x2(__builtin_amdgcn_exp2f(fmaf(s0[0], scale_log2e, nm)), __builtin_amdgcn_exp2f(fmaf(s0[1], scale_log2e, nm)));
            const uint32_t p1 = pack_bf16x2(__builtin_amdgcn_exp2f(fmaf(s0[2], scale_log2e, nm)), __builtin_amdgcn_exp2f(fmaf(s0[3], scale_log2e, nm)));
            const uint32_t p2 = pack_bf16x2(__builtin_amdgcn_exp2f(fmaf(s1[0], scale_log2e, nm)), __builtin_amdgcn_exp2f(fmaf(s1[1], scale_log2e, nm)));
            const uint32_t p3 = pack_bf16x2(__builtin_amdgcn_exp2f(fmaf(s1[2], scale_log2e, nm)), __builtin_amdgcn_exp2f(fmaf(s1[3], scale_log2e, nm)));
            bf16x8 pf = __builtin_bit_cast(bf16x8, u32x4{p0, p1, p2, p3});
            // V^T(t): everything but the K(t+1) and V^T(t+1) requests behind it (the last tile: nothing is behind it)
            if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(KPW + CPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const uint8_t* spv = smem + 2 * KT + sl * VT;
            read_v(spv, 0, fa);
            __builtin_amdgcn_sched_barrier(0);
            if (t + 2 < nt) issue_k(t + 2, sl);
            // the packs (VALU) feed MFMAs that hipcc does not recognise as such: an MFMA issued right behind v_cvt_pk_bf16_f32 read the
            // OLD register contents (measured: row sums of garbage, run to run different) — wait states here, and the row-sum MFMA last
            asm volatile("s_nop 7" : "+v"(pf));
            va_for([&](auto bc_) VA_INLINE {
                constexpr int b = decltype(bc_)::value;
                (void)fa; (void)fb; (void)acc; (void)pf; (void)spv;
                bf16x8 (&cur)[8] = (b & 1) ? fb : fa;
                bf16x8 (&nxt)[8] = (b & 1) ? fa : fb;
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (b + 1 < NVB) read_v(spv, b + 1, nxt);
                __builtin_amdgcn_sched_barrier(0);
                VA_FENCE8(cur);
#pragma unroll
                for (int i = 0; i < 8; ++i) VA_MFMA_A(acc[8 * b + i], cur[i], pf);
            }, std::make_integer_sequence<int, NVB>{});
            VA_MFMA_A(lsum, ones, pf);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");     // the last tile's MFMAs have retired
        if (pass == 1) break;
        // a numerator above 2^60 somewhere in the workgroup: once more with the exact maxima (then every numerator is <= 1)
        const float qm = query_max(mx) * scale_log2e;
        if (!__syncthreads_or(qm - m_off > VA_HEADROOM)) break;
        m_off = qm;
    }

    // ---- epilogue: lane (g, r) holds O^T[16 cb + 4 g + i][query r]; every component of lsum is the row sum of query r
    const float inv = 1.0f / lsum[0];
    if (q0 + r < P) {
        uint16_t* dst = O + (size_t)(q0 + r) * HD + 4 * g;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            const f32x4 b = vbias ? *reinterpret_cast<const f32x4*>(vbias + 16 * cb + 4 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
            const f32x4 o = acc[cb] * inv + b;
            *reinterpret_cast<uint2*>(dst + 16 * cb) = make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
        }
    }
}

}  // namespace

// true when launch_vae_attention takes head dim C
bool vae_attention_supported(int C) { return C == 512 || C == 384; }

// O [P, C] bf16 = softmax(Q K^T * scale) V + vbias, with Q [P, C], K [ceil32(P), C] bf16 row-major (the rows behind P: readable, any values —
// their scores are masked) and V^T [C][ldv] bf16 (ldv >= ceil32(P), the
// columns P .. ceil32(P) - 1 finite — zero; ldv % 8 == 0).  vbias f32 [C] or NULL.  Returns 0 ok / 1 unsupported / 2 launch error.
int launch_vae_attention(hipStream_t stream, const uint16_t* Q, const uint16_t* K, const uint16_t* Vt, int ldv, const float* vbias,
                         uint16_t* O, int P, int C, float scale) {
    if (!vae_attention_supported(C) || P <= 0 || ldv < ((P + VA_BN - 1) / VA_BN) * VA_BN || (ldv & 7)) return 1;
    // 32-bit DMA lane offsets: 15 rows of V^T + a tile; every pointer 16-byte aligned
    if ((unsigned long long)16 * (unsigned long long)ldv * 2ull >= (1ull << 32)) return 1;
    if ((reinterpret_cast<uintptr_t>(Q) | reinterpret_cast<uintptr_t>(K) | reinterpret_cast<uintptr_t>(Vt) | reinterpret_cast<uintptr_t>(O)) & 15) return 1;
    if (vbias && (reinterpret_cast<uintptr_t>(vbias) & 15)) return 1;
    const dim3 grid((P + VA_BM - 1) / VA_BM), block(64 * VA_NW);
    const float sl = scale * 1.4426950408889634f;
    if (C == 512) hipLaunchKernelGGL(vae_attention_kernel<512>, grid, block, 0, stream, Q, K, Vt, ldv, vbias, O, P, sl);
    else hipLaunchKernelGGL(vae_attention_kernel<384>, grid, block, 0, stream, Q, K, Vt, ldv, vbias, O, P, sl);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
