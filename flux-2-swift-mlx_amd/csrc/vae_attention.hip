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
//     (global_load_lds_dwordx4, 1 KB per wave instruction, no VGPR round trip), two stages, one barrier per tile: the DMA of tile t + 1
//     runs under the MFMAs of tile t;
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
// MFMA work 4 P^2 HD; per tile and workgroup 64 KB of L2 -> LDS and 256 KB of LDS reads per 1024 MFMA cycles per wave: the kernel is
// bound by those two streams, not the matrix pipe.
#include "dev.h"
#include "kernels.h"

namespace {

constexpr int VA_BN = 32;                         // keys per tile
constexpr int VA_NW = 4;                          // waves per workgroup
constexpr int VA_BM = 16 * VA_NW;                 // queries per workgroup
constexpr float VA_HEADROOM = 60.0f;              // log2 of the growth the numerators may take over the sampled offset before the loop is redone

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
    static_assert(HD % 128 == 0 && CB % VA_NW == 0 && VA_BN % VA_NW == 0, "tile shape");
    __shared__ __attribute__((aligned(1024))) uint8_t smem[2 * STG];

    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, r = lane & 15;
    const int q0 = blockIdx.x * VA_BM + wid * 16;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem;

    // ---- Q fragments: B operand, lane (g, r) holds Q[q0 + r][32 ks + 8 g .. + 7]
    bf16x8 qf[KS];
    {
        const int qr = min(q0 + r, P - 1);
        const uint16_t* src = Q + (size_t)qr * HD + 8 * g;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(src + 32 * ks);
    }

    // ---- DMA of one key tile into stage `st`: wave w brings key rows 8w .. 8w+7 and V^T blocks (CB/4) w .. + CB/4 - 1
    const uint32_t lane16 = (uint32_t)lane * 16u;
    const uint32_t v_voff = ((uint32_t)r * (uint32_t)ldv + 8u * (uint32_t)g) * 2u;
    auto issue = [&](int tile, int st) __attribute__((always_inline)) {
        const int key0 = tile * VA_BN;
        const uint32_t dst = lds_base + (uint32_t)st * STG;
#pragma unroll
        for (int j = 0; j < VA_BN / VA_NW; ++j) {
            const int k = wid * (VA_BN / VA_NW) + j;                         // key row of the tile (wave-uniform)
            const int row = min(key0 + k, P - 1);                            // rows behind the last key: any valid row (masked below)
            const uint32_t swz = (uint32_t)((k & 3) | (((k >> 3) & 3) << 2));
            uint32_t ch = (uint32_t)lane ^ swz;                              // LDS chunk `lane` of the row <- global chunk lane ^ swz
            if (HD < 512 && ch >= (uint32_t)(HD / 8)) ch -= 16;              // HD 384: lanes 48..63 fill the unused quarter with a valid chunk
            const uint8_t* sb = reinterpret_cast<const uint8_t*>(K) + (size_t)row * (HD * 2);
            const uint32_t vo = ch * 16u;
            const uint32_t m0v = dst + (uint32_t)k * KROW;
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(vo), "s"(sb), "s"(m0v) : "memory");
        }
#pragma unroll
        for (int j = 0; j < CB / VA_NW; ++j) {
            const int cb = wid * (CB / VA_NW) + j;
            const uint8_t* sb = reinterpret_cast<const uint8_t*>(Vt) + ((size_t)(16 * cb) * (size_t)ldv + (size_t)key0) * 2;
            const uint32_t m0v = dst + KT + (uint32_t)cb * 1024u;
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(v_voff), "s"(sb), "s"(m0v) : "memory");
        }
        (void)lane16;
    };

    // ---- fragment addresses.  K: A row r of key block kb is key 8 (r >> 2) + 4 kb + (r & 3); chunk (4 ks + g) ^ r
    const uint32_t k_row = (uint32_t)(8 * (r >> 2) + (r & 3)) * KROW;
    uint32_t k_x[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) k_x[j] = k_row + ((((uint32_t)(4 * j + g)) ^ (uint32_t)r) << 4);
    const uint32_t v_x = KT + lane16;

    const int nt = (P + VA_BN - 1) / VA_BN;
    float sv[8];
    // S^T = K Q^T of the tile staged in `sp` (two key blocks of 16), scaled + masked scores in sv
    // lane (g, r): s0[i] = key 8g + i, s1[i] = key 8g + 4 + i of the tile, query r
    auto scores = [&](int t, const uint8_t* sp) __attribute__((always_inline)) {
        f32x4 s0 = f32x4{0.f, 0.f, 0.f, 0.f}, s1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const uint8_t* a = sp + k_x[ks & 3] + (ks >> 2) * 256;
            const bf16x8 k0 = *reinterpret_cast<const bf16x8*>(a);
            const bf16x8 k1 = *reinterpret_cast<const bf16x8*>(a + 4 * KROW);
            s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, qf[ks], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, qf[ks], s1, 0, 0, 0);
        }
        const int kbase = t * VA_BN + 8 * g;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            sv[i] = (kbase + i < P) ? s0[i] * scale_log2e : -1.0e30f;
            sv[4 + i] = (kbase + 4 + i < P) ? s1[i] * scale_log2e : -1.0e30f;
        }
    };
    auto max8 = [&]() __attribute__((always_inline)) -> float {
        return fmaxf(fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3])), fmaxf(fmaxf(sv[4], sv[5]), fmaxf(sv[6], sv[7])));
    };
    auto query_max = [&](float v) __attribute__((always_inline)) -> float {   // over the 4 lanes (g) of a query
        v = fmaxf(v, __shfl_xor(v, 16));
        return fmaxf(v, __shfl_xor(v, 32));
    };

    // ---- the offset: the query's maximum over the first key tile
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    scores(0, smem);
    float m_off = query_max(max8());

    f32x4 acc[CB];
    float l_part = 0.f;          // this lane's share of the row sum (its 8 keys of every tile)
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
        l_part = 0.f;
        float mx = -1.0e30f;     // this lane's running maximum
        __syncthreads();         // stage 0 is free (the sample above / the previous pass)
        issue(0, 0);
        asm volatile("s_nop 7" ::: "memory");
        for (int t = 0; t < nt; ++t) {
            const int st = t & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces of tile t have landed
            __syncthreads();                                      // everybody's have; everybody is done with tile t - 1 (the other stage)
            if (t + 1 < nt) issue(t + 1, st ^ 1);
            const uint8_t* sp = smem + st * STG;
            scores(t, sp);
            mx = fmaxf(mx, max8());
            // numerators (bf16) and their sum
            uint32_t pk[4];
            float psum = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float a = __builtin_amdgcn_exp2f(sv[2 * i] - m_off), b = __builtin_amdgcn_exp2f(sv[2 * i + 1] - m_off);
                pk[i] = pack_bf16x2(a, b);
                psum += lo_bf16(pk[i]) + hi_bf16(pk[i]);
            }
            l_part += psum;
            const bf16x8 pf = __builtin_bit_cast(bf16x8, u32x4{pk[0], pk[1], pk[2], pk[3]});
            // O^T += V^T P^T, in place in the AGPRs ("+a": as a builtin hipcc selects the untied MFMA form here and copies all 128
            // accumulators every tile; the constraint also makes it allocate AGPRs at all).  Not volatile: hipcc schedules the LDS
            // reads around them.  Hazards hipcc cannot see (it does not count an asm MFMA's passes): the next reader of acc[cb] is this
            // statement one tile later, or the code behind the loop (explicit s_nops there).
            const uint8_t* vp = sp + v_x;
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                const bf16x8 vf = *reinterpret_cast<const bf16x8*>(vp + cb * 1024);
                asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[cb]) : "v"(vf), "v"(pf));
            }
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");     // the last tile's MFMAs have retired
        if (pass == 1) break;
        // a numerator above 2^60 somewhere in the workgroup: once more with the exact maxima (then every numerator is <= 1)
        const float qm = query_max(mx);
        if (!__syncthreads_or(qm - m_off > VA_HEADROOM)) break;
        m_off = qm;
    }

    // ---- epilogue: lane (g, r) holds O^T[16 cb + 4 g + i][query r]
    float l = l_part;
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    const float inv = 1.0f / l;
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

// O [P, C] bf16 = softmax(Q K^T * scale) V + vbias, with Q, K [P, C] bf16 row-major and V^T [C][ldv] bf16 (ldv >= ceil32(P), the
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
