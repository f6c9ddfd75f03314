// dev.h — device-side helpers shared by the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;  // staging registers (HIP uint4 arrays are not SROA-promoted)

__device__ __forceinline__ float bf16_bits_to_f32(uint16_t b) {
    return __builtin_bit_cast(float, (uint32_t)b << 16);
}
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) {
    __bf16 h = (__bf16)f;  // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN-preserving
    return __builtin_bit_cast(uint16_t, h);
}
// the three-instruction form (convert, convert, shift + or): kept for the e4m3 GEMM's epilogue, whose register allocation has no slack
// (with the packed form below hipcc spills one VGPR there)
__device__ __forceinline__ uint32_t pack_bf16x2_3op(float lo, float hi) {
    return (uint32_t)f32_to_bf16_bits(lo) | ((uint32_t)f32_to_bf16_bits(hi) << 16);
}
// one v_cvt_pk_bf16_f32 (the scalar form above, shifted and or-ed, is three instructions per pair: with one wave per SIMD every VALU
// instruction of an epilogue is four cycles of its own)
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    typedef __bf16 pk_bf16x2 __attribute__((ext_vector_type(2)));
    typedef float pk_f32x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(pk_f32x2{lo, hi}, pk_bf16x2));
}
__device__ __forceinline__ float lo_bf16(uint32_t u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float hi_bf16(uint32_t u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

// Block-scaled e4m3 activations (GemmProblem::a_bscale): the exponent e of a block with absolute maximum amax is the largest integer
// with amax 2^e <= 448, clamped to [-60, 60] (amax = 0: 60); the block stores x 2^e as e4m3 and the e8m0 byte 127 - e.  Exact integer
// arithmetic on the float's fields (448 = 0.875 x 2^9), so that every producer and the oracle agree on every block.
__device__ __forceinline__ int fp8_block_exp(float amax) {
    if (!(amax > 0.f)) return 60;
    int x;
    const float m = frexpf(amax, &x);          // amax = m 2^x, m in [0.5, 1)
    const int e = (m <= 0.875f ? 9 : 8) - x;
    return e < -60 ? -60 : (e > 60 ? 60 : e);
}

// Maximum of a non-negative value over the 16 lanes of a DPP row; valid in lane 15 of the row (row_shr 1, 2, 4, 8 with zero fill)
__device__ __forceinline__ float row16_max_nonneg(float v) {
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true)));
    return v;
}

// OCP e4m3 pack of 4 floats the caller knows to lie inside +-448 (block-scaled values: the block exponent guarantees it)
__device__ __forceinline__ uint32_t pack_fp8x4_inrange(float a, float b, float c, float d) {
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return (uint32_t)w;
}

// OCP e4m3 pack of 4 floats (already divided by the scale); clamped to +-448 (the cast has no saturation)
__device__ __forceinline__ uint32_t pack_fp8x4(float a, float b, float c, float d) {
    a = fminf(fmaxf(a, -448.f), 448.f); b = fminf(fmaxf(b, -448.f), 448.f);
    c = fminf(fmaxf(c, -448.f), 448.f); d = fminf(fmaxf(d, -448.f), 448.f);
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return (uint32_t)w;
}

// x * sigmoid(x) with v_exp_f32 + v_rcp_f32 (1 ulp each; every use rounds the result to bf16 or feeds a
// bf16 GEMM).  The IEEE division `x / (1 + expf(-x))` costs ~12 VALU instructions per element
// (v_div_scale x2, v_rcp, 4 fma, v_div_fmas, v_div_fixup): in the SwiGLU epilogue of a 256x256 GEMM tile that
// was ~800 instructions per lane.
__device__ __forceinline__ float silu_f(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
