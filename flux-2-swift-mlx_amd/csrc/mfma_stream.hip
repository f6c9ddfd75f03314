// mfma_stream.hip — what the matrix pipe of THIS card sustains: a register-only stream of the MFMA the hot kernels use (bf16
// v_mfma_f32_16x16x32_bf16, or e4m3 v_mfma_scale_f32_16x16x128_f8f6f4 with unit scales) on pseudo-random operands, one wave per SIMD
// on every CU, no LDS and no memory traffic in the loop.  The GEMM and attention kernels run at the board's power cap, where the
// clock — not the issue rate — sets the ceiling; this number is that ceiling on the box at hand (bench.py reports it next to
// `roofline.achieved`; DESIGN.md §3 "the chip runs these kernels at its power cap").  Diagnostic entry point, not on the image path.
#include "kernels.h"
#include <hip/hip_runtime.h>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 ms_bf16x8;
typedef __attribute__((ext_vector_type(4))) float ms_f32x4;
typedef int ms_i32x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ uint32_t ms_hash(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <bool FP8>
__global__ __launch_bounds__(256, 1) void mfma_stream_kernel(float* out, int trips) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    ms_f32x4 acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = ms_f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (FP8) {
        ms_i32x8 a[4], b[4];   // bytes with the exponent's top bit clear: finite e4m3 values, |x| < 2
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                a[i][j] = (int)(ms_hash(t * 64 + i * 8 + j) & 0xBFBFBFBFu);
                b[i][j] = (int)(ms_hash(t * 64 + 32 + i * 8 + j) & 0xBFBFBFBFu);
            }
        for (int it = 0; it < trips; ++it) {
#pragma unroll
            for (int g = 0; g < 64; ++g)
                acc[g & 31] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[g & 3], b[(g >> 2) & 3], acc[g & 31], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
        }
    } else {
        ms_bf16x8 a[4], b[4];   // bf16 values in [-1, 1) with random mantissas
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                a[i][j] = (__bf16)((float)(int)(ms_hash(t * 64 + i * 8 + j) >> 8) * (1.0f / 8388608.0f) - 1.0f);
                b[i][j] = (__bf16)((float)(int)(ms_hash(t * 64 + 32 + i * 8 + j) >> 8) * (1.0f / 8388608.0f) - 1.0f);
            }
        for (int it = 0; it < trips; ++it) {
#pragma unroll
            for (int g = 0; g < 128; ++g)
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[g & 31]) : "v"(a[g & 3]), "v"(b[(g >> 2) & 3]));
        }
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
        for (int i = 0; i < 32; ++i) asm volatile("" : "+a"(acc[i]));
    }
    float l = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) l += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[t] = l;
}

}  // namespace

// Runs the stream for about `seconds` (the power controller needs a few hundred ms to settle) and times its second half with events.
// out: workgroups x 256 floats of device scratch.  Returns 0 ok / 2 HIP error; *tflops = sustained rate of the timed half.
int launch_mfma_stream(hipStream_t stream, float* out, int workgroups, bool fp8, double seconds, double* tflops) {
    const int trips = 2000;                                            // 2000 x 2048 MFMA cycles: ~2 ms per launch
    const double flops_per_launch = (double)workgroups * 4 * trips * (fp8 ? 64.0 * 2 * 16 * 16 * 128 : 128.0 * 2 * 16 * 16 * 32);
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess) return 2;
    if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return 2; }
    auto go = [&]() {
        if (fp8) hipLaunchKernelGGL(mfma_stream_kernel<true>, dim3(workgroups), dim3(256), 0, stream, out, trips);
        else hipLaunchKernelGGL(mfma_stream_kernel<false>, dim3(workgroups), dim3(256), 0, stream, out, trips);
    };
    const int n = (int)(seconds * 0.5 / 2.2e-3) + 1;
    for (int i = 0; i < n; ++i) go();
    (void)hipEventRecord(e0, stream);
    for (int i = 0; i < n; ++i) go();
    (void)hipEventRecord(e1, stream);
    float ms = 0.f;
    const bool ok = hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms > 0.f;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (!ok || hipGetLastError() != hipSuccess) return 2;
    *tflops = flops_per_launch * n / (ms * 1e-3) / 1e12;
    return 0;
}
