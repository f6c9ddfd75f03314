// mfma_power_probe.hip — register-only streams at the chip's power cap: 32x32x16 vs 16x16x32 bf16 MFMAs, alone and with the
// softmax fillers of an attention tile (64 v_exp_f32 + 64 v_add_f32 + 32 v_cvt_pk_bf16_f32 per 2048 MFMA cycles), one wave per SIMD.
// Question: which MFMA shape sustains more FLOP/s once the clock follows the power?  Dev tool (standalone: hipcc -O3 --offload-arch=gfx950).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int SHAPE, bool FILL>   // SHAPE 32: 64 x 32x32x16 per trip; 16: 128 x 16x16x32 per trip (the same FLOPs)
__global__ __launch_bounds__(256, 1) void probe(float* out, const float* in, int trips, unsigned long long* stamps) {
    const int t = threadIdx.x;
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 8; ++j) { a[i][j] = (__bf16)in[(t * 8 + j + i * 3) & 1023]; b[i][j] = (__bf16)in[(t * 8 + j + i * 5 + 7) & 1023]; }
    float s[64];
    for (int i = 0; i < 64; ++i) s[i] = in[(t + i) & 1023] * 0.01f;
    float l = 0.f;
    unsigned int pk[32];
    unsigned long long c0, r0, c1, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0)::"memory");
    if (SHAPE == 32) {
        f32x16 acc[8];
        for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < trips; ++it) {
#pragma unroll
            for (int g = 0; g < 64; ++g) {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[g & 7]) : "v"(a[g & 3]), "v"(b[(g >> 2) & 3]));
                if (FILL) {
                    asm volatile("v_exp_f32 %0, %0\n\tv_add_f32 %1, %1, %2" : "+v"(s[g]), "+v"(l) : "v"(s[(g + 63) & 63]));
                    if (g & 1) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[g >> 1]) : "v"(s[g]), "v"(s[g - 1]));
                }
            }
        }
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
        for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) l += acc[i][r];
    } else {
        f32x4 acc[32];
        for (int i = 0; i < 32; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < trips; ++it) {
#pragma unroll
            for (int g = 0; g < 128; ++g) {
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[g & 31]) : "v"(a[g & 3]), "v"(b[(g >> 2) & 3]));
                if (FILL) {
                    if ((g & 1) == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(s[g >> 1]));
                    else {
                        asm volatile("v_add_f32 %0, %0, %1" : "+v"(l) : "v"(s[g >> 1]));
                        if (g & 2) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[g >> 2]) : "v"(s[g >> 1]), "v"(s[(g >> 1) - 1]));
                    }
                }
            }
        }
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
        for (int i = 0; i < 32; ++i) for (int r = 0; r < 4; ++r) l += acc[i][r];
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1)::"memory");
    for (int i = 0; i < 64; ++i) l += s[i];
    if (FILL) for (int i = 0; i < 32; ++i) l += (float)pk[i];
    out[blockIdx.x * 256 + t] = l;
    if ((t & 63) == 0) { stamps[(blockIdx.x * 4 + (t >> 6)) * 2] = c1 - c0; stamps[(blockIdx.x * 4 + (t >> 6)) * 2 + 1] = r1 - r0; }
}

template <int SHAPE, bool FILL>
void run(const char* name, float* out, float* in, unsigned long long* st, int trips, int blocks) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((probe<SHAPE, FILL>), dim3(blocks), dim3(256), 0, 0, out, in, trips, st);
    hipEventRecord(e0);
    const int reps = 10;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((probe<SHAPE, FILL>), dim3(blocks), dim3(256), 0, 0, out, in, trips, st);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * 8);
    hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0;
    for (size_t i = 0; i < h.size() / 2; ++i) { cyc += h[2 * i]; rt += h[2 * i + 1]; }
    const double flops = (double)blocks * 4 * trips * 64.0 * 2 * 32 * 32 * 16 * 64 / 64;   // per wave and trip: 64 x (2*32*32*16) FLOPs
    printf("%-34s %7.3f ms  %7.1f TF/s  %6.0f cycles per 2048-cycle trip  in-loop clock %5.0f MHz\n", name, ms / reps, flops * reps / ms / 1e9,
           cyc / (h.size() / 2) / trips, cyc / rt * 100.0);
}

typedef int i32x8 __attribute__((ext_vector_type(8)));
// e4m3 operands through the block-scaled 16x16x128 form (unit scales): 4x the FLOPs of a 16x16x32 bf16 MFMA in twice its time
__global__ __launch_bounds__(256, 1) void probe_fp8(float* out, const float* in, int trips, unsigned long long* stamps) {
    const int t = threadIdx.x;
    i32x8 a[4], b[4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 8; ++j) { a[i][j] = (int)(in[(t * 8 + j + i * 3) & 1023] * 100000.f) * 2654435 + t; b[i][j] = (int)(in[(t * 8 + j + i * 5 + 7) & 1023] * 100000.f) * 40503 + 7 * t; }
    // keep the bytes finite e4m3 values: clear the exponent's top bit of every byte (|x| < 2)
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 8; ++j) { a[i][j] &= 0xBFBFBFBF; b[i][j] &= 0xBFBFBFBF; }
    f32x4 acc[32];
    for (int i = 0; i < 32; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    unsigned long long c0, r0, c1, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0)::"memory");
    for (int it = 0; it < trips; ++it) {
#pragma unroll
        for (int g = 0; g < 64; ++g)
            acc[g & 31] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[g & 3], b[(g >> 2) & 3], acc[g & 31], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1)::"memory");
    float l = 0.f;
    for (int i = 0; i < 32; ++i) for (int r = 0; r < 4; ++r) l += acc[i][r];
    out[blockIdx.x * 256 + t] = l;
    if ((t & 63) == 0) { stamps[(blockIdx.x * 4 + (t >> 6)) * 2] = c1 - c0; stamps[(blockIdx.x * 4 + (t >> 6)) * 2 + 1] = r1 - r0; }
}

void run_fp8(float* out, float* in, unsigned long long* st, int trips, int blocks) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(probe_fp8, dim3(blocks), dim3(256), 0, 0, out, in, trips, st);
    hipEventRecord(e0);
    const int reps = 10;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(probe_fp8, dim3(blocks), dim3(256), 0, 0, out, in, trips, st);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * 8);
    hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0;
    for (size_t i = 0; i < h.size() / 2; ++i) { cyc += h[2 * i]; rt += h[2 * i + 1]; }
    const double flops = (double)blocks * 4 * trips * 64.0 * 2 * 16 * 16 * 128;
    printf("%-34s %7.3f ms  %7.1f TF/s  %6.0f cycles per 64-MFMA trip (2048 = issue alone)  in-loop clock %5.0f MHz\n", "e4m3 16x16x128 (scaled form) alone", ms / reps,
           flops * reps / ms / 1e9, cyc / (h.size() / 2) / trips, cyc / rt * 100.0);
}

int main() {
    const int blocks = 256, trips = 2000;
    float *in, *out; unsigned long long* st;
    hipMalloc(&in, 4096); hipMalloc(&out, blocks * 256 * 4); hipMalloc(&st, blocks * 8 * 8);
    std::vector<float> h(1024);
    srand(1);
    for (auto& v : h) v = (float)(rand() % 2001 - 1000) / 500.f;
    hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
    for (int round = 0; round < 2; ++round) {
        run<32, false>("32x32x16 alone", out, in, st, trips, blocks);
        run<16, false>("16x16x32 alone", out, in, st, trips, blocks);
        run<32, true>("32x32x16 + softmax fillers", out, in, st, trips, blocks);
        run<16, true>("16x16x32 + softmax fillers", out, in, st, trips, blocks);
        run_fp8(out, in, st, trips, blocks);
    }
    return 0;
}
