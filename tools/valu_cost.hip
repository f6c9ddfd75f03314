// tools/valu_cost.hip — what an instruction costs a lone wave (one wave per SIMD, as the GEMM epilogues run): cycles per instruction of
// straight-line blocks of one kind, executed once (cold) and again (warm).  Development probe, not part of the library.
//   hipcc -O3 --offload-arch=gfx950 tools/valu_cost.hip -o /tmp/valu_cost && /tmp/valu_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

typedef float f32x2 __attribute__((ext_vector_type(2)));

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
#define REP512(x) REP8(REP64(x))

// KIND 0: v_add_f32 (4 independent chains)  1: v_pk_add_f32  2: v_pk_fma_f32  3: v_cvt_pk_bf16_f32  4: v_accvgpr_read_b32
//      5: v_lshlrev_b32 + v_and_b32 (the bf16 unpack pair)  6: s_nop 0  7: dependent v_add_f32 chain  8: v_mov_b64
template <int KIND>
__global__ __launch_bounds__(256, 1) void cost_kernel(float* out, uint64_t* clk, int passes) {
    float a = threadIdx.x * 0.5f, b = 1.25f, c = 2.5f, d = 3.75f;
    f32x2 p = {a, b}, q = {c, d}, r = {b, c}, s = {d, a};
    uint32_t u = threadIdx.x, w = threadIdx.x * 7u;
    uint64_t t0, t1, t2;
    for (int pass = 0; pass < passes; ++pass) {
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        if (KIND == 0) { REP512(asm volatile("v_add_f32 %0, %0, %4\n\tv_add_f32 %1, %1, %4\n\tv_add_f32 %2, %2, %4\n\tv_add_f32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(1.0f));) }
        if (KIND == 1) { REP512(asm volatile("v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %4\n\tv_pk_add_f32 %2, %2, %4\n\tv_pk_add_f32 %3, %3, %4" : "+v"(p), "+v"(q), "+v"(r), "+v"(s) : "v"(p));) }
        if (KIND == 2) { REP512(asm volatile("v_pk_fma_f32 %0, %4, %4, %0\n\tv_pk_fma_f32 %1, %4, %4, %1\n\tv_pk_fma_f32 %2, %4, %4, %2\n\tv_pk_fma_f32 %3, %4, %4, %3" : "+v"(p), "+v"(q), "+v"(r), "+v"(s) : "v"(p));) }
        if (KIND == 3) { REP512(asm volatile("v_cvt_pk_bf16_f32 %0, %4, %5\n\tv_cvt_pk_bf16_f32 %1, %4, %5\n\tv_cvt_pk_bf16_f32 %2, %4, %5\n\tv_cvt_pk_bf16_f32 %3, %4, %5" : "+v"(u), "+v"(w), "+v"(a), "+v"(b) : "v"(c), "v"(d));) }
        if (KIND == 4) { REP512(asm volatile("v_accvgpr_read_b32 %0, a0\n\tv_accvgpr_read_b32 %1, a1\n\tv_accvgpr_read_b32 %2, a2\n\tv_accvgpr_read_b32 %3, a3" : "=v"(a), "=v"(b), "=v"(c), "=v"(d)::"a0", "a1", "a2", "a3");) }
        if (KIND == 5) { REP512(asm volatile("v_lshlrev_b32 %0, 16, %4\n\tv_and_b32 %1, 0xffff0000, %4\n\tv_lshlrev_b32 %2, 16, %5\n\tv_and_b32 %3, 0xffff0000, %5" : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"(u), "v"(w));) }
        if (KIND == 6) { REP512(asm volatile("s_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0");) }
        if (KIND == 7) { REP512(asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1" : "+v"(a) : "v"(1.0f));) }
        if (KIND == 8) { REP512(asm volatile("v_mov_b64 %0, %4\n\tv_mov_b64 %1, %4\n\tv_mov_b64 %2, %4\n\tv_mov_b64 %3, %4" : "=v"(p), "=v"(q), "=v"(r), "=v"(s) : "v"(p));) }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        if (threadIdx.x == 0) clk[blockIdx.x * 8 + (pass < 7 ? pass : 7)] = t1 - t0;
    }
    (void)t2;
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + p[0] + q[1] + r[0] + s[1] + (float)u + (float)w;
}

template <int KIND>
static void run(const char* name, float* out, uint64_t* clk, int nwg) {
    hipMemset(clk, 0, nwg * 64);
    hipLaunchKernelGGL((cost_kernel<KIND>), dim3(nwg), dim3(256), 0, 0, out, clk, 3);
    hipDeviceSynchronize();
    uint64_t* h = (uint64_t*)malloc(nwg * 64);
    hipMemcpy(h, clk, nwg * 64, hipMemcpyDeviceToHost);
    double c0 = 0, c2 = 0;
    for (int i = 0; i < nwg; ++i) { c0 += (double)h[i * 8]; c2 += (double)h[i * 8 + 2]; }
    printf("%-28s 2048 instructions: first pass %.1f cycles per instruction, third pass %.1f\n", name, c0 / nwg / 2048.0, c2 / nwg / 2048.0);
    free(h);
}

int main() {
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    const int nwg = pr.multiProcessorCount;
    float* out; uint64_t* clk;
    hipMalloc((void**)&out, nwg * 1024);
    hipMalloc((void**)&clk, nwg * 64);
    printf("%d workgroups of 4 waves (one per SIMD)\n", nwg);
    run<0>("v_add_f32 (4 chains)", out, clk, nwg);
    run<7>("v_add_f32 (dependent)", out, clk, nwg);
    run<1>("v_pk_add_f32", out, clk, nwg);
    run<2>("v_pk_fma_f32", out, clk, nwg);
    run<3>("v_cvt_pk_bf16_f32", out, clk, nwg);
    run<4>("v_accvgpr_read_b32", out, clk, nwg);
    run<5>("v_lshlrev / v_and", out, clk, nwg);
    run<8>("v_mov_b64", out, clk, nwg);
    run<6>("s_nop 0", out, clk, nwg);
    return 0;
}
