// tools/store_pattern.hip — what a CU's vector-memory path gives a 16-byte-per-lane store / load, by lane -> address mapping
// (development probe, not part of the library).
//   hipcc -O3 --offload-arch=gfx950 tools/store_pattern.hip -o /tmp/store_pattern && /tmp/store_pattern
// The GEMM epilogues write a 16 x 16 MFMA block row as lane (frow = lane & 15, fq = lane >> 4) -> row frow, 16 bytes at
// ((fq & 1) 32 + (fq >> 1) 16): the four lanes of a row's 64-byte run are 16 lanes apart.  Is that pattern bound by the number of
// requests (every lane its own 16-byte request) rather than by bytes?  Variants, each one wave per SIMD on every CU (as the GEMM runs),
// every wave writing / reading its own region of `rows` x pitch bytes again and again:
//   mfma   : the epilogue's pattern as it is (16 rows x 64 B per instruction)
//   quad   : lane -> row lane >> 2, 16-byte chunk lane & 3 (the same 16 rows x 64 B, the four lanes of a run adjacent)
//   line   : lane -> row lane >> 3, chunk lane & 7 (8 rows x 128 B: whole cache lines)
//   row256 : lane -> row lane >> 4, chunk lane & 15 (4 rows x 256 B: the N = 128 convolution's whole output rows)
// A wave owns 256 bytes (a wave tile's 128 bf16 columns) x 512 rows of a matrix whose row pitch is the parameter (256: the N = 128
// convolution; 512; 6144: d = 3072; 55296: linear1's C).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int PAT, bool LOAD>
__global__ __launch_bounds__(256, 1) void pat_kernel(uint8_t* buf, size_t wave_bytes, int pitch, int trips, uint64_t* clk, uint32_t* sink) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    // wave w owns 256 bytes x nrows of a matrix with row pitch `pitch`: column slot w % (pitch / 256) of row band w / (pitch / 256)
    const int w = blockIdx.x * 4 + wid, slots = pitch / 256, nrows = (int)(wave_bytes / 256);
    uint8_t* base = buf + (size_t)(w / slots) * nrows * pitch + (size_t)(w % slots) * 256;
    int row, off, rows_per;
    if (PAT == 0) { row = lane & 15; const int fq = lane >> 4; off = (fq & 1) * 32 + (fq >> 1) * 16; rows_per = 16; }
    else if (PAT == 1) { row = lane >> 2; off = (lane & 3) * 16; rows_per = 16; }
    else if (PAT == 2) { row = lane >> 3; off = (lane & 7) * 16; rows_per = 8; }
    else { row = lane >> 4; off = (lane & 15) * 16; rows_per = 4; }
    const int run = PAT == 0 || PAT == 1 ? 64 : (PAT == 2 ? 128 : 256);   // bytes of a row one instruction covers
    const int per_row = 256 / run;                                         // instructions per block of rows_per rows
    const int blocks = nrows / rows_per;
    u32x4 v = {(uint32_t)lane, (uint32_t)wid, blockIdx.x, 0x3f803f80u};
    uint32_t acc = 0;
    uint64_t c0, c1, r0, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0)::"memory");
    for (int t = 0; t < trips; ++t) {
        asm volatile("" ::: "memory");
#pragma unroll 4
        for (int b = 0; b < blocks; ++b) {
            uint8_t* p = base + (size_t)(b * rows_per + row) * pitch + off;
#pragma unroll 4
            for (int s = 0; s < per_row; ++s) {
                if (LOAD) { const u32x4 x = *reinterpret_cast<const u32x4*>(p + s * run); acc += x[0] ^ x[3]; }   // hipcc's own load: it counts vmcnt
                else asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p + s * run), "v"(v) : "memory");
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1)::"memory");
    if (lane == 0 && wid == 0) { clk[blockIdx.x * 2] = c1 - c0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
    if (LOAD && acc == 0x12345678u) sink[0] = acc;
}

template <int PAT, bool LOAD>
static void run(const char* name, uint8_t* buf, size_t wave_bytes, int pitch, uint64_t* clk, uint32_t* sink, int ncu) {
    const int trips = 24;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((pat_kernel<PAT, LOAD>), dim3(ncu), dim3(256), 0, 0, buf, wave_bytes, pitch, 2, clk, sink);
    hipEventRecord(e0);
    hipLaunchKernelGGL((pat_kernel<PAT, LOAD>), dim3(ncu), dim3(256), 0, 0, buf, wave_bytes, pitch, trips, clk, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    uint64_t* h = (uint64_t*)malloc(ncu * 16);
    hipMemcpy(h, clk, ncu * 16, hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0;
    for (int i = 0; i < ncu; ++i) { cyc += (double)h[2 * i]; rt += (double)h[2 * i + 1]; }
    const double bytes_cu = (double)wave_bytes * 4 * trips;
    printf("%-5s %-7s pitch %6d: %7.2f B/clk/CU  %6.2f TB/s chip-wide  (clock %.0f MHz, %.3f ms)\n", LOAD ? "load" : "store", name, pitch,
           bytes_cu / (cyc / ncu), bytes_cu * ncu / (ms * 1e-3) / 1e12, cyc / rt * 100.0, ms);
    free(h);
}

int main() {
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    const int ncu = pr.multiProcessorCount;
    const size_t wave_bytes = 128 * 1024;   // per wave: 512 KB per CU, 128 MB chip-wide (beyond the L2s; a GEMM's C is written once)
    uint8_t* buf; uint64_t* clk; uint32_t* sink;
    const size_t total = wave_bytes * 4 * ncu + (size_t)64 * 1024 * 1024;   // + slack for the last partial row band
    hipMalloc((void**)&buf, total);
    hipMemset(buf, 1, total);
    hipMalloc((void**)&clk, ncu * 16);
    hipMalloc((void**)&sink, 4);
    printf("%d CUs, one wave per SIMD, each wave its own %zu KB\n", ncu, wave_bytes / 1024);
    // how much of the figure is the chip's, how much a CU's own: the epilogue's pattern and the best one on fewer CUs
    for (int n : {8, 64}) {
        printf("-- %d workgroups\n", n);
        run<0, false>("mfma", buf, wave_bytes, 256, clk, sink, n);
        run<1, false>("quad", buf, wave_bytes, 256, clk, sink, n);
        run<2, false>("line", buf, wave_bytes, 256, clk, sink, n);
        run<3, false>("row256", buf, wave_bytes, 256, clk, sink, n);
        run<2, false>("line", buf, wave_bytes, 6144, clk, sink, n);
        run<3, false>("row256", buf, wave_bytes, 6144, clk, sink, n);
        run<0, true>("mfma", buf, wave_bytes, 256, clk, sink, n);
        run<1, true>("quad", buf, wave_bytes, 256, clk, sink, n);
        run<2, true>("line", buf, wave_bytes, 256, clk, sink, n);
        run<3, true>("row256", buf, wave_bytes, 256, clk, sink, n);
        run<2, true>("line", buf, wave_bytes, 6144, clk, sink, n);
        run<3, true>("row256", buf, wave_bytes, 6144, clk, sink, n);
    }
    printf("-- all CUs\n");
    for (int pitch : {256, 512, 6144, 55296}) {
        run<0, false>("mfma", buf, wave_bytes, pitch, clk, sink, ncu);
        run<1, false>("quad", buf, wave_bytes, pitch, clk, sink, ncu);
        run<2, false>("line", buf, wave_bytes, pitch, clk, sink, ncu);
        run<3, false>("row256", buf, wave_bytes, pitch, clk, sink, ncu);
        run<0, true>("mfma", buf, wave_bytes, pitch, clk, sink, ncu);
        run<1, true>("quad", buf, wave_bytes, pitch, clk, sink, ncu);
        run<2, true>("line", buf, wave_bytes, pitch, clk, sink, ncu);
        run<3, true>("row256", buf, wave_bytes, pitch, clk, sink, ncu);
    }
    return 0;
}
