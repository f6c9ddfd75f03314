// tools/mfma_shapes.hip — what sets the rate of an MFMA-bound loop on THIS card (development probe, not part of the library).
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_shapes.hip -o /tmp/mfma_shapes && /tmp/mfma_shapes
// One wave per SIMD on every CU (as gemm256s / attention64b run), bf16 MFMA streams in the two shapes 16x16x32 (what the kernels use)
// and 32x32x16 (half the VGPR operand reads per FLOP), each
//   * alone (registers only),
//   * with the LDS fragment reads of a 128x128 wave tile next to it (one ds_read_b128 per 4 MFMAs of 16x16x32 = 16 B/clk/SIMD), and with
//     half / twice that,
//   * with the global->LDS delivery of a 256x256 tile's K-tile pair (64 KB per CU per 2048 MFMA clocks = 32 B/clk/CU, LDS-DMA out of L2),
//     alone, on top of the LDS reads (= the traffic mix of gemm256s' bf16 K loop), and doubled.
// Every variant runs ~0.5 s after a 0.3 s warm-up.  A line gives the sustained rate, the shader clock measured in the kernel (s_memtime
// over s_memrealtime: the power controller's answer to that mix) and the loop's clocks per 2048 MFMA clocks (above 2048 = issue stalls).
// Nothing waits inside the loops; the asynchronous LDS reads land in registers that are kept live ("+v") so that hipcc never re-uses them.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) int i32x4;

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// SHAPE 0: 16x16x32, 128 MFMAs per trip (2048 clk); SHAPE 1: 32x32x16, 64 MFMAs per trip (2048 clk).
// LDS_READS: ds_read_b128 per trip (GEMM ratio = 32: 16 B/lane x 64 lanes x 32 = 32 KB per 2048 clk = 16 B/clk/SIMD)
// DMA: global_load_lds_dwordx4 per wave per trip (GEMM ratio = 16: a 64 KB K-tile pair per CU per 2048 clk = 32 B/clk/CU), from a 2 MB
// region every workgroup walks (L2 hits, as the shared A / W panels of a GEMM are)
// PAT 1 (with DMA 32): the addresses of a 256x256 tile grid instead — 1 MB panels of 256 rows x 4096 B, a piece = 8 rows x 128 B, two 128-B
// K-tiles per trip, 16 trips per tile; 4 A panels x 8 W panels per XCD (blockIdx & 7), the A panels re-read by every tile, the W panels
// walking through 1000 MB (streamed from HBM as a weight matrix is): what gemm256s' e4m3 loop asks of the memory system, 64 B/clk/CU
template <int SHAPE, int LDS_READS, int DMA = 0, int PAT = 0, int SYNC = 0>
__global__ __launch_bounds__(256, 1) void stream_kernel(float* out, int trips, const uint8_t* src, uint64_t* clk) {
    extern __shared__ uint32_t lds[];
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    for (int i = threadIdx.x; i < 32768 / 4; i += 256) lds[i] = (hash32(t + i * 977u) & 0xBF7FBF7Fu);   // finite bf16 pairs
    __syncthreads();
    bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            a[i][j] = (__bf16)((float)(int)(hash32(t * 64 + i * 8 + j) >> 8) * (1.0f / 8388608.0f) - 1.0f);
            b[i][j] = (__bf16)((float)(int)(hash32(t * 64 + 32 + i * 8 + j) >> 8) * (1.0f / 8388608.0f) - 1.0f);
        }
    i32x4 sink[4] = {};   // "+v" below: the reads land asynchronously, so their registers must stay live (never re-used) through the loop
    const uint32_t lane_off = (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 8192;   // each wave reads its own 8 KB, conflict-free
    float l = 0.f;
    uint64_t c0, r0, c1, r1;                                            // shader clocks and the 100 MHz reference around the loop
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0)::"memory");
    const uint32_t dma_lane = (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 16384;
    const uint32_t dma_m0 = __builtin_amdgcn_readfirstlane((threadIdx.x >> 6) * 8192);
    constexpr size_t MB = 1u << 20;
    const int xcd = blockIdx.x & 7, local = (blockIdx.x >> 3) & 31;
    const int tm = (xcd & 3) * 4 + (local >> 3), tn = (xcd >> 2) * 8 + (local & 7);
    uint32_t rowoff[8];                                                  // piece j of wave w: rows (4j + w) 8 .. + 8, 16-B chunk lane & 7
#pragma unroll
    for (int j = 0; j < 8; ++j) rowoff[j] = (uint32_t)((((4 * j + (threadIdx.x >> 6)) * 8 + ((threadIdx.x & 63) >> 3)) * 4096) + (threadIdx.x & 7) * 16);
    uint32_t packoff[8];                                                 // the same piece in a K-tile-major panel: 1 KB contiguous
#pragma unroll
    for (int j = 0; j < 8; ++j) packoff[j] = (uint32_t)((4 * j + (threadIdx.x >> 6)) * 1024 + (threadIdx.x & 63) * 16);
    if constexpr (SHAPE == 0) {
        f32x4 acc[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < trips; ++it) {
            // PAT 1: row-major panels (pitch 4096 B).  PAT 2: W packed K-tile-major (a K-tile's 256 rows x 128 B contiguous: a piece is 1 KB
            // contiguous), A row-major.  PAT 3: both packed.  PAT 4: PAT 1 with the W panels NOT walking (everything stays in L2 / MALL).
            // PAT 5 / 6: PAT 1 with the W walk wrapping in 128 / 64 MB (W stays in the 256 MB memory-side cache, not in the 4 MB L2s).
            constexpr bool A_PACKED = PAT == 3, W_PACKED = PAT == 2 || PAT == 3;
            const uint8_t* sA = src + (size_t)tm * MB + (size_t)(it & 15) * (A_PACKED ? 65536 : 256);
            const uint8_t* sW = src + 16 * MB + (size_t)(((PAT == 4 ? 0 : (it >> 4)) * 16 + tn) % (PAT == 5 ? 128 : (PAT == 6 ? 64 : 1000))) * MB + (size_t)(it & 15) * (W_PACKED ? 65536 : 256);
#pragma unroll
            for (int g = 0; g < 128; ++g) {
                if (LDS_READS > 0 && (g % (128 / (LDS_READS > 0 ? LDS_READS : 1))) == 0)
                    asm volatile("ds_read_b128 %0, %1" : "+v"(sink[g & 3]) : "v"(lane_off + ((g * 1024) & 7168)));
                if (DMA > 0 && (g % (128 / (DMA > 0 ? DMA : 1))) == 0) {
                    const int piece = g / (128 / (DMA > 0 ? DMA : 1));
                    asm volatile("s_add_u32 m0, %0, %1" ::"s"(dma_m0), "s"((piece & 7) * 1024) : "memory", "scc");
                    if constexpr (PAT == 0)
                        asm volatile("global_load_lds_dwordx4 %0, %1" ::"v"(dma_lane + (uint32_t)((it & 31) * 65536 + piece * 1024)), "s"(src) : "memory");
                    else if ((piece & 15) < 8)
                        asm volatile("global_load_lds_dwordx4 %0, %1" ::"v"(A_PACKED ? packoff[piece & 7] + (uint32_t)((piece >> 4) * 32768) : rowoff[piece & 7] + (uint32_t)((piece >> 4) * 128)), "s"(sA) : "memory");
                    else
                        asm volatile("global_load_lds_dwordx4 %0, %1" ::"v"(W_PACKED ? packoff[piece & 7] + (uint32_t)((piece >> 4) * 32768) : rowoff[piece & 7] + (uint32_t)((piece >> 4) * 128)), "s"(sW) : "memory");
                }
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[g & 31]) : "v"(a[g & 3]), "v"(b[(g >> 2) & 3]));
                // SYNC n (with DMA 32): a two-stage ring's hand-over every 64 MFMAs (= one e4m3 K-tile): at most n pieces may still be in
                // flight (16 = the tile issued during the last 64 MFMAs, as gemm256s' B2 allows), then the workgroup's barrier
                if constexpr (SYNC > 0) if ((g & 63) == 63) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(SYNC) : "memory");
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // nothing waits inside the loop: the counters back-pressure by themselves
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
        for (int i = 0; i < 32; ++i) { asm volatile("" : "+a"(acc[i])); l += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3]; }
    } else {
        f32x16 acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        for (int it = 0; it < trips; ++it) {
#pragma unroll
            for (int g = 0; g < 64; ++g) {
                if (LDS_READS > 0 && (g % (64 / (LDS_READS > 0 ? (LDS_READS > 64 ? 64 : LDS_READS) : 1))) == 0) {
                    asm volatile("ds_read_b128 %0, %1" : "+v"(sink[g & 3]) : "v"(lane_off + ((g * 1024) & 7168)));
                    if (LDS_READS > 64)
                        asm volatile("ds_read_b128 %0, %1" : "+v"(sink[(g + 1) & 3]) : "v"(lane_off + ((g * 1024 + 512) & 7168)));
                }
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[g & 7]) : "v"(a[g & 3]), "v"(b[(g >> 2) & 3]));
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            asm volatile("" : "+a"(acc[i]));
#pragma unroll
            for (int j = 0; j < 16; ++j) l += acc[i][j];
        }
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1)::"memory");
    out[t] = l + (float)(sink[0][0] ^ sink[1][1] ^ sink[2][2] ^ sink[3][3]) * 1e-30f;
    if (clk && threadIdx.x == 0) { clk[blockIdx.x * 2] = c1 - c0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

__global__ void fill_kernel(uint32_t* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = hash32((uint32_t)i * 2654435761u + (uint32_t)(i >> 32)) & 0xBF7FBF7Fu;
}

// every byte of [p, p + 16 n16) read once
__global__ __launch_bounds__(256) void touch_kernel(const uint4* p, size_t n16) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const uint4 v = p[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    asm volatile("" ::"v"(acc));
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

static uint64_t* g_clk = nullptr;

template <int SHAPE, int LDS_READS, int DMA = 0, int PAT = 0, int SYNC = 0>
static void run(const char* name, float* out, int n_cu, hipStream_t s, const uint8_t* src = nullptr) {
    const int trips = 2000;
    const double flops = (double)n_cu * 4 * trips * 2048.0 * 1024.0;     // 2048 MFMA clocks per trip x 1024 flop/clk/SIMD
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto go = [&]() { hipLaunchKernelGGL((stream_kernel<SHAPE, LDS_READS, DMA, PAT, SYNC>), dim3(n_cu), dim3(256), 32768, s, out, trips, src, g_clk); };
    for (int i = 0; i < 140; ++i) go();
    CK(hipEventRecord(e0, s));
    const int n = 230;
    for (int i = 0; i < n; ++i) go();
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    const double tf = flops * n / (ms * 1e-3) / 1e12;
    static uint64_t h[4096];                                            // the last launch's counters: mean clock, and loop clocks per trip
    CK(hipMemcpy(h, g_clk, (size_t)n_cu * 16, hipMemcpyDeviceToHost));
    double cyc = 0, ref = 0;
    for (int i = 0; i < n_cu; ++i) { cyc += (double)h[2 * i]; ref += (double)h[2 * i + 1]; }
    printf("%-44s %8.1f TFLOP/s   clock %.3f GHz   %6.0f clk per 2048 MFMA clk   (%.3f ms per launch)\n", name, tf, cyc / ref * 0.1,
           cyc / n_cu / trips, ms / n);
    fflush(stdout);
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int n_cu = p.multiProcessorCount;
    printf("%s, %d CUs, clock rate %.2f GHz\n", p.gcnArchName, n_cu, p.clockRate * 1e-6);
    hipStream_t s;
    CK(hipStreamCreate(&s));
    float* out;
    CK(hipMalloc(&out, (size_t)n_cu * 256 * 4));
    if (n_cu != 256) { fprintf(stderr, "the panel map assumes 256 CUs\n"); return 2; }
    if (n_cu > 2048) { fprintf(stderr, "more CUs than the counter table holds\n"); return 2; }
    CK(hipMalloc(&g_clk, (size_t)n_cu * 16));
    uint8_t* src;                                                        // 16 A panels + 1000 W panels of 1 MB (+ slack), finite bf16 pairs
    const size_t src_bytes = (size_t)1018 << 20;
    CK(hipMalloc(&src, src_bytes));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, s, (uint32_t*)src, src_bytes / 4);
    CK(hipStreamSynchronize(s));
    for (int rep = 0; rep < 2; ++rep) {
        run<0, 0>("16x16x32 registers only", out, n_cu, s);
        run<1, 0>("32x32x16 registers only", out, n_cu, s);
        run<0, 16>("16x16x32 + LDS reads  8 B/clk/SIMD", out, n_cu, s);
        run<0, 32>("16x16x32 + LDS reads 16 B/clk/SIMD (GEMM)", out, n_cu, s);
        run<0, 64>("16x16x32 + LDS reads 32 B/clk/SIMD", out, n_cu, s);
        run<0, 0, 16>("16x16x32 + LDS-DMA 32 B/clk/CU", out, n_cu, s, src);
        run<0, 32, 16>("16x16x32 + LDS reads 16 + LDS-DMA 32 (GEMM)", out, n_cu, s, src);
        run<0, 32, 32>("16x16x32 + LDS reads 16 + LDS-DMA 64", out, n_cu, s, src);
        run<0, 0, 32>("16x16x32 + LDS-DMA 64 alone", out, n_cu, s, src);
        run<0, 64, 32>("16x16x32 + LDS reads 32 + LDS-DMA 64", out, n_cu, s, src);
        run<0, 32, 32, 1>("  reads 16 + DMA 64, a GEMM's panels (A in L2, W from HBM)", out, n_cu, s, src);
        run<0, 0, 32, 1>("16x16x32 + LDS-DMA 64, a GEMM's panels", out, n_cu, s, src);
        run<0, 32, 32, 4>("  reads + DMA 64, panels, W not walking", out, n_cu, s, src);
        run<0, 32, 32, 5>("  reads + DMA 64, panels, W walk wraps in 128 MB", out, n_cu, s, src);
        run<0, 32, 32, 6>("  reads + DMA 64, panels, W walk wraps in 64 MB", out, n_cu, s, src);
        run<0, 32, 32, 5, 16>("  ... wraps in 128 MB, hand-over: 16 in flight", out, n_cu, s, src);
        run<0, 32, 32, 5, 24>("  ... wraps in 128 MB, hand-over: 24 in flight", out, n_cu, s, src);
        run<0, 32, 32, 5, 32>("  ... wraps in 128 MB, hand-over: 32 in flight", out, n_cu, s, src);
        run<0, 32, 32, 1, 16>("  ... W from HBM, hand-over: 16 in flight", out, n_cu, s, src);
        run<0, 32, 32, 1, 32>("  ... W from HBM, hand-over: 32 in flight", out, n_cu, s, src);
        run<0, 32, 32, 2>("  reads + DMA 64, panels, W K-tile-major", out, n_cu, s, src);
        run<0, 32, 32, 3>("  reads + DMA 64, panels, A and W K-tile-major", out, n_cu, s, src);
        run<1, 16>("32x32x16 + LDS reads  8 B/clk/SIMD", out, n_cu, s);
        run<1, 32>("32x32x16 + LDS reads 16 B/clk/SIMD (GEMM)", out, n_cu, s);
        run<1, 128>("32x32x16 + LDS reads 32 B/clk/SIMD", out, n_cu, s);
    }
    // Does reading W right before the loop that streams it help?  One pass over 144 MB of W panels (144 trips), the memory-side cache
    // flushed by a 600 MB read before every launch; "touched": the 144 MB read once more between the flush and the launch.
    for (int touched = 0; touched < 2; ++touched) {
        double cyc = 0, ref = 0;
        const int reps = 12, trips = 144;
        static uint64_t h[4096];
        for (int r = 0; r < reps; ++r) {
            hipLaunchKernelGGL(touch_kernel, dim3(2048), dim3(256), 0, s, (const uint4*)(src + ((size_t)416 << 20)), ((size_t)600 << 20) / 16);
            if (touched) hipLaunchKernelGGL(touch_kernel, dim3(2048), dim3(256), 0, s, (const uint4*)(src + ((size_t)16 << 20)), ((size_t)144 << 20) / 16);
            hipLaunchKernelGGL((stream_kernel<0, 32, 32, 1, 16>), dim3(n_cu), dim3(256), 32768, s, out, trips, src, g_clk);
            CK(hipStreamSynchronize(s));
            CK(hipMemcpy(h, g_clk, (size_t)n_cu * 16, hipMemcpyDeviceToHost));
            if (r >= 2) for (int i = 0; i < n_cu; ++i) { cyc += (double)h[2 * i]; ref += (double)h[2 * i + 1]; }
        }
        printf("one pass over 144 MB of W, cache flushed before%s: %6.0f clk per 2048 MFMA clk at %.3f GHz\n",
               touched ? ", W read once right before the launch" : "", cyc / ((reps - 2) * n_cu) / trips, cyc / ref * 0.1);
    }
    CK(hipFree(out));
    CK(hipFree(src));
    return 0;
}
