// imaging.hip — pixel-space pre/post-processing of the inpainting / outpainting chains on the device:
// u8 resampling (mask -> latent grid, generated crop -> crop size), mask bounding box, pixel composite,
// u8 image -> VAE input.  Byte work, HBM-bound (one read + one write per pixel and tap), no MFMA.
//
// Reference: Flux2Pipeline.packMaskForLatentBlending (Flux2Core/Pipeline/Flux2Pipeline+ChainHelpers.swift:148-236),
// Flux2InpaintCompositing.{maskBoundingBox, expandCropRegion, composite, renderInpaintWeights}
// (Flux2Chains/Flux2InpaintCompositing.swift:23-227), preprocessImageForVAE (Flux2Pipeline.swift:2359-2418).
// The reference resamples with CoreGraphics' closed `.high` filter; here the filter is the antialiased triangle
// filter in Pillow's 8-bit fixed-point form (22-bit coefficients, horizontal then vertical pass, u8 intermediate),
// bit-exact against Pillow's Image.resize(BILINEAR, box=) — see oracle/imaging_ref.py for the parity statement.
#include <math.h>
#include <algorithm>
#include <vector>
#include "common.h"
#include "dev.h"

// every float expression in this file is a restatement of scalar Swift / C arithmetic: no fused multiply-add
// (plain operators under this pragma: HIP's __fmul_rn / __fadd_rn are header inlines that still carry the default
// -ffp-contract=fast and get fused after inlining)
#pragma clang fp contract(off)

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

struct AxisCoeffs {
    int out = 0, ksize = 0;
    std::vector<int32_t> bounds;  // [out][2] = first tap, tap count
    std::vector<int32_t> kk;      // [out][ksize]
};

// Pillow precompute_coeffs + normalize_coeffs_8bpc, bilinear filter (support 1)
AxisCoeffs axis_coeffs(int in_size, double in0, double in1, int out_size) {
    AxisCoeffs c;
    c.out = out_size;
    const double scale = (in1 - in0) / out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale;
    c.ksize = (int)ceil(support) * 2 + 1;
    c.bounds.assign((size_t)out_size * 2, 0);
    c.kk.assign((size_t)out_size * c.ksize, 0);
    const double ss = 1.0 / filterscale;
    std::vector<double> w(c.ksize);
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = in0 + (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        if (xmax < 0) xmax = 0;
        if (xmax > c.ksize) xmax = c.ksize;
        double ww = 0.0;
        for (int x = 0; x < xmax; ++x) {
            double t = (x + xmin - center + 0.5) * ss;
            if (t < 0.0) t = -t;
            w[x] = t < 1.0 ? 1.0 - t : 0.0;
            ww += w[x];
        }
        for (int x = 0; x < xmax; ++x) {
            const double v = ww != 0.0 ? w[x] / ww : w[x];
            c.kk[(size_t)xx * c.ksize + x] = v < 0 ? (int32_t)(-0.5 + v * (1 << PRECISION_BITS)) : (int32_t)(0.5 + v * (1 << PRECISION_BITS));
        }
        c.bounds[(size_t)xx * 2] = xmin;
        c.bounds[(size_t)xx * 2 + 1] = xmax;
    }
    return c;
}

__device__ __forceinline__ uint8_t clip8(int32_t acc) {
    const int32_t v = acc >> PRECISION_BITS;
    return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
}

// horizontal pass: src u8 [*, src_w, src_c] rows [row0, row0+rows) -> tmp u8 [rows, out_w, nc], channels c0..c0+nc
__global__ void resample_h_kernel(const uint8_t* __restrict__ src, int src_w, int src_c, int c0, int nc, int row0, int rows,
                                  uint8_t* __restrict__ tmp, int out_w, const int32_t* __restrict__ bounds,
                                  const int32_t* __restrict__ kk, int ksize) {
    const int64_t total = (int64_t)rows * out_w * nc;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % nc);
        const int xo = (int)((i / nc) % out_w);
        const int y = (int)(i / ((int64_t)nc * out_w));
        const int lo = bounds[2 * xo], n = bounds[2 * xo + 1];
        const uint8_t* s = src + ((int64_t)(row0 + y) * src_w + lo) * src_c + c0 + c;
        const int32_t* k = kk + (int64_t)xo * ksize;
        int32_t acc = 1 << (PRECISION_BITS - 1);
        for (int t = 0; t < n; ++t) acc += (int32_t)s[(int64_t)t * src_c] * k[t];
        tmp[i] = clip8(acc);
    }
}
// vertical pass: tmp u8 [rows, out_w, nc] (row 0 = source row `row0`) -> dst u8 [out_h, out_w, nc]
__global__ void resample_v_kernel(const uint8_t* __restrict__ tmp, int row0, uint8_t* __restrict__ dst, int out_h, int out_w,
                                  int nc, const int32_t* __restrict__ bounds, const int32_t* __restrict__ kk, int ksize) {
    const int64_t row_elems = (int64_t)out_w * nc;
    const int64_t total = (int64_t)out_h * row_elems;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int yo = (int)(i / row_elems);
        const int64_t col = i % row_elems;
        const int lo = bounds[2 * yo], n = bounds[2 * yo + 1];
        const uint8_t* s = tmp + (int64_t)(lo - row0) * row_elems + col;
        const int32_t* k = kk + (int64_t)yo * ksize;
        int32_t acc = 1 << (PRECISION_BITS - 1);
        for (int t = 0; t < n; ++t) acc += (int32_t)s[(int64_t)t * row_elems] * k[t];
        dst[i] = clip8(acc);
    }
}
// u8 -> inpaint weight: grayscale v/255, alpha 1 - a/255 (renderInpaintWeights :196-227)
__global__ void mask_weights_kernel(const uint8_t* __restrict__ in, float* __restrict__ out, int64_t n, int invert) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = (float)in[i] / 255.0f;
        out[i] = invert ? 1.0f - v : v;
    }
}
// bounding box of weights > threshold: box = {min_x, min_y, max_x, max_y}, integer atomics (order-independent)
__global__ void mask_bbox_kernel(const float* __restrict__ w, int h, int wd, float threshold, int* __restrict__ box) {
    const int64_t n = (int64_t)h * wd;
    int mnx = 0x7fffffff, mny = 0x7fffffff, mxx = -1, mxy = -1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (w[i] > threshold) {
            const int x = (int)(i % wd), y = (int)(i / wd);
            mnx = min(mnx, x), mny = min(mny, y), mxx = max(mxx, x), mxy = max(mxy, y);
        }
    }
    if (mxx >= 0) {
        atomicMin(box + 0, mnx);
        atomicMin(box + 1, mny);
        atomicMax(box + 2, mxx);
        atomicMax(box + 3, mxy);
    }
}
// composite (:112-162): m <= 0 keep, m >= 1 paste, else u8(trunc(clamp(base*(1-m) + gen*m))) in f32, two products
// and one sum as the reference computes them (no fused multiply-add)
__global__ void composite_kernel(uint8_t* __restrict__ base, int full_w, int base_c, const uint8_t* __restrict__ gen, int gen_c,
                                 int crop_x, int crop_y, int crop_w, int crop_h, const float* __restrict__ weights) {
    const int64_t n = (int64_t)crop_w * crop_h;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float m = weights[i];
        if (m <= 0.f) continue;
        const int x = (int)(i % crop_w), y = (int)(i / crop_w);
        uint8_t* b = base + ((int64_t)(crop_y + y) * full_w + crop_x + x) * base_c;
        const uint8_t* g = gen + i * gen_c;
        if (m >= 1.f) {
            b[0] = g[0], b[1] = g[1], b[2] = g[2];
        } else {
            const float inv = 1.0f - m;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float vb = (float)b[c] * inv, vg = (float)g[c] * m;
                float v = vb + vg;
                v = fmaxf(0.f, fminf(255.f, v));
                b[c] = (uint8_t)v;
            }
        }
    }
}
// u8 [H,W,C] -> f32 planar [3,H,W] = x / 127.5 - 1 (Flux2Pipeline.swift:2359-2418)
__global__ void image_to_vae_input_kernel(const uint8_t* __restrict__ img, int64_t hw, int c, float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += (int64_t)gridDim.x * blockDim.x) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) out[ch * hw + i] = (float)img[i * c + ch] / 127.5f - 1.0f;
    }
}

// Flux2OutpaintingChain.buildOutpaintCanvas (Flux2Chains/Flux2OutpaintingChain.swift:331-395): grey Gaussian noise
// (mean 128, sigma 32; SplitMix64 + Box-Muller in f64, two draws per pixel) with the source pasted at the offset.
// SplitMix64's state is a counter (s0 + j*G), so pixel i takes draws 2i+1 and 2i+2 directly.
__device__ __forceinline__ uint64_t splitmix64_at(uint64_t s0, uint64_t j) {
    uint64_t z = s0 + j * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ void outpaint_canvas_kernel(const uint8_t* __restrict__ src, int src_h, int src_w, int src_c, uint8_t* __restrict__ canvas,
                                       int canvas_h, int canvas_w, int off_x, int off_y, uint64_t s0) {
    const int64_t n = (int64_t)canvas_h * canvas_w;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % canvas_w), y = (int)(i / canvas_w);
        uint8_t* o = canvas + i * 3;
        if (x >= off_x && x < off_x + src_w && y >= off_y && y < off_y + src_h) {
            const uint8_t* p = src + ((int64_t)(y - off_y) * src_w + (x - off_x)) * src_c;
            o[0] = p[0], o[1] = p[1], o[2] = p[2];
            continue;
        }
        const double u1 = ((double)(splitmix64_at(s0, 2 * (uint64_t)i + 1) >> 11) + 1.0) / 9007199254740992.0;
        const double u2 = ((double)(splitmix64_at(s0, 2 * (uint64_t)i + 2) >> 11) + 1.0) / 9007199254740992.0;
        const double g = 128.0 + (sqrt(-2.0 * log(u1)) * cos(2.0 * 3.14159265358979323846 * u2)) * 32.0;
        long long v = (long long)round(g);  // .rounded(): half away from zero
        v = v < 0 ? 0 : v > 255 ? 255 : v;
        o[0] = o[1] = o[2] = (uint8_t)v;
    }
}
// Flux2OutpaintingChain.buildSmartMask (:397-474): white strips, black keep, linear ramps inside the keep on the
// sides that have a strip, band capped at half the keep
__global__ void outpaint_mask_kernel(uint8_t* __restrict__ mask, int canvas_h, int canvas_w, int keep_x, int keep_y, int keep_w,
                                     int keep_h, int transition) {
    const int64_t n = (int64_t)canvas_h * canvas_w;
    const int band_w = max(1, min(transition, keep_w / 2)), band_h = max(1, min(transition, keep_h / 2));
    const bool has_left = keep_x > 0, has_right = keep_x + keep_w < canvas_w, has_top = keep_y > 0, has_bottom = keep_y + keep_h < canvas_h;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % canvas_w) - keep_x, y = (int)(i / canvas_w) - keep_y;
        int v = 255;
        if (x >= 0 && x < keep_w && y >= 0 && y < keep_h) {
            v = 0;
            if (has_left && x < band_w) v = max(v, (255 * (band_w - x) + band_w / 2) / band_w);
            if (has_right && keep_w - 1 - x < band_w) v = max(v, (255 * (band_w - (keep_w - 1 - x)) + band_w / 2) / band_w);
            if (has_top && y < band_h) v = max(v, (255 * (band_h - y) + band_h / 2) / band_h);
            if (has_bottom && keep_h - 1 - y < band_h) v = max(v, (255 * (band_h - (keep_h - 1 - y)) + band_h / 2) / band_h);
        }
        mask[i] = (uint8_t)v;
    }
}

inline unsigned grid_for(int64_t n) {
    int64_t b = (n + 255) / 256;
    if (b > 256 * 16) b = 256 * 16;
    return (unsigned)(b < 1 ? 1 : b);
}

// resample channels [c0, c0+nc) of src (u8 [src_h, src_w, src_c]) over the source rectangle `box` into
// dst (u8 [out_h, out_w, nc]); device scratch (tmp rows + tables) comes from ctx->scratch
int resample_impl(flux2mi_ctx* ctx, const uint8_t* src, int src_h, int src_w, int src_c, int c0, int nc, const double box[4],
                  uint8_t* dst, int out_h, int out_w) {
    const AxisCoeffs ch = axis_coeffs(src_w, box[0], box[2], out_w);
    const AxisCoeffs cv = axis_coeffs(src_h, box[1], box[3], out_h);
    int row0 = src_h, row1 = 0;
    for (int y = 0; y < out_h; ++y) {
        row0 = std::min(row0, cv.bounds[2 * y]);
        row1 = std::max(row1, cv.bounds[2 * y] + cv.bounds[2 * y + 1]);
    }
    if (row1 <= row0) { row0 = 0; row1 = 1; }
    const int rows = row1 - row0;
    auto pad = [](size_t n) { return (n + 255) & ~size_t(255); };
    const size_t tmp_bytes = pad((size_t)rows * out_w * nc);
    const size_t tab[4] = {pad(ch.bounds.size() * 4), pad(ch.kk.size() * 4), pad(cv.bounds.size() * 4), pad(cv.kk.size() * 4)};
    F2_TRY(ctx->scratch.reserve(tmp_bytes + tab[0] + tab[1] + tab[2] + tab[3]));
    uint8_t* tmp = ctx->scratch.as<uint8_t>();
    uint8_t* p = tmp + tmp_bytes;
    int32_t* d_bh = reinterpret_cast<int32_t*>(p); p += tab[0];
    int32_t* d_kh = reinterpret_cast<int32_t*>(p); p += tab[1];
    int32_t* d_bv = reinterpret_cast<int32_t*>(p); p += tab[2];
    int32_t* d_kv = reinterpret_cast<int32_t*>(p);
    hipStream_t st = ctx->stream;
    F2_CHECK_HIP(hipMemcpyAsync(d_bh, ch.bounds.data(), ch.bounds.size() * 4, hipMemcpyHostToDevice, st));
    F2_CHECK_HIP(hipMemcpyAsync(d_kh, ch.kk.data(), ch.kk.size() * 4, hipMemcpyHostToDevice, st));
    F2_CHECK_HIP(hipMemcpyAsync(d_bv, cv.bounds.data(), cv.bounds.size() * 4, hipMemcpyHostToDevice, st));
    F2_CHECK_HIP(hipMemcpyAsync(d_kv, cv.kk.data(), cv.kk.size() * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(resample_h_kernel, dim3(grid_for((int64_t)rows * out_w * nc)), dim3(256), 0, st, src, src_w, src_c, c0, nc, row0,
                       rows, tmp, out_w, d_bh, d_kh, ch.ksize);
    hipLaunchKernelGGL(resample_v_kernel, dim3(grid_for((int64_t)out_h * out_w * nc)), dim3(256), 0, st, tmp, row0, dst, out_h, out_w, nc,
                       d_bv, d_kv, cv.ksize);
    F2_CHECK_HIP(hipGetLastError());
    F2_CHECK_HIP(hipStreamSynchronize(st));  // the host tables above leave scope
    return FLUX2MI_OK;
}

bool valid_box(const float* box, int src_h, int src_w, double out[4]) {
    if (!box) {
        out[0] = 0, out[1] = 0, out[2] = src_w, out[3] = src_h;
        return true;
    }
    for (int i = 0; i < 4; ++i) out[i] = box[i];
    return out[0] >= 0 && out[1] >= 0 && out[2] <= src_w && out[3] <= src_h && out[2] > out[0] && out[3] > out[1];
}

}  // namespace

extern "C" int flux2mi_image_resample_u8(flux2mi_ctx* ctx, const uint8_t* src, int src_h, int src_w, int channels, const float* box,
                                         uint8_t* dst, int dst_h, int dst_w) {
    F2_REQUIRE(ctx && src && dst && src_h > 0 && src_w > 0 && dst_h > 0 && dst_w > 0 && channels >= 1 && channels <= 4,
               FLUX2MI_ERR_INVALID_ARG, "image_resample_u8: bad argument");
    F2_WORKSPACE_USE(ctx->device);
    double b[4];
    F2_REQUIRE(valid_box(box, src_h, src_w, b), FLUX2MI_ERR_INVALID_ARG, "image_resample_u8: box outside the image");
    return resample_impl(ctx, src, src_h, src_w, channels, 0, channels, b, dst, dst_h, dst_w);
}

extern "C" int flux2mi_mask_inpaint_weights(flux2mi_ctx* ctx, const uint8_t* mask, int mask_h, int mask_w, int channels, int convention,
                                            const float* box, float* weights, int out_h, int out_w) {
    F2_REQUIRE(ctx && mask && weights && mask_h > 0 && mask_w > 0 && out_h > 0 && out_w > 0 && (channels == 1 || channels == 3 || channels == 4),
               FLUX2MI_ERR_INVALID_ARG, "mask_inpaint_weights: bad argument");
    F2_WORKSPACE_USE(ctx->device);
    F2_REQUIRE(convention == FLUX2MI_MASK_GRAYSCALE_WHITE_INPAINT || (convention == FLUX2MI_MASK_ALPHA_TRANSPARENT_INPAINT && channels == 4),
               FLUX2MI_ERR_INVALID_ARG, "mask_inpaint_weights: the alpha convention needs an RGBA mask");
    double b[4];
    F2_REQUIRE(valid_box(box, mask_h, mask_w, b), FLUX2MI_ERR_INVALID_ARG, "mask_inpaint_weights: box outside the mask");
    // the reference renders into an 8-bit context first: stage the u8 rendering, then convert.  A 3-channel mask is
    // read through its first channel (a grey RGB image; CoreGraphics would take the luminance)
    DevBuf stage;
    F2_TRY(stage.reserve((size_t)out_h * out_w));
    const int c0 = convention == FLUX2MI_MASK_ALPHA_TRANSPARENT_INPAINT ? 3 : 0;
    int s = resample_impl(ctx, mask, mask_h, mask_w, channels, c0, 1, b, stage.as<uint8_t>(), out_h, out_w);
    if (s == FLUX2MI_OK) {
        const int64_t n = (int64_t)out_h * out_w;
        hipLaunchKernelGGL(mask_weights_kernel, dim3(grid_for(n)), dim3(256), 0, ctx->stream, stage.as<uint8_t>(), weights, n,
                           convention == FLUX2MI_MASK_ALPHA_TRANSPARENT_INPAINT ? 1 : 0);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) {
            flux2mi_set_error("mask_inpaint_weights: kernel failed");
            s = FLUX2MI_ERR_HIP;
        }
    }
    stage.release();
    return s;
}

extern "C" int flux2mi_mask_bounding_box(flux2mi_ctx* ctx, const uint8_t* mask, int mask_h, int mask_w, int channels, int convention,
                                         int image_width, int image_height, float threshold, int out_rect[4], int* found) {
    F2_REQUIRE(ctx && mask && out_rect && found && image_width > 0 && image_height > 0, FLUX2MI_ERR_INVALID_ARG, "mask_bounding_box: bad argument");
    F2_WORKSPACE_USE(ctx->device);
    *found = 0;
    // :31-33 — scan at <= 512 on the long side
    const double scale = std::min(1.0, 512.0 / (double)std::max(mask_w, mask_h));
    const int scan_w = std::max(1, (int)((double)mask_w * scale));
    const int scan_h = std::max(1, (int)((double)mask_h * scale));
    DevBuf wbuf;
    F2_TRY(wbuf.reserve((size_t)scan_w * scan_h * 4 + 16));
    float* w = wbuf.as<float>();
    int* d_box = reinterpret_cast<int*>(w + (size_t)scan_w * scan_h);
    int s = flux2mi_mask_inpaint_weights(ctx, mask, mask_h, mask_w, channels, convention, nullptr, w, scan_h, scan_w);
    int box[4] = {0x7fffffff, 0x7fffffff, -1, -1};
    if (s == FLUX2MI_OK) {
        hipError_t e = hipMemcpyAsync(d_box, box, sizeof(box), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(mask_bbox_kernel, dim3(grid_for((int64_t)scan_w * scan_h)), dim3(256), 0, ctx->stream, w, scan_h, scan_w,
                               threshold, d_box);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(box, d_box, sizeof(box), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            flux2mi_set_error("mask_bounding_box: %s", hipGetErrorString(e));
            s = FLUX2MI_ERR_HIP;
        }
    }
    wbuf.release();
    if (s != FLUX2MI_OK) return s;
    if (box[2] < 0) return FLUX2MI_OK;  // empty mask
    // :52-63 — back to image space, rounded outward by one scan pixel
    const double sx = (double)image_width / (double)scan_w, sy = (double)image_height / (double)scan_h;
    const int x0 = std::max(0, (int)((double)(box[0] - 1) * sx));
    const int y0 = std::max(0, (int)((double)(box[1] - 1) * sy));
    const int x1 = std::min(image_width, (int)((double)(box[2] + 2) * sx));
    const int y1 = std::min(image_height, (int)((double)(box[3] + 2) * sy));
    if (!(x1 > x0 && y1 > y0)) return FLUX2MI_OK;
    out_rect[0] = x0, out_rect[1] = y0, out_rect[2] = x1 - x0, out_rect[3] = y1 - y0;
    *found = 1;
    return FLUX2MI_OK;
}

// Flux2InpaintCompositing.expandCropRegion (:69-108): pure integer/double host logic
extern "C" void flux2mi_expand_crop_region(const int bbox[4], int padding, int image_width, int image_height, int out_rect[4]) {
    int x0 = bbox[0] - padding, y0 = bbox[1] - padding;
    int x1 = bbox[0] + bbox[2] + padding, y1 = bbox[1] + bbox[3] + padding;
    const double target = (double)image_width / (double)image_height;
    const int w = x1 - x0, h = y1 - y0;
    if ((double)w / (double)h > target) {
        const int grow = (int)ceil((double)w / target) - h;
        y0 -= grow / 2;
        y1 += grow - grow / 2;
    } else {
        const int grow = (int)ceil((double)h * target) - w;
        x0 -= grow / 2;
        x1 += grow - grow / 2;
    }
    if (x0 < 0) { x1 -= x0; x0 = 0; }
    if (y0 < 0) { y1 -= y0; y0 = 0; }
    if (x1 > image_width) { x0 -= (x1 - image_width); x1 = image_width; x0 = std::max(0, x0); }
    if (y1 > image_height) { y0 -= (y1 - image_height); y1 = image_height; y0 = std::max(0, y0); }
    out_rect[0] = x0, out_rect[1] = y0, out_rect[2] = x1 - x0, out_rect[3] = y1 - y0;
}

extern "C" int flux2mi_composite_u8(flux2mi_ctx* ctx, uint8_t* base, int full_h, int full_w, int base_channels, const uint8_t* generated,
                                    int gen_channels, const int crop_rect[4], const float* weights) {
    F2_REQUIRE(ctx && base && generated && crop_rect && weights && (base_channels == 3 || base_channels == 4) &&
                   (gen_channels == 3 || gen_channels == 4),
               FLUX2MI_ERR_INVALID_ARG, "composite_u8: bad argument");
    F2_WORKSPACE_USE(ctx->device);
    const int cx = crop_rect[0], cy = crop_rect[1], cw = crop_rect[2], ch = crop_rect[3];
    // :124-126 — the reference returns nil for a crop that leaves the image
    F2_REQUIRE(cw > 0 && ch > 0 && cx >= 0 && cy >= 0 && cx + cw <= full_w && cy + ch <= full_h, FLUX2MI_ERR_SHAPE,
               "composite_u8: crop rect %d,%d %dx%d leaves the %dx%d image", cx, cy, cw, ch, full_w, full_h);
    hipLaunchKernelGGL(composite_kernel, dim3(grid_for((int64_t)cw * ch)), dim3(256), 0, ctx->stream, base, full_w, base_channels, generated,
                       gen_channels, cx, cy, cw, ch, weights);
    F2_CHECK_HIP(hipGetLastError());
    return FLUX2MI_OK;
}

extern "C" int flux2mi_preprocess_image_for_vae(flux2mi_ctx* ctx, const uint8_t* image, int height, int width, int channels, float* out) {
    F2_REQUIRE(ctx && image && out && height > 0 && width > 0 && (channels == 3 || channels == 4), FLUX2MI_ERR_INVALID_ARG,
               "preprocess_image_for_vae: bad argument");
    F2_WORKSPACE_USE(ctx->device);
    const int64_t hw = (int64_t)height * width;
    hipLaunchKernelGGL(image_to_vae_input_kernel, dim3(grid_for(hw)), dim3(256), 0, ctx->stream, image, hw, channels, out);
    F2_CHECK_HIP(hipGetLastError());
    return FLUX2MI_OK;
}

extern "C" int flux2mi_outpaint_canvas(flux2mi_ctx* ctx, const uint8_t* source, int src_h, int src_w, int channels, uint8_t* canvas,
                                       int canvas_h, int canvas_w, int offset_x, int offset_y, uint64_t noise_seed) {
    F2_REQUIRE(ctx && source && canvas && src_h > 0 && src_w > 0 && (channels == 3 || channels == 4) && canvas_h > 0 && canvas_w > 0 &&
                   offset_x >= 0 && offset_y >= 0 && offset_x + src_w <= canvas_w && offset_y + src_h <= canvas_h,
               FLUX2MI_ERR_INVALID_ARG, "outpaint_canvas: the source does not fit the canvas at the offset");
    F2_WORKSPACE_USE(ctx->device);
    const uint64_t s0 = noise_seed + 0x9E3779B97F4A7C15ull;  // :345
    hipLaunchKernelGGL(outpaint_canvas_kernel, dim3(grid_for((int64_t)canvas_h * canvas_w)), dim3(256), 0, ctx->stream, source, src_h, src_w,
                       channels, canvas, canvas_h, canvas_w, offset_x, offset_y, s0);
    F2_CHECK_HIP(hipGetLastError());
    return FLUX2MI_OK;
}

extern "C" int flux2mi_outpaint_smart_mask(flux2mi_ctx* ctx, uint8_t* mask, int canvas_h, int canvas_w, int keep_x, int keep_y, int keep_w,
                                           int keep_h, int transition_pixels) {
    F2_REQUIRE(ctx && mask && canvas_h > 0 && canvas_w > 0 && keep_x >= 0 && keep_y >= 0 && keep_w > 0 && keep_h > 0 &&
                   keep_x + keep_w <= canvas_w && keep_y + keep_h <= canvas_h,
               FLUX2MI_ERR_INVALID_ARG, "outpaint_smart_mask: the keep rectangle leaves the canvas");
    F2_WORKSPACE_USE(ctx->device);
    hipLaunchKernelGGL(outpaint_mask_kernel, dim3(grid_for((int64_t)canvas_h * canvas_w)), dim3(256), 0, ctx->stream, mask, canvas_h, canvas_w,
                       keep_x, keep_y, keep_w, keep_h, transition_pixels);
    F2_CHECK_HIP(hipGetLastError());
    return FLUX2MI_OK;
}
