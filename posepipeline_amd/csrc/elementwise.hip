// PP_OP_UPSAMPLE_ADD: out[n][y][x][c] = act((up(t)[n][y][x][c] + res1[n][y][x][c]) + res2[...]),  up = nearest 2^u.
//
// HRNet's fuse layers (mmpose HRModule: 1x1 conv + BN on the coarser branch, nearest upsample, `y += ...`) used to
// scatter each conv result over its 2^u x 2^u patch from inside the conv epilogue.  The conv of a coarse branch has few
// tiles (8x6 maps: 48 blocks for a batch of 128), and each of those blocks then read-modified-wrote up to 64x its own
// output: 0.29 ms for 50 MB.  Here the conv writes its small fp32 result once and this kernel -- one float4 per thread,
// every CU busy, HBM-bound -- does the upsample + add.  The additions are the same fp32 operations in the same order
// as the conv epilogue's ((acc + bias) + res1) + res2, so results are bit-identical.
#include <algorithm>

#include "pp_internal.h"
#include "pp_amax.h"

namespace {

// UA_R float4 per thread (consecutive 4 KB chunks of the block): with the maximum tracked for a fp16-form reader (pp_amax.h) a block
// then issues ONE atomic per sample for 32 KB of output -- same-address atomics serialise at ~0.6 us each at the memory side
constexpr int UA_R = 8;
__global__ __launch_bounds__(256) void upsample_add_kernel(const float4* __restrict__ t, const float4* __restrict__ r1,
                                                           const float4* __restrict__ r2, float4* __restrict__ y, size_t total,
                                                           int H, int W, int c4, int up, int relu, const float4* __restrict__ t2,
                                                           int up2, const float4* __restrict__ t3, int up3, unsigned* y_amax) {
    const size_t per = (size_t)H * W * c4;            // float4 per sample
    const size_t b0 = (size_t)blockIdx.x * (256 * UA_R);
    const int img_first = (int)(b0 / per);
    float ym[2] = {0.f, 0.f};                         // max |y| of this thread's elements in sample img_first / img_first + 1
#pragma unroll
    for (int r = 0; r < UA_R; ++r) {
        const size_t i = b0 + (size_t)r * 256 + threadIdx.x;
        if (i >= total) break;
        const int cc = (int)(i % c4);
        size_t p = i / c4;
        const int x = (int)(p % W);
        p /= W;
        const int yy = (int)(p % H);
        const size_t n = p / H;
        const int hs = H >> up, ws = W >> up;
        float4 v = t[((n * hs + (yy >> up)) * ws + (x >> up)) * c4 + cc];
        if (r1) {
            const float4 a = r1[i];
            v.x = __fadd_rn(v.x, a.x); v.y = __fadd_rn(v.y, a.y); v.z = __fadd_rn(v.z, a.z); v.w = __fadd_rn(v.w, a.w);
        }
        if (t2) {          // further coarse terms, in mmpose's `y += ...` order
            const float4 a = t2[((n * (H >> up2) + (yy >> up2)) * (W >> up2) + (x >> up2)) * c4 + cc];
            v.x = __fadd_rn(v.x, a.x); v.y = __fadd_rn(v.y, a.y); v.z = __fadd_rn(v.z, a.z); v.w = __fadd_rn(v.w, a.w);
        }
        if (t3) {
            const float4 a = t3[((n * (H >> up3) + (yy >> up3)) * (W >> up3) + (x >> up3)) * c4 + cc];
            v.x = __fadd_rn(v.x, a.x); v.y = __fadd_rn(v.y, a.y); v.z = __fadd_rn(v.z, a.z); v.w = __fadd_rn(v.w, a.w);
        }
        if (r2) {
            const float4 a = r2[i];
            v.x = __fadd_rn(v.x, a.x); v.y = __fadd_rn(v.y, a.y); v.z = __fadd_rn(v.z, a.z); v.w = __fadd_rn(v.w, a.w);
        }
        if (relu) {
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        y[i] = v;
        if (y_amax) {
            const float m = pp_abs4max(v);
            const int d = (int)n - img_first;
            if (d == 0) ym[0] = fmaxf(ym[0], m);
            else if (d == 1) ym[1] = fmaxf(ym[1], m);
            else if (m > 0.f) atomicMax(y_amax + n, __float_as_uint(m));      // samples smaller than a quarter of a block: rare, direct
        }
    }
    if (y_amax) {
        __shared__ float red[16];
        const size_t last = (b0 + 256 * UA_R - 1 < total ? b0 + 256 * UA_R - 1 : total - 1) / per;
        const int im[2] = {img_first, img_first + 1};
        pp_amax_commit_wg<4, 2>(y_amax, im, ym, img_first, (int)last > img_first ? img_first + 1 : img_first, red);
    }
}

// PP_OP_AVGPOOL: nn.AvgPool2d((kh, kw), stride) without padding (the GlobalAveragePooling neck of mmtrack's ReID model,
// mot/deepsort/deepsort_*.py:27): float32 sum over the window in (kh, kw) order, divided by the window size.
__global__ __launch_bounds__(256) void avgpool_kernel(const float4* __restrict__ x, float4* __restrict__ y, size_t total, int Hin,
                                                      int Win, int Hout, int Wout, int c4, int kh, int kw, int stride) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int cc = (int)(i % c4);
    size_t p = i / c4;
    const int wo = (int)(p % Wout);
    p /= Wout;
    const int ho = (int)(p % Hout);
    const size_t n = p / Hout;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int a = 0; a < kh; ++a)
        for (int b = 0; b < kw; ++b) {
            const float4 v = x[((n * Hin + (size_t)(ho * stride + a)) * Win + (wo * stride + b)) * c4 + cc];
            acc.x = __fadd_rn(acc.x, v.x); acc.y = __fadd_rn(acc.y, v.y); acc.z = __fadd_rn(acc.z, v.z); acc.w = __fadd_rn(acc.w, v.w);
        }
    const float d = (float)(kh * kw);
    y[i] = make_float4(__fdiv_rn(acc.x, d), __fdiv_rn(acc.y, d), __fdiv_rn(acc.z, d), __fdiv_rn(acc.w, d));
}

// F.interpolate(crop, size, mode='bilinear', align_corners=False) of an integer rectangle of an NHWC4 float32 image
// (mmtrack SortTracker.crop_imgs: the ReID crops are cut out of the detector's normalised input tensor).  PyTorch's
// upsample_bilinear2d arithmetic in float32: src = max(0, scale * (dst + 0.5) - 0.5), scale = in / out;
// val = h0 * (w0 * p00 + w1 * p01) + h1 * (w0 * p10 + w1 * p11).
struct CropRect { int frame, x1, y1, x2, y2; };
__global__ __launch_bounds__(256) void crop_resize_kernel(const float4* __restrict__ src, int src_h, int src_w, const CropRect* __restrict__ rects,
                                                          int out_h, int out_w, float4* __restrict__ out) {
    const int r = blockIdx.y;
    const CropRect q = rects[r];
    const int ch = q.y2 - q.y1, cw = q.x2 - q.x1;
    const float rh = __fdiv_rn((float)ch, (float)out_h), rw = __fdiv_rn((float)cw, (float)out_w);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < out_h * out_w; i += gridDim.x * 256) {
        const int h2 = i / out_w, w2 = i - h2 * out_w;
        const float h1r = fmaxf(__fsub_rn(__fmul_rn(rh, __fadd_rn((float)h2, 0.5f)), 0.5f), 0.f);
        const float w1r = fmaxf(__fsub_rn(__fmul_rn(rw, __fadd_rn((float)w2, 0.5f)), 0.5f), 0.f);
        const int h1 = (int)h1r, w1 = (int)w1r;
        const int h1p = h1 < ch - 1 ? 1 : 0, w1p = w1 < cw - 1 ? 1 : 0;
        const float h1l = __fsub_rn(h1r, (float)h1), h0l = __fsub_rn(1.f, h1l);
        const float w1l = __fsub_rn(w1r, (float)w1), w0l = __fsub_rn(1.f, w1l);
        const float4* base = src + ((size_t)q.frame * src_h + (q.y1 + h1)) * src_w + (q.x1 + w1);
        const float4 p00 = base[0], p01 = base[w1p], p10 = base[(size_t)h1p * src_w], p11 = base[(size_t)h1p * src_w + w1p];
        float4 v;
#define PP_BILERP(f) v.f = __fadd_rn(__fmul_rn(h0l, __fadd_rn(__fmul_rn(w0l, p00.f), __fmul_rn(w1l, p01.f))), \
                                     __fmul_rn(h1l, __fadd_rn(__fmul_rn(w0l, p10.f), __fmul_rn(w1l, p11.f))))
        PP_BILERP(x); PP_BILERP(y); PP_BILERP(z); PP_BILERP(w);
#undef PP_BILERP
        out[(size_t)r * out_h * out_w + i] = v;
    }
}

}  // namespace

int pp_launch_avgpool(const float* x, float* y, int n, int hin, int win, int c, int kh, int kw, int stride, hipStream_t stream) {
    PP_REQUIRE(n > 0 && c > 0 && (c & 3) == 0 && kh > 0 && kw > 0 && stride > 0 && hin >= kh && win >= kw,
               "avgpool: bad dims (c = %d must be a multiple of 4, window %dx%d inside %dx%d)", c, kh, kw, hin, win);
    const int ho = (hin - kh) / stride + 1, wo = (win - kw) / stride + 1;
    const size_t total = (size_t)n * ho * wo * (c / 4);
    hipLaunchKernelGGL(avgpool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, reinterpret_cast<const float4*>(x),
                       reinterpret_cast<float4*>(y), total, hin, win, ho, wo, c / 4, kh, kw, stride);
    PP_HIP_CHECK(hipGetLastError());
    return PP_OK;
}

extern "C" int pp_crop_resize_bilinear(pp_ctx* ctx, const float* src_nhwc4, int n_src, int src_h, int src_w, const int32_t* rects5,
                                       int n, int out_h, int out_w, float* out_nhwc4) {
    PP_REQUIRE(ctx && src_nhwc4 && out_nhwc4 && (n == 0 || rects5), "pp_crop_resize_bilinear: NULL argument");
    PP_REQUIRE(n >= 0 && n_src > 0 && src_h > 0 && src_w > 0 && out_h > 0 && out_w > 0, "pp_crop_resize_bilinear: bad dims");
    if (n == 0) return PP_OK;
    for (int i = 0; i < n; ++i) {
        const int32_t* q = rects5 + 5 * i;
        PP_REQUIRE(q[0] >= 0 && q[0] < n_src && q[1] >= 0 && q[2] >= 0 && q[3] > q[1] && q[4] > q[2] && q[3] <= src_w && q[4] <= src_h,
                   "pp_crop_resize_bilinear: rect %d = (frame %d, %d, %d, %d, %d) outside %d x %dx%d", i, q[0], q[1], q[2], q[3], q[4],
                   n_src, src_h, src_w);
    }
    const size_t bytes = (size_t)n * sizeof(CropRect);
    int rc = ctx->ensure_scratch(bytes);
    if (rc != PP_OK) return rc;
    hipStream_t s = ctx->stream;
    PP_HIP_CHECK(hipMemcpyAsync(ctx->scratch, rects5, bytes, hipMemcpyHostToDevice, s));
    PP_HIP_CHECK(hipStreamSynchronize(s));      // rects5 is the caller's pageable memory
    const int bx = std::min((out_h * out_w + 255) / 256, 64);
    hipLaunchKernelGGL(crop_resize_kernel, dim3(bx, n), dim3(256), 0, s, reinterpret_cast<const float4*>(src_nhwc4), src_h, src_w,
                       static_cast<const CropRect*>(ctx->scratch), out_h, out_w, reinterpret_cast<float4*>(out_nhwc4));
    PP_HIP_CHECK(hipGetLastError());
    return PP_OK;
}

int pp_launch_upsample_add(const float* t, const float* res1, const float* res2, float* y, int n, int H, int W, int c,
                           int up_log2, int relu, hipStream_t stream, const float* t2, int up2, const float* t3, int up3,
                           unsigned* y_amax) {
    PP_REQUIRE(n > 0 && H > 0 && W > 0 && c > 0 && (c & 3) == 0, "upsample_add: c = %d must be a multiple of 4", c);
    PP_REQUIRE(up_log2 >= 0 && up_log2 <= 5 && (H >> up_log2 << up_log2) == H && (W >> up_log2 << up_log2) == W,
               "upsample_add: %dx%d is not a multiple of 2^%d", H, W, up_log2);
    const size_t total = (size_t)n * H * W * (c / 4);
    hipLaunchKernelGGL(upsample_add_kernel, dim3((unsigned)((total + 256 * UA_R - 1) / (256 * UA_R))), dim3(256), 0, stream,
                       reinterpret_cast<const float4*>(t), reinterpret_cast<const float4*>(res1),
                       reinterpret_cast<const float4*>(res2), reinterpret_cast<float4*>(y), total, H, W, c / 4, up_log2, relu,
                       reinterpret_cast<const float4*>(t2), up2, reinterpret_cast<const float4*>(t3), up3, y_amax);
    PP_HIP_CHECK(hipGetLastError());
    return PP_OK;
}
