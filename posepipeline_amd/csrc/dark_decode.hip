// flip-merge + heatmap decode (argmax, DARK "unbiased" Taylor refinement or the +-0.25 px
// "default" shift, back-map to image pixels), one workgroup per (person, joint) heatmap.
//
// Replaces the CPU/numpy tail of mmpose's TopDown.forward_test reached from
// pose_pipeline/wrappers/mmpose.py:75:
//   head.inference_model: flip_back (swap left/right channels, reverse W), shift_heatmap
//       (hm[..., 1:] = hm[..., :-1]) and (hm + hm_flipped) * 0.5       (test_cfg ...dark.py:81-85)
//   keypoints_from_heatmaps(post_process='unbiased', kernel=17): _get_max_preds, _gaussian_blur
//       (zero-pad 8, cv2.GaussianBlur(17x17, sigma 2.9), rescale to the original max),
//       log(max(.,1e-10)), _taylor, transform_preds.
// The only in-tree statement of that maths is pose_pipeline/utils/inference.py:27-114 (float64,
// zero instead of -1 for empty maps); this kernel follows the mmpose float32 variant that the
// wrapper actually runs -- see oracle/decode.py for the line-by-line restatement.
//
// The heatmap lives in LDS for the whole decode (96x72 fp32 = 27.6 KB, two planes for the
// separable blur); argmax / max are wave-shuffle reductions.  HBM traffic = the two heatmap
// reads (2 * K*H*W*4 bytes per person) + 12 bytes per joint.
#include "pp_internal.h"

namespace {

constexpr int MAX_BLUR = 33;

struct DecodeArgs {
    const float* hm;
    const float* hm_flip;      // may be null
    const int32_t* flip_perm;  // may be null when hm_flip is null
    const float* center_scale; // [n][4]
    float* kpts;               // [n][k][3]
    float* merged;             // optional [n][k][h][w]
    int n, k, h, w;
    int shift_heatmap, post, blur_kernel;
    float gk[MAX_BLUR];        // Gaussian taps (float32, normalised in double)
};

// (value, index) max with first-occurrence tie-break, i.e. numpy argmax semantics
__device__ __forceinline__ void argmax_combine(float& v, int& i, float ov, int oi) {
    if (ov > v || (ov == v && oi < i)) {
        v = ov;
        i = oi;
    }
}

__device__ void block_argmax(float& v, int& i, float* s_v, int* s_i) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_down(v, off, 64);
        const int oi = __shfl_down(i, off, 64);
        argmax_combine(v, i, ov, oi);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) {
        s_v[wave] = v;
        s_i[wave] = i;
    }
    __syncthreads();
    v = s_v[0];
    i = s_i[0];
    for (int wv = 1; wv < (int)(blockDim.x >> 6); ++wv) argmax_combine(v, i, s_v[wv], s_i[wv]);
}

__global__ __launch_bounds__(256) void flip_merge_decode_kernel(DecodeArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int HW = a.h * a.w;
    float* A = smem;        // merged heatmap, later the blurred map
    float* B = smem + HW;   // row-pass intermediate
    __shared__ float s_v[4];
    __shared__ int s_i[4];

    const int k = blockIdx.x % a.k;
    const int n = blockIdx.x / a.k;
    const float* src = a.hm + ((size_t)n * a.k + k) * HW;
    const float* fsrc = a.hm_flip ? a.hm_flip + ((size_t)n * a.k + a.flip_perm[k]) * HW : nullptr;
    float* mdst = a.merged ? a.merged + ((size_t)n * a.k + k) * HW : nullptr;

    // ---- 1. flip-merge into LDS + argmax -----------------------------------------------------
    float best = -INFINITY;
    int besti = 0x7fffffff;
    for (int i = threadIdx.x; i < HW; i += blockDim.x) {
        float v = src[i];
        if (fsrc) {
            const int y = i / a.w, x = i - y * a.w;
            // flipped back: column W-1-x'; shifted right by one: x' = max(x-1, 0)
            const int xs = a.shift_heatmap ? (x > 0 ? x - 1 : 0) : x;
            const float fv = fsrc[y * a.w + (a.w - 1 - xs)];
            v = __fmul_rn(__fadd_rn(v, fv), 0.5f);
        }
        A[i] = v;
        if (mdst) mdst[i] = v;
        argmax_combine(best, besti, v, i);
    }
    block_argmax(best, besti, s_v, s_i);
    const float maxval = best;
    const int idx = besti;
    int px = idx % a.w, py = idx / a.w;
    const bool has_peak = maxval > 0.0f;
    // mmpose: preds = -1 where maxval <= 0
    float cxp = has_peak ? (float)px : -1.0f;
    float cyp = has_peak ? (float)py : -1.0f;
    if (!has_peak) {
        px = -1;
        py = -1;
    }

    if (a.post == 1 || a.post == 2) {
        // ---- 2. separable Gaussian blur (row pass k = 0..ks-1 sequential, column pass centre + symmetric pairs),
        //         float32 without contraction.  post 1: mmpose pads with zeros before cv2.GaussianBlur; post 2
        //         (post_dark_udp) calls cv2.GaussianBlur on the map itself: default border BORDER_REFLECT_101 ------
        const int ks = a.blur_kernel, r = ks >> 1;
        const bool reflect = a.post == 2;
        auto mirror = [](int p, int n) { p = p < 0 ? -p : p; p = p >= n ? 2 * (n - 1) - p : p; return min(max(p, 0), n - 1); };
        __syncthreads();
        for (int i = threadIdx.x; i < HW; i += blockDim.x) {
            const int y = i / a.w, x = i - y * a.w;
            const float* row = A + y * a.w;
            float s = 0.f;
            bool first = true;
            for (int j = 0; j < ks; ++j) {
                const int xx = x + j - r;
                const float sv = ((unsigned)xx < (unsigned)a.w) ? row[xx] : (reflect ? row[mirror(xx, a.w)] : 0.f);
                const float t = __fmul_rn(a.gk[j], sv);
                s = first ? t : __fadd_rn(s, t);
                first = false;
            }
            B[i] = s;
        }
        __syncthreads();
        float bmax = -INFINITY;
        int bidx = 0x7fffffff;
        for (int i = threadIdx.x; i < HW; i += blockDim.x) {
            const int y = i / a.w, x = i - y * a.w;
            float s = __fmul_rn(a.gk[r], B[i]);
            for (int j = 1; j <= r; ++j) {
                const float up = (y - j >= 0) ? B[i - j * a.w] : (reflect ? B[mirror(y - j, a.h) * a.w + x] : 0.f);
                const float dn = (y + j < a.h) ? B[i + j * a.w] : (reflect ? B[mirror(y + j, a.h) * a.w + x] : 0.f);
                s = __fadd_rn(s, __fmul_rn(a.gk[r + j], __fadd_rn(dn, up)));
            }
            A[i] = s;
            argmax_combine(bmax, bidx, s, i);
        }
        block_argmax(bmax, bidx, s_v, s_i);   // also orders the A[] writes before the reads below
        if (a.post == 2) {
            // ---- post_dark_udp: clip to [0.001, 50], log, edge-replicated 3x3 stencil, Newton step with the
            //      float64 inverse of (Hessian + float32 eps * I); applied wherever there is a peak ----------------
            if (threadIdx.x == 0 && has_peak) {
                auto L = [&](int yy, int xx) -> float {
                    yy = min(max(yy, 0), a.h - 1);
                    xx = min(max(xx, 0), a.w - 1);
                    const float v = fminf(fmaxf(A[yy * a.w + xx], 0.001f), 50.0f);
                    return (float)log((double)v);
                };
                const float i_ = L(py, px), ix1 = L(py, px + 1), iy1 = L(py + 1, px), ix1y1 = L(py + 1, px + 1);
                const float ix1_y1_ = L(py - 1, px - 1), ix1_ = L(py, px - 1), iy1_ = L(py - 1, px);
                const float dx = __fmul_rn(0.5f, __fsub_rn(ix1, ix1_)), dy = __fmul_rn(0.5f, __fsub_rn(iy1, iy1_));
                const float dxx = __fadd_rn(__fsub_rn(ix1, __fmul_rn(2.0f, i_)), ix1_);
                const float dyy = __fadd_rn(__fsub_rn(iy1, __fmul_rn(2.0f, i_)), iy1_);
                float t = __fsub_rn(ix1y1, ix1);
                t = __fsub_rn(t, iy1); t = __fadd_rn(t, i_); t = __fadd_rn(t, i_);
                t = __fsub_rn(t, ix1_); t = __fsub_rn(t, iy1_); t = __fadd_rn(t, ix1_y1_);
                const float dxy = __fmul_rn(0.5f, t);
                const double eps = 1.1920928955078125e-07;
                const double h00 = (double)dxx + eps, h01 = (double)dxy, h11 = (double)dyy + eps;
                const double det = h00 * h11 - h01 * h01;
                const double ox = (h11 * (double)dx - h01 * (double)dy) / det;
                const double oy = (h00 * (double)dy - h01 * (double)dx) / det;
                cxp = (float)((double)cxp - ox);
                cyp = (float)((double)cyp - oy);
            }
        } else
        // ---- 3. rescale, clamp, log at the Taylor stencil; 4. Taylor step in float64 ------------
        if (threadIdx.x == 0 && 1 < px && px < a.w - 2 && 1 < py && py < a.h - 2) {
            const float scale = maxval / bmax;     // float32 / float32
            auto L = [&](int yy, int xx) -> float {
                float v = __fmul_rn(A[yy * a.w + xx], scale);
                v = fmaxf(v, 1e-10f);
                return (float)log((double)v);      // correctly rounded float32 log
            };
            const float l00 = L(py, px);
            const float lxp = L(py, px + 1), lxm = L(py, px - 1), lyp = L(py + 1, px), lym = L(py - 1, px);
            const float lxpp = L(py, px + 2), lxmm = L(py, px - 2), lypp = L(py + 2, px), lymm = L(py - 2, px);
            const float lpp = L(py + 1, px + 1), lmp = L(py - 1, px + 1), lpm = L(py + 1, px - 1), lmm = L(py - 1, px - 1);
            // numpy scalar arithmetic (legacy promotion): f32 - f32 stays f32, python-float factors promote to f64
            const double dx = 0.5 * (double)__fsub_rn(lxp, lxm);
            const double dy = 0.5 * (double)__fsub_rn(lyp, lym);
            const double dxx = 0.25 * (((double)lxpp - 2.0 * (double)l00) + (double)lxmm);
            const double dyy = 0.25 * (((double)lypp - 2.0 * (double)l00) + (double)lymm);
            const double dxy = 0.25 * (double)__fadd_rn(__fsub_rn(__fsub_rn(lpp, lmp), lpm), lmm);
            const double det = dxx * dyy - dxy * dxy;
            if (det != 0.0) {
                // offset = -inv([[dxx,dxy],[dxy,dyy]]) @ [dx,dy]
                const double i00 = dyy / det, i01 = -dxy / det, i11 = dxx / det;
                const double ox = -(i00 * dx + i01 * dy);
                const double oy = -(i01 * dx + i11 * dy);
                cxp = (float)((double)cxp + ox);
                cyp = (float)((double)cyp + oy);
            }
        }
    } else if (a.post == 0) {
        __syncthreads();
        if (threadIdx.x == 0 && 1 < px && px < a.w - 1 && 1 < py && py < a.h - 1) {
            const float ddx = __fsub_rn(A[py * a.w + px + 1], A[py * a.w + px - 1]);
            const float ddy = __fsub_rn(A[(py + 1) * a.w + px], A[(py - 1) * a.w + px]);
            const float sx = ddx > 0.f ? 1.f : (ddx < 0.f ? -1.f : 0.f);
            const float sy = ddy > 0.f ? 1.f : (ddy < 0.f ? -1.f : 0.f);
            cxp = __fadd_rn(cxp, sx * 0.25f);
            cyp = __fadd_rn(cyp, sy * 0.25f);
        }
    }

    // ---- 5. transform_preds (float32 array arithmetic with float64 scalars cast to float32) -----
    if (threadIdx.x == 0) {
        const float* cs = a.center_scale + 4 * n;
        const float s200x = __fmul_rn(cs[2], 200.0f), s200y = __fmul_rn(cs[3], 200.0f);
        // use_udp (post 2): the grid spans output_size - 1
        const float scale_x = (float)((double)s200x / (a.post == 2 ? (double)a.w - 1.0 : (double)a.w));
        const float scale_y = (float)((double)s200y / (a.post == 2 ? (double)a.h - 1.0 : (double)a.h));
        const float hx = (float)((double)s200x * 0.5), hy = (float)((double)s200y * 0.5);
        float* o = a.kpts + ((size_t)n * a.k + k) * 3;
        o[0] = __fsub_rn(__fadd_rn(__fmul_rn(cxp, scale_x), cs[0]), hx);
        o[1] = __fsub_rn(__fadd_rn(__fmul_rn(cyp, scale_y), cs[1]), hy);
        o[2] = maxval;
    }
}

}  // namespace

static void fill_gaussian_taps(DecodeArgs& a) {
    // cv::getGaussianKernel(ksize, sigma<=0): sigma = 0.3*((ksize-1)*0.5 - 1) + 0.8, taps in double,
    // normalised to sum 1, stored as float32
    const int ks = a.blur_kernel;
    const double sigma = 0.3 * ((ks - 1) * 0.5 - 1) + 0.8;
    const double scale2x = -0.5 / (sigma * sigma);
    double tmp[MAX_BLUR], sum = 0;
    for (int i = 0; i < ks; ++i) {
        const double x = i - (ks - 1) * 0.5;
        tmp[i] = std::exp(scale2x * x * x);
        sum += tmp[i];
    }
    sum = 1.0 / sum;
    for (int i = 0; i < ks; ++i) a.gk[i] = (float)(tmp[i] * sum);
}

int pp_enqueue_decode(hipStream_t s, const DecodeParams& p, const float* hm, const float* hm_flip,
                      const int32_t* flip_perm, const float* center_scale, float* kpts, float* merged) {
    PP_REQUIRE(p.n >= 0 && p.k > 0 && p.h > 0 && p.w > 0, "decode: bad dims");
    PP_REQUIRE(!hm_flip || flip_perm, "decode: hm_flip given without flip_perm");
    PP_REQUIRE(p.post >= -1 && p.post <= 2, "decode: post must be -1 (none), 0 (default), 1 (unbiased) or 2 (UDP)");
    PP_REQUIRE(p.post < 1 || ((p.blur_kernel & 1) && p.blur_kernel >= 3 && p.blur_kernel <= MAX_BLUR),
               "decode: blur_kernel must be odd in [3,%d]", MAX_BLUR);
    const size_t lds = (size_t)2 * p.h * p.w * sizeof(float);
    PP_REQUIRE(lds <= 160 * 1024 - 64, "decode: heatmap %dx%d does not fit LDS", p.h, p.w);
    if (p.n == 0) return PP_OK;
    DecodeArgs a{};
    a.n = p.n; a.k = p.k; a.h = p.h; a.w = p.w;
    a.shift_heatmap = p.shift_heatmap; a.post = p.post; a.blur_kernel = p.blur_kernel;
    if (p.post >= 1) fill_gaussian_taps(a);
    a.hm = hm; a.hm_flip = hm_flip; a.flip_perm = flip_perm; a.center_scale = center_scale;
    a.kpts = kpts; a.merged = merged;
    static bool attr_set = false;
    if (!attr_set && lds > 64 * 1024) {
        PP_HIP_CHECK(hipFuncSetAttribute((const void*)flip_merge_decode_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
        attr_set = true;
    }
    hipLaunchKernelGGL(flip_merge_decode_kernel, dim3(p.n * p.k), dim3(256), lds, s, a);
    PP_HIP_CHECK(hipGetLastError());
    return PP_OK;
}

extern "C" int pp_flip_merge_decode(pp_ctx* ctx, const float* hm, const float* hm_flip, int n, int k, int h, int w,
                                    const int32_t* flip_perm, int shift_heatmap, int post, int blur_kernel,
                                    const float* center_scale, float* kpts, float* merged, int mem) {
    PP_REQUIRE(ctx && hm && center_scale && kpts, "pp_flip_merge_decode: NULL argument");
    PP_REQUIRE(n >= 0 && k > 0 && h > 0 && w > 0, "pp_flip_merge_decode: bad dims");
    PP_REQUIRE(!hm_flip || flip_perm, "pp_flip_merge_decode: hm_flip given without flip_perm");
    if (n == 0) return PP_OK;
    const DecodeParams dp{n, k, h, w, shift_heatmap, post, blur_kernel};
    const size_t hm_e = (size_t)n * k * h * w;
    hipStream_t s = ctx->stream;
    size_t need = ScratchCursor::align(k * sizeof(int32_t));
    if (mem == PP_MEM_HOST) need += 3 * ScratchCursor::align(hm_e * 4) + ScratchCursor::align(n * 16) + ScratchCursor::align((size_t)n * k * 12);
    int rc = ctx->ensure_scratch(need);
    if (rc != PP_OK) return rc;
    ScratchCursor cur(ctx);
    int32_t* dperm = cur.take<int32_t>(k);
    if (flip_perm) {
        for (int i = 0; i < k; ++i) PP_REQUIRE(flip_perm[i] >= 0 && flip_perm[i] < k, "flip_perm[%d] out of range", i);
        PP_HIP_CHECK(hipMemcpyAsync(dperm, flip_perm, k * sizeof(int32_t), hipMemcpyHostToDevice, s));
    }
    const float *a_hm = hm, *a_hf = hm_flip, *a_cs = center_scale;
    float *dk = kpts, *dm = merged;
    if (mem == PP_MEM_HOST) {
        float* dh = cur.take<float>(hm_e);
        float* dhf = cur.take<float>(hm_e);
        dm = cur.take<float>(hm_e);
        float* dcs = cur.take<float>((size_t)n * 4);
        dk = cur.take<float>((size_t)n * k * 3);
        PP_HIP_CHECK(hipMemcpyAsync(dh, hm, hm_e * 4, hipMemcpyHostToDevice, s));
        if (hm_flip) PP_HIP_CHECK(hipMemcpyAsync(dhf, hm_flip, hm_e * 4, hipMemcpyHostToDevice, s));
        PP_HIP_CHECK(hipMemcpyAsync(dcs, center_scale, (size_t)n * 16, hipMemcpyHostToDevice, s));
        a_hm = dh; a_hf = hm_flip ? dhf : nullptr; a_cs = dcs;
        if (!merged) dm = nullptr;
    }
    rc = pp_enqueue_decode(s, dp, a_hm, a_hf, flip_perm ? dperm : nullptr, a_cs, dk, dm);
    if (rc != PP_OK) return rc;
    if (mem == PP_MEM_HOST) {
        PP_HIP_CHECK(hipMemcpyAsync(kpts, dk, (size_t)n * k * 12, hipMemcpyDeviceToHost, s));
        if (merged) PP_HIP_CHECK(hipMemcpyAsync(merged, dm, hm_e * 4, hipMemcpyDeviceToHost, s));
    }
    // flip_perm was staged in ctx scratch: finish before another call can reuse it
    PP_HIP_CHECK(hipStreamSynchronize(s));
    return PP_OK;
}
