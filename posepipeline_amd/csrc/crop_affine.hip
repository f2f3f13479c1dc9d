// bbox-conditioned affine crop + normalise, fused:  u8 HWC frame -> fp32 NHWC(4) network input.
//
// Replaces, for the top-down 2D stage reached from pose_pipeline/wrappers/mmpose.py:75,
//   mmpose `_box2cs`                      (bbox -> centre/scale, aspect fix, /200, x1.25)
//   mmpose `get_affine_transform` + cv2.getAffineTransform (3-point similarity, solved in double)
//   cv2.warpAffine(INTER_LINEAR, BORDER_CONSTANT 0)  -- OpenCV's fixed-point bilinear:
//        AB_BITS=10 coordinates, INTER_BITS=5 (1/32 px) fractions, 15-bit weights
//   ToTensor (/255) + NormalizeTensor(mean, std)   (hrnet_w48_coco_384x288_dark.py:129-136)
//   `img.flip(3)` of flip_test (test_cfg :81-85): the mirrored sample is written by the same pass.
// In-tree twin of the crop idiom: pose_pipeline/utils/bounding_box.py:32-53.
// The integer part (the u8 crop) is bit-exact against oracle/preprocess.py; the normalisation is
// a 3x256 fp32 table computed by the caller in fp32, so the float tensor is bit-exact as well.
//
// HBM-bound: one thread per output pixel reads 4 taps x 3 bytes (neighbouring threads hit
// neighbouring texels, served by L1/L2) and writes one (two with flip) 16-byte NHWC pixel.
#include <cmath>

#include "pp_internal.h"

namespace {

__device__ __forceinline__ int sat_round_i32(double v) {
    // cv::saturate_cast<int>(double) == cvRound: round-half-to-even, saturating
    double r = rint(v);
    if (r > 2147483647.0) return 2147483647;
    if (r < -2147483648.0) return (int)0x80000000;
    return (int)r;
}

__device__ __forceinline__ int sat_short(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

__global__ __launch_bounds__(256) void crop_affine_kernel(const uint8_t* __restrict__ frames, int H, int W,
                                                          const PersonXform* __restrict__ xf, int n_person,
                                                          int out_w, int out_h, const float* __restrict__ lut,
                                                          int cm0, int cm1, int cm2, int flip,
                                                          float* __restrict__ out, uint8_t* __restrict__ crop_u8) {
    __shared__ float s_lut[3 * 256];
    for (int i = threadIdx.x; i < 768; i += blockDim.x) s_lut[i] = lut[i];
    __syncthreads();
    const int person = blockIdx.y;
    const PersonXform t = xf[person];
    const int npix = out_w * out_h;
    const size_t sample = (size_t)npix * 4;
    float* o0 = out + (size_t)person * sample;
    float* o1 = out + (size_t)(n_person + person) * sample;
    const uint8_t* img = frames + (size_t)t.frame * H * W * 3;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += gridDim.x * blockDim.x) {
        const int y = p / out_w;
        const int x = p - y * out_w;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        uint8_t c3[3] = {0, 0, 0};
        if (t.valid) {
            // no fused multiply-add here: OpenCV evaluates (M*x)*1024 and (M*y + b)*1024 in plain doubles
            const int adelta = sat_round_i32(__dmul_rn(__dmul_rn(t.a00, (double)x), 1024.0));
            const int bdelta = sat_round_i32(__dmul_rn(__dmul_rn(t.a10, (double)x), 1024.0));
            const int X0 = sat_round_i32(__dmul_rn(__dadd_rn(__dmul_rn(t.a01, (double)y), t.b0), 1024.0)) + 16;
            const int Y0 = sat_round_i32(__dmul_rn(__dadd_rn(__dmul_rn(t.a11, (double)y), t.b1), 1024.0)) + 16;
            const int X = (X0 + adelta) >> 5;
            const int Y = (Y0 + bdelta) >> 5;
            const int sx = sat_short(X >> 5), sy = sat_short(Y >> 5);
            const int fx = X & 31, fy = Y & 31;
            const int w00 = (32 - fx) * (32 - fy), w01 = fx * (32 - fy), w10 = (32 - fx) * fy, w11 = fx * fy;
            const bool x0in = (unsigned)sx < (unsigned)W, x1in = (unsigned)(sx + 1) < (unsigned)W;
            const bool y0in = (unsigned)sy < (unsigned)H, y1in = (unsigned)(sy + 1) < (unsigned)H;
            const uint8_t* r0 = img + ((size_t)sy * W + sx) * 3;
            const uint8_t* r1 = r0 + (size_t)W * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int p00 = (y0in && x0in) ? r0[c] : 0;
                const int p01 = (y0in && x1in) ? r0[3 + c] : 0;
                const int p10 = (y1in && x0in) ? r1[c] : 0;
                const int p11 = (y1in && x1in) ? r1[3 + c] : 0;
                // weights are (..)*(..)*32 of 32768: (sum*32 + 16384) >> 15
                c3[c] = (uint8_t)(((p00 * w00 + p01 * w01 + p10 * w10 + p11 * w11) * 32 + 16384) >> 15);
            }
            v.x = s_lut[0 * 256 + c3[cm0]];
            v.y = s_lut[1 * 256 + c3[cm1]];
            v.z = s_lut[2 * 256 + c3[cm2]];
        }
        *reinterpret_cast<float4*>(o0 + (size_t)p * 4) = v;
        if (flip) *reinterpret_cast<float4*>(o1 + ((size_t)y * out_w + (out_w - 1 - x)) * 4) = v;
        if (crop_u8) {
            uint8_t* cp = crop_u8 + ((size_t)person * npix + p) * 3;
            cp[0] = c3[0]; cp[1] = c3[1]; cp[2] = c3[2];
        }
    }
}

__global__ __launch_bounds__(256) void flip_w_kernel(const float4* __restrict__ src, float4* __restrict__ dst,
                                                     size_t rows, int w) {
    const size_t total = rows * w;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / w;
        const int x = (int)(i - r * w);
        dst[r * w + (w - 1 - x)] = src[i];
    }
}

// cv::solve(DECOMP_LU) restated for the 6x6 system of cv2.getAffineTransform: Gaussian elimination
// with partial pivoting in double.
bool solve6(double A[6][6], double b[6], double x[6]) {
    const int n = 6;
    for (int i = 0; i < n; ++i) {
        int k = i;
        for (int j = i + 1; j < n; ++j)
            if (std::fabs(A[j][i]) > std::fabs(A[k][i])) k = j;
        if (std::fabs(A[k][i]) < 2.220446049250313e-16 * 100) return false;
        if (k != i) {
            for (int j = i; j < n; ++j) std::swap(A[i][j], A[k][j]);
            std::swap(b[i], b[k]);
        }
        const double d = -1.0 / A[i][i];
        for (int j = i + 1; j < n; ++j) {
            const double alpha = A[j][i] * d;
            for (int c = i + 1; c < n; ++c) A[j][c] += alpha * A[i][c];
            b[j] += alpha * b[i];
        }
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = b[i];
        for (int c = i + 1; c < n; ++c) s -= A[i][c] * x[c];
        x[i] = s / A[i][i];
    }
    return true;
}

}  // namespace

// mmpose `_box2cs` + `get_affine_transform(rot=0)` + OpenCV's inversion.  Returns false for NaN boxes.
bool pp_person_transform(const double* bb, int out_w, int out_h, float cs[4], PersonXform* t, int udp) {
    t->valid = 0;
    cs[0] = cs[1] = cs[2] = cs[3] = 0.f;
    if (std::isnan(bb[0]) || std::isnan(bb[1]) || std::isnan(bb[2]) || std::isnan(bb[3])) return false;
    double x = bb[0], y = bb[1], w = bb[2], h = bb[3];
    const double aspect = (double)out_w / (double)out_h;
    const float cx = (float)(x + w * 0.5), cy = (float)(y + h * 0.5);
    if (w > aspect * h) h = w * 1.0 / aspect;
    else if (w < aspect * h) w = h * aspect;
    float sx = (float)(w / 200.0), sy = (float)(h / 200.0);
    sx = sx * 1.25f;   // float32 array * python float stays float32
    sy = sy * 1.25f;
    cs[0] = cx; cs[1] = cy; cs[2] = sx; cs[3] = sy;
    double M[6];
    if (udp) {
        // mmpose TopDownAffine(use_udp=True): get_warp_matrix(theta = 0, size_input = c * 2.0, size_dst = image_size - 1.0,
        // size_target = s * 200.0): scalars in float64, the 2x3 matrix itself is float32
        const float in_x = cx * 2.0f, in_y = cy * 2.0f, tg_x = sx * 200.0f, tg_y = sy * 200.0f;
        const double scale_x = ((double)out_w - 1.0) / (double)tg_x, scale_y = ((double)out_h - 1.0) / (double)tg_y;
        M[0] = (double)(float)(1.0 * scale_x);
        M[1] = (double)(float)(-0.0 * scale_x);
        M[2] = (double)(float)(scale_x * (-0.5 * (double)in_x * 1.0 + 0.5 * (double)in_y * 0.0 + 0.5 * (double)tg_x));
        M[3] = (double)(float)(0.0 * scale_y);
        M[4] = (double)(float)(1.0 * scale_y);
        M[5] = (double)(float)(scale_y * (-0.5 * (double)in_x * 0.0 - 0.5 * (double)in_y * 1.0 + 0.5 * (double)tg_y));
        if (!(std::isfinite(M[0]) && std::isfinite(M[4]) && std::isfinite(M[2]) && std::isfinite(M[5])))
            for (double& m : M) m = 0.0;   // zero-size box
    } else {
    // get_affine_transform: points are float32, the intermediate sums float64
    const float src_w = sx * 200.0f;
    const float s0x = cx, s0y = cy;
    const float s1x = (float)((double)cx + 0.0);
    const float s1y = (float)((double)cy + (double)src_w * -0.5);
    const float d0 = s0x - s1x, d1 = s0y - s1y;            // direction = a - b (float32)
    const float s2x = s1x + (-d1), s2y = s1y + d0;
    const float dst_w = (float)out_w, dst_h = (float)out_h;
    const float t0x = (float)(dst_w * 0.5), t0y = (float)(dst_h * 0.5);
    const float t1x = t0x, t1y = (float)((double)t0y + (double)dst_w * -0.5);
    const float e0 = t0x - t1x, e1 = t0y - t1y;
    const float t2x = t1x + (-e1), t2y = t1y + e0;
    const float sp[3][2] = {{s0x, s0y}, {s1x, s1y}, {s2x, s2y}};
    const float dp[3][2] = {{t0x, t0y}, {t1x, t1y}, {t2x, t2y}};
    double A[6][6] = {}, b[6];
    for (int i = 0; i < 3; ++i) {
        A[2 * i][0] = sp[i][0]; A[2 * i][1] = sp[i][1]; A[2 * i][2] = 1.0;
        A[2 * i + 1][3] = sp[i][0]; A[2 * i + 1][4] = sp[i][1]; A[2 * i + 1][5] = 1.0;
        b[2 * i] = dp[i][0]; b[2 * i + 1] = dp[i][1];
    }
    if (!solve6(A, b, M)) {
        // singular (zero-size box): OpenCV returns zeros -> D == 0 -> all-zero inverse map
        for (double& m : M) m = 0.0;
    }
    }
    double D = M[0] * M[4] - M[1] * M[3];
    D = D != 0 ? 1.0 / D : 0;
    const double A11 = M[4] * D, A22 = M[0] * D;
    M[0] = A11; M[1] *= -D; M[3] *= -D; M[4] = A22;
    const double b1 = -M[0] * M[2] - M[1] * M[5];
    const double b2 = -M[3] * M[2] - M[4] * M[5];
    t->a00 = M[0]; t->a01 = M[1]; t->b0 = b1;
    t->a10 = M[3]; t->a11 = M[4]; t->b1 = b2;
    t->valid = 1;
    return true;
}

int pp_enqueue_crop(hipStream_t s, const uint8_t* frames, int h, int w, const PersonXform* xf, int n_person,
                    int out_w, int out_h, const float* lut, const int32_t chan_map[3], int flip, float* out,
                    uint8_t* crop_u8) {
    if (n_person <= 0) return PP_OK;
    const int npix = out_w * out_h;
    dim3 grid(std::min((npix + 255) / 256, 64), n_person);
    hipLaunchKernelGGL(crop_affine_kernel, grid, dim3(256), 0, s, frames, h, w, xf, n_person, out_w, out_h, lut,
                       chan_map[0], chan_map[1], chan_map[2], flip, out, crop_u8);
    PP_HIP_CHECK(hipGetLastError());
    return PP_OK;
}

int pp_enqueue_flip_w(hipStream_t s, const float* src, float* dst, int n, int h, int w) {
    const size_t rows = (size_t)n * h;
    if (rows == 0) return PP_OK;
    const int blocks = (int)std::min<size_t>((rows * w + 255) / 256, 2048);
    hipLaunchKernelGGL(flip_w_kernel, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const float4*>(src),
                       reinterpret_cast<float4*>(dst), rows, w);
    PP_HIP_CHECK(hipGetLastError());
    return PP_OK;
}

extern "C" int pp_crop_affine_normalize(pp_ctx* ctx, const uint8_t* frames, int n_frames, int h, int w,
                                        const int32_t* frame_idx, const double* bbox_tlwh, int n_person,
                                        int out_w, int out_h, const float* lut, const int32_t* chan_map,
                                        int flip, float* out, float* center_scale, uint8_t* crop_u8,
                                        int32_t* valid, int mem) {
    PP_REQUIRE(ctx && frames && frame_idx && bbox_tlwh && lut && chan_map && out, "pp_crop_affine_normalize: NULL argument");
    PP_REQUIRE(n_frames > 0 && h > 0 && w > 0 && out_w > 0 && out_h > 0, "pp_crop_affine_normalize: empty dims");
    if (n_person <= 0) return PP_OK;
    for (int c = 0; c < 3; ++c) PP_REQUIRE(chan_map[c] >= 0 && chan_map[c] < 3, "chan_map[%d] out of range", c);
    std::vector<PersonXform> xf(n_person);
    const int udp = (flip >> 1) & 1;   // bit 1 of `flip`: UDP transform
    flip &= 1;
    for (int i = 0; i < n_person; ++i) {
        PP_REQUIRE(frame_idx[i] >= 0 && frame_idx[i] < n_frames, "frame_idx[%d]=%d out of range", i, frame_idx[i]);
        float cs[4];
        pp_person_transform(bbox_tlwh + 4 * i, out_w, out_h, cs, &xf[i], udp);
        xf[i].frame = frame_idx[i];
        if (center_scale) memcpy(center_scale + 4 * i, cs, sizeof(cs));
        if (valid) valid[i] = xf[i].valid;
    }
    const size_t frames_b = (size_t)n_frames * h * w * 3;
    const size_t n_out = (size_t)n_person * (flip ? 2 : 1);
    const size_t out_e = n_out * out_w * out_h * 4;
    const size_t crop_b = crop_u8 ? (size_t)n_person * out_w * out_h * 3 : 0;
    size_t need = ScratchCursor::align(n_person * sizeof(PersonXform)) + ScratchCursor::align(768 * sizeof(float));
    if (mem == PP_MEM_HOST) need += ScratchCursor::align(frames_b) + ScratchCursor::align(out_e * 4) + ScratchCursor::align(crop_b);
    int rc = ctx->ensure_scratch(need);
    if (rc != PP_OK) return rc;
    ScratchCursor cur(ctx);
    hipStream_t s = ctx->stream;
    PersonXform* dxf = cur.take<PersonXform>(n_person);
    float* dlut = cur.take<float>(768);
    // pageable host memory: the copy engine reads the source before the call returns
    PP_HIP_CHECK(hipMemcpyAsync(dxf, xf.data(), n_person * sizeof(PersonXform), hipMemcpyHostToDevice, s));
    PP_HIP_CHECK(hipMemcpyAsync(dlut, lut, 768 * sizeof(float), hipMemcpyHostToDevice, s));
    const uint8_t* dframes = frames;
    float* dout = out;
    uint8_t* dcrop = crop_u8;
    if (mem == PP_MEM_HOST) {
        uint8_t* df = cur.take<uint8_t>(frames_b);
        PP_HIP_CHECK(hipMemcpyAsync(df, frames, frames_b, hipMemcpyHostToDevice, s));
        dframes = df;
        dout = cur.take<float>(out_e);
        if (crop_u8) dcrop = cur.take<uint8_t>(crop_b);
    }
    rc = pp_enqueue_crop(s, dframes, h, w, dxf, n_person, out_w, out_h, dlut, chan_map, flip, dout, dcrop);
    if (rc != PP_OK) return rc;
    // the staged transforms / LUT live in ctx scratch: finish before another call can reuse it
    if (mem == PP_MEM_HOST) {
        PP_HIP_CHECK(hipMemcpyAsync(out, dout, out_e * 4, hipMemcpyDeviceToHost, s));
        if (crop_u8) PP_HIP_CHECK(hipMemcpyAsync(crop_u8, dcrop, crop_b, hipMemcpyDeviceToHost, s));
    }
    PP_HIP_CHECK(hipStreamSynchronize(s));
    return PP_OK;
}
