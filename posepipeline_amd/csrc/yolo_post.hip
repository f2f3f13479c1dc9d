// Pre / post-processing kernels of the DeepSortYOLOv4 tracking method (pose_pipeline/wrappers/deep_sort_yolov4/):
//   pp_letterbox_bicubic  yolo4/utils.py:21-32 letterbox_image (PIL Image.resize BICUBIC = Pillow's 8-bit two-pass
//                         resampler, 22-bit fixed-point coefficients) + paste on the (128,128,128) canvas +
//                         yolo.py:93-95 float32 / 255, with the BGR->RGB swap of parser.py:55 folded in
//   pp_yolo_decode        yolo4/model.py:193-254 yolo_head + yolo_correct_boxes + box_confidence * class prob, one class
//   pp_reid_patches       tools/generate_detections.py:61-62 cv2.resize(INTER_LINEAR, 8-bit) of the clipped box to
//                         64x128 + the encoder graph's uint8 -> float cast and BGR -> RGB (freeze_model.py:239-255)
// Every float32 step follows oracle/yolo.py / oracle/reid.py operation by operation (sigmoid / exp evaluated in
// double and rounded once; -ffp-contract=off), the integer resamplers are bit-exact.
#include <algorithm>
#include <cmath>
#include <vector>

#include "pp_internal.h"

namespace {

constexpr int PIL_PRECISION_BITS = 32 - 8 - 2;

// horizontal pass: frames [n][H][W][3] u8 -> tmp [n][H][nw][3] u8;  tab: [nw][2 + kx] = (xmin, count, coefficients)
__global__ __launch_bounds__(256) void pil_horizontal_kernel(const uint8_t* __restrict__ frames, int H, int W, int nw, int kx,
                                                             const int32_t* __restrict__ tab, uint8_t* __restrict__ tmp) {
    const int f = blockIdx.z, y = blockIdx.y;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= nw) return;
    const int32_t* t = tab + (size_t)x * (2 + kx);
    const int xmin = t[0], cnt = t[1];
    const uint8_t* row = frames + ((size_t)f * H + y) * W * 3 + (size_t)xmin * 3;
    int s0 = 1 << (PIL_PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int k = 0; k < cnt; ++k) {
        const int c = t[2 + k];
        s0 += row[3 * k] * c;
        s1 += row[3 * k + 1] * c;
        s2 += row[3 * k + 2] * c;
    }
    uint8_t* o = tmp + (((size_t)f * H + y) * nw + x) * 3;
    o[0] = (uint8_t)min(max(s0 >> PIL_PRECISION_BITS, 0), 255);
    o[1] = (uint8_t)min(max(s1 >> PIL_PRECISION_BITS, 0), 255);
    o[2] = (uint8_t)min(max(s2 >> PIL_PRECISION_BITS, 0), 255);
}

// vertical pass + canvas + / 255: tmp [n][H][nw][3] (BGR) -> out [n][SH][SW][4] fp32 (R, G, B, 0)
__global__ __launch_bounds__(256) void pil_vertical_canvas_kernel(const uint8_t* __restrict__ tmp, int H, int nw, int nh, int ky,
                                                                  const int32_t* __restrict__ tab, int SH, int SW, int dx, int dy,
                                                                  float* __restrict__ out) {
    const int f = blockIdx.z, y = blockIdx.y;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= SW) return;
    int v0 = 128, v1 = 128, v2 = 128;                       // Image.new("RGB", size, (128, 128, 128))
    const int yy = y - dy, xx = x - dx;
    if (yy >= 0 && yy < nh && xx >= 0 && xx < nw) {
        const int32_t* t = tab + (size_t)yy * (2 + ky);
        const int ymin = t[0], cnt = t[1];
        const uint8_t* col = tmp + (((size_t)f * H + ymin) * nw + xx) * 3;
        int s0 = 1 << (PIL_PRECISION_BITS - 1), s1 = s0, s2 = s0;
        for (int k = 0; k < cnt; ++k) {
            const int c = t[2 + k];
            const uint8_t* p = col + (size_t)k * nw * 3;
            s0 += p[0] * c;
            s1 += p[1] * c;
            s2 += p[2] * c;
        }
        v0 = min(max(s0 >> PIL_PRECISION_BITS, 0), 255);
        v1 = min(max(s1 >> PIL_PRECISION_BITS, 0), 255);
        v2 = min(max(s2 >> PIL_PRECISION_BITS, 0), 255);
    }
    // the source is BGR, the network wants RGB
    const float4 o = make_float4((float)v2 / 255.0f, (float)v1 / 255.0f, (float)v0 / 255.0f, 0.f);
    *reinterpret_cast<float4*>(out + (((size_t)f * SH + y) * SW + x) * 4) = o;
}

__device__ __forceinline__ float sigmoid_f32(float x) { return (float)(1.0 / (1.0 + exp(-(double)x))); }
__device__ __forceinline__ float exp_f32(float x) { return (float)exp((double)x); }

struct DecodeArgs {
    const float* feats;
    int n, gh, gw, nc, cls;
    float anchors[3][2];
    float in_h, in_w, img_h, img_w;
    float new_h, new_w;           // K.round(image_shape * K.min(input_shape / image_shape))
    float* boxes;
    float* scores;
};

__global__ __launch_bounds__(256) void yolo_decode_kernel(DecodeArgs a) {
    const int per = a.gh * a.gw * 3;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n * per) return;
    const int f = i / per, r = i - f * per;
    const int an = r % 3, cell = r / 3;
    const int gy = cell / a.gw, gx = cell - gy * a.gw;
    const float* p = a.feats + ((size_t)f * a.gh * a.gw + cell) * (3 * (5 + a.nc)) + an * (5 + a.nc);
    // yolo_head
    const float bx = (sigmoid_f32(p[0]) + (float)gx) / (float)a.gw;
    const float by = (sigmoid_f32(p[1]) + (float)gy) / (float)a.gh;
    const float bw = exp_f32(p[2]) * a.anchors[an][0] / a.in_w;
    const float bh = exp_f32(p[3]) * a.anchors[an][1] / a.in_h;
    const float conf = sigmoid_f32(p[4]);
    const float prob = sigmoid_f32(p[5 + a.cls]);
    // yolo_correct_boxes (yx order)
    const float off_y = (a.in_h - a.new_h) / 2.0f / a.in_h, off_x = (a.in_w - a.new_w) / 2.0f / a.in_w;
    const float sc_y = a.in_h / a.new_h, sc_x = a.in_w / a.new_w;
    const float cy = (by - off_y) * sc_y, cx = (bx - off_x) * sc_x;
    const float hh = bh * sc_y, ww = bw * sc_x;
    const float y1 = cy - hh / 2.0f, x1 = cx - ww / 2.0f, y2 = cy + hh / 2.0f, x2 = cx + ww / 2.0f;
    float* o = a.boxes + (size_t)i * 4;
    o[0] = y1 * a.img_h;
    o[1] = x1 * a.img_w;
    o[2] = y2 * a.img_h;
    o[3] = x2 * a.img_w;
    a.scores[i] = conf * prob;
}

// rects: [n][5] = (frame, sx, sy, ex, ey); tabs: [n][(pw + ph) * 3] cv::resize tables (x entries first)
__global__ __launch_bounds__(256) void reid_patch_kernel(const uint8_t* __restrict__ frames, int H, int W,
                                                         const int32_t* __restrict__ rects, const int32_t* __restrict__ tabs,
                                                         int ph, int pw, float* __restrict__ out) {
    const int pidx = blockIdx.y;
    const int32_t* r = rects + (size_t)pidx * 5;
    const int f = r[0], sx0 = r[1], sy0 = r[2], cw = r[3] - r[1], ch = r[4] - r[2];
    if (cw <= 0 || ch <= 0) {   // empty patch (the reference substitutes an unseeded random patch): all zeros here
        for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < ph * pw; p += gridDim.x * blockDim.x)
            *reinterpret_cast<float4*>(out + ((size_t)pidx * ph * pw + p) * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const int32_t* xt = tabs + (size_t)pidx * (pw + ph) * 3;
    const int32_t* yt = xt + pw * 3;
    const uint8_t* img = frames + ((size_t)f * H + sy0) * W * 3 + (size_t)sx0 * 3;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < ph * pw; p += gridDim.x * blockDim.x) {
        const int y = p / pw, x = p - y * pw;
        const int sx = xt[3 * x], a0 = xt[3 * x + 1], a1 = xt[3 * x + 2];
        const int sy = yt[3 * y], b0 = yt[3 * y + 1], b1 = yt[3 * y + 2];
        const int sx1 = min(sx + 1, cw - 1), sy1 = min(sy + 1, ch - 1);
        const uint8_t* r0 = img + (size_t)sy * W * 3;
        const uint8_t* r1 = img + (size_t)sy1 * W * 3;
        int c3[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int h0 = r0[sx * 3 + c] * a0 + r0[sx1 * 3 + c] * a1;
            const int h1 = r1[sx * 3 + c] * a0 + r1[sx1 * 3 + c] * a1;
            const int val = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
            c3[c] = min(max(val, 0), 255);
        }
        // the patch is BGR like the frame; the graph reverses the channel axis before conv1_1
        *reinterpret_cast<float4*>(out + ((size_t)pidx * ph * pw + p) * 4) = make_float4((float)c3[2], (float)c3[1], (float)c3[0], 0.f);
    }
}

// cv::resize(INTER_LINEAR) 8-bit coefficient table for one axis: (source index, w0, w1), weights * 2048
void cv_resize_table(int src, int dst, int32_t* tab) {
    const double scale = 1.0 / ((double)dst / src);
    for (int d = 0; d < dst; ++d) {
        float fx = (float)((d + 0.5) * scale - 0.5);
        int s = (int)std::floor(fx);
        fx -= (float)s;
        if (s < 0) { fx = 0.f; s = 0; }
        if (s >= src - 1) { fx = 0.f; s = src - 1; }
        tab[3 * d] = s;
        tab[3 * d + 1] = (int32_t)std::lrintf((1.f - fx) * 2048.f);
        tab[3 * d + 2] = (int32_t)std::lrintf(fx * 2048.f);
    }
}

int launch_check(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        pp_set_error("%s launch failed: %s", what, hipGetErrorString(e));
        return PP_ERR_HIP;
    }
    return PP_OK;
}

}  // namespace

extern "C" {

int pp_letterbox_bicubic(pp_ctx* ctx, const uint8_t* frames, int n, int src_h, int src_w, int frames_mem,
                         const int32_t* xtab, int nw, int kx, const int32_t* ytab, int nh, int ky, int size_h, int size_w,
                         float* out_device) {
    PP_REQUIRE(ctx && frames && xtab && ytab && out_device, "pp_letterbox_bicubic: NULL argument");
    PP_REQUIRE(n > 0 && src_h > 0 && src_w > 0 && nw > 0 && nh > 0 && nw <= size_w && nh <= size_h && kx > 0 && ky > 0,
               "pp_letterbox_bicubic: bad dims");
    const size_t frame_bytes = (size_t)n * src_h * src_w * 3;
    const size_t xt_bytes = (size_t)nw * (2 + kx) * 4, yt_bytes = (size_t)nh * (2 + ky) * 4;
    const size_t tmp_bytes = (size_t)n * src_h * nw * 3;
    size_t need = ScratchCursor::align(xt_bytes) + ScratchCursor::align(yt_bytes) + ScratchCursor::align(tmp_bytes);
    if (frames_mem == PP_MEM_HOST) need += ScratchCursor::align(frame_bytes);
    int rc = ctx->ensure_scratch(need);
    if (rc != PP_OK) return rc;
    ScratchCursor cur(ctx);
    int32_t* d_xt = cur.take<int32_t>(xt_bytes / 4);
    int32_t* d_yt = cur.take<int32_t>(yt_bytes / 4);
    uint8_t* d_tmp = cur.take<uint8_t>(tmp_bytes);
    hipStream_t s = ctx->stream;
    const uint8_t* d_frames = frames;
    if (frames_mem == PP_MEM_HOST) {
        uint8_t* st = cur.take<uint8_t>(frame_bytes);
        PP_HIP_CHECK(hipMemcpyAsync(st, frames, frame_bytes, hipMemcpyHostToDevice, s));
        d_frames = st;
    }
    PP_HIP_CHECK(hipMemcpyAsync(d_xt, xtab, xt_bytes, hipMemcpyHostToDevice, s));
    PP_HIP_CHECK(hipMemcpyAsync(d_yt, ytab, yt_bytes, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(pil_horizontal_kernel, dim3((nw + 255) / 256, src_h, n), dim3(256), 0, s, d_frames, src_h, src_w, nw, kx,
                       d_xt, d_tmp);
    rc = launch_check("pil_horizontal");
    if (rc != PP_OK) return rc;
    hipLaunchKernelGGL(pil_vertical_canvas_kernel, dim3((size_w + 255) / 256, size_h, n), dim3(256), 0, s, d_tmp, src_h, nw, nh,
                       ky, d_yt, size_h, size_w, (size_w - nw) / 2, (size_h - nh) / 2, out_device);
    rc = launch_check("pil_vertical_canvas");
    if (rc != PP_OK) return rc;
    PP_HIP_CHECK(hipStreamSynchronize(s));     // the host tables / staged frames must outlive the copies
    return PP_OK;
}

int pp_yolo_decode(pp_ctx* ctx, const float* feats, int n, int gh, int gw, int num_classes, int cls,
                   const float* anchors3x2, int input_h, int input_w, int image_h, int image_w, float* boxes,
                   float* scores, int out_mem) {
    PP_REQUIRE(ctx && feats && anchors3x2 && boxes && scores, "pp_yolo_decode: NULL argument");
    PP_REQUIRE(n > 0 && gh > 0 && gw > 0 && num_classes > 0 && cls >= 0 && cls < num_classes, "pp_yolo_decode: bad dims");
    const size_t cnt = (size_t)n * gh * gw * 3;
    hipStream_t s = ctx->stream;
    float* d_boxes = boxes;
    float* d_scores = scores;
    if (out_mem == PP_MEM_HOST) {
        int rc = ctx->ensure_scratch(ScratchCursor::align(cnt * 16) + ScratchCursor::align(cnt * 4));
        if (rc != PP_OK) return rc;
        ScratchCursor cur(ctx);
        d_boxes = cur.take<float>(cnt * 4);
        d_scores = cur.take<float>(cnt);
    }
    DecodeArgs a{};
    a.feats = feats; a.n = n; a.gh = gh; a.gw = gw; a.nc = num_classes; a.cls = cls;
    for (int i = 0; i < 3; ++i) { a.anchors[i][0] = anchors3x2[2 * i]; a.anchors[i][1] = anchors3x2[2 * i + 1]; }
    a.in_h = (float)input_h; a.in_w = (float)input_w; a.img_h = (float)image_h; a.img_w = (float)image_w;
    // new_shape = K.round(image_shape * K.min(input_shape / image_shape)): float32 ops, round half to even
    const float m = std::min(a.in_h / a.img_h, a.in_w / a.img_w);
    a.new_h = std::nearbyintf(a.img_h * m);
    a.new_w = std::nearbyintf(a.img_w * m);
    a.boxes = d_boxes; a.scores = d_scores;
    hipLaunchKernelGGL(yolo_decode_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, s, a);
    int rc = launch_check("yolo_decode");
    if (rc != PP_OK) return rc;
    if (out_mem == PP_MEM_HOST) {
        PP_HIP_CHECK(hipMemcpyAsync(boxes, d_boxes, cnt * 16, hipMemcpyDeviceToHost, s));
        PP_HIP_CHECK(hipMemcpyAsync(scores, d_scores, cnt * 4, hipMemcpyDeviceToHost, s));
        PP_HIP_CHECK(hipStreamSynchronize(s));
    }
    return PP_OK;
}

int pp_reid_patches(pp_ctx* ctx, const uint8_t* frames, int n_frames, int src_h, int src_w, int frames_mem,
                    const int32_t* rects, int n, int ph, int pw, float* out_device) {
    PP_REQUIRE(ctx && frames && out_device && (n == 0 || rects), "pp_reid_patches: NULL argument");
    PP_REQUIRE(n_frames > 0 && src_h > 0 && src_w > 0 && n >= 0 && ph > 0 && pw > 0, "pp_reid_patches: bad dims");
    if (n == 0) return PP_OK;
    std::vector<int32_t> tabs((size_t)n * (pw + ph) * 3);
    for (int i = 0; i < n; ++i) {
        const int32_t* r = rects + (size_t)i * 5;
        if (r[3] <= r[1] || r[4] <= r[2]) continue;      // empty patch -> zeros
        PP_REQUIRE(r[0] >= 0 && r[0] < n_frames && r[1] >= 0 && r[2] >= 0 && r[3] <= src_w && r[4] <= src_h,
                   "pp_reid_patches: rect %d out of the frame", i);
        cv_resize_table(r[3] - r[1], pw, tabs.data() + (size_t)i * (pw + ph) * 3);
        cv_resize_table(r[4] - r[2], ph, tabs.data() + (size_t)i * (pw + ph) * 3 + pw * 3);
    }
    const size_t frame_bytes = (size_t)n_frames * src_h * src_w * 3;
    size_t need = ScratchCursor::align(tabs.size() * 4) + ScratchCursor::align((size_t)n * 5 * 4);
    if (frames_mem == PP_MEM_HOST) need += ScratchCursor::align(frame_bytes);
    int rc = ctx->ensure_scratch(need);
    if (rc != PP_OK) return rc;
    ScratchCursor cur(ctx);
    int32_t* d_tabs = cur.take<int32_t>(tabs.size());
    int32_t* d_rects = cur.take<int32_t>((size_t)n * 5);
    hipStream_t s = ctx->stream;
    const uint8_t* d_frames = frames;
    if (frames_mem == PP_MEM_HOST) {
        uint8_t* st = cur.take<uint8_t>(frame_bytes);
        PP_HIP_CHECK(hipMemcpyAsync(st, frames, frame_bytes, hipMemcpyHostToDevice, s));
        d_frames = st;
    }
    PP_HIP_CHECK(hipMemcpyAsync(d_tabs, tabs.data(), tabs.size() * 4, hipMemcpyHostToDevice, s));
    PP_HIP_CHECK(hipMemcpyAsync(d_rects, rects, (size_t)n * 5 * 4, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(reid_patch_kernel, dim3((ph * pw + 255) / 256, n), dim3(256), 0, s, d_frames, src_h, src_w, d_rects, d_tabs,
                       ph, pw, out_device);
    rc = launch_check("reid_patch");
    if (rc != PP_OK) return rc;
    PP_HIP_CHECK(hipStreamSynchronize(s));
    return PP_OK;
}

}  // extern "C"
