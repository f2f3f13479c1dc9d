// Non-maximum suppression on the device: score sort -> 64x64 bitmask tiles (one 64-bit word per
// lane, the wave64 analogue of the classic 32-bit mask kernel) -> sequential sweep by one wave.
//
// convention 0 -- mmcv-full `nms` / torchvision semantics used by the Faster-RCNN RPN and RoI head
//   (3rdparty/mmtracking/_base_/models/faster_rcnn_r50_fpn.py:101-109): float32 x1y1x2y2, area = w*h,
//   suppress IoU > thr, survivors in descending score order (ties: lower index first).
// convention 1 -- the in-tree greedy NMS, wrappers/deep_sort_yolov4/deep_sort/preprocessing.py:5-70:
//   float64 (x, y, w, h) boxes, +1-pixel areas (:47), overlap = intersection / area of the OTHER box (:66),
//   suppress overlap > thr, survivors in descending score order.
#include "pp_internal.h"

namespace {

constexpr int MAX_N = 8192;

template <typename T>
__global__ __launch_bounds__(1024) void sort_desc_kernel(const T* __restrict__ scores, int n, int npow2,
                                                         int32_t* __restrict__ order) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* key = reinterpret_cast<T*>(smem_raw);
    int32_t* idx = reinterpret_cast<int32_t*>(key + npow2);
    for (int i = threadIdx.x; i < npow2; i += blockDim.x) {
        key[i] = i < n ? scores[i] : (T)-INFINITY;
        idx[i] = i < n ? i : 0x7fffffff;
    }
    __syncthreads();
    // bitonic sort; "a before b" <=> key_a > key_b || (key_a == key_b && idx_a < idx_b)
    for (int k = 2; k <= npow2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < npow2; i += blockDim.x) {
                const int l = i ^ j;
                if (l > i) {
                    const bool up = (i & k) == 0;
                    const T ka = key[i], kb = key[l];
                    const int ia = idx[i], ib = idx[l];
                    const bool a_first = ka > kb || (ka == kb && ia < ib);
                    if (a_first != up) {
                        key[i] = kb; key[l] = ka;
                        idx[i] = ib; idx[l] = ia;
                    }
                }
            }
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < n; i += blockDim.x) order[i] = idx[i];
}

template <typename T, int CONV>
__device__ __forceinline__ bool suppresses(const T* a, const T* b, T thr) {
    // does (higher-score) box a suppress box b?
    if (CONV == 0) {
        const T area_a = (a[2] - a[0]) * (a[3] - a[1]);
        const T area_b = (b[2] - b[0]) * (b[3] - b[1]);
        const T w = max(min(a[2], b[2]) - max(a[0], b[0]), (T)0);
        const T h = max(min(a[3], b[3]) - max(a[1], b[1]), (T)0);
        const T inter = w * h;
        return inter / (area_a + area_b - inter) > thr;
    } else {
        const T ax2 = a[2] + a[0], ay2 = a[3] + a[1], bx2 = b[2] + b[0], by2 = b[3] + b[1];
        const T area_b = (bx2 - b[0] + 1) * (by2 - b[1] + 1);
        const T w = max((T)0, min(ax2, bx2) - max(a[0], b[0]) + 1);
        const T h = max((T)0, min(ay2, by2) - max(a[1], b[1]) + 1);
        return (w * h) / area_b > thr;
    }
}

template <typename T, int CONV>
__global__ __launch_bounds__(64) void nms_mask_kernel(const T* __restrict__ boxes, const int32_t* __restrict__ order,
                                                      int n, T thr, unsigned long long* __restrict__ mask, int words) {
    const int row0 = blockIdx.y * 64, col0 = blockIdx.x * 64;
    if (col0 + 63 < row0) return;            // only j > i matters
    __shared__ T cb[64 * 4];
    const int cj = col0 + threadIdx.x;
    if (cj < n) {
        const T* b = boxes + (size_t)order[cj] * 4;
        cb[threadIdx.x * 4 + 0] = b[0]; cb[threadIdx.x * 4 + 1] = b[1];
        cb[threadIdx.x * 4 + 2] = b[2]; cb[threadIdx.x * 4 + 3] = b[3];
    }
    __syncthreads();
    const int i = row0 + threadIdx.x;
    if (i >= n) return;
    T a[4];
    const T* ap = boxes + (size_t)order[i] * 4;
    a[0] = ap[0]; a[1] = ap[1]; a[2] = ap[2]; a[3] = ap[3];
    unsigned long long bits = 0;
    const int jmax = min(64, n - col0);
    for (int j = 0; j < jmax; ++j) {
        if (col0 + j > i && suppresses<T, CONV>(a, cb + j * 4, thr)) bits |= 1ull << j;
    }
    mask[(size_t)i * words + blockIdx.x] = bits;
}

// one wave: lane l owns removed-words l, l+64, ...
__global__ __launch_bounds__(64) void nms_sweep_kernel(const unsigned long long* __restrict__ mask,
                                                       const int32_t* __restrict__ order, int n, int words,
                                                       int32_t* __restrict__ keep, int32_t* __restrict__ n_keep) {
    constexpr int WPL = MAX_N / 64 / 64;     // words per lane (2)
    unsigned long long removed[WPL];
#pragma unroll
    for (int k = 0; k < WPL; ++k) removed[k] = 0;
    const int lane = threadIdx.x;
    int cnt = 0;
    for (int i = 0; i < n; ++i) {
        const int w = i >> 6;
        // is bit i set in removed-word w (held by lane w & 63, slot w >> 6)?
        unsigned long long word = 0;
#pragma unroll
        for (int k = 0; k < WPL; ++k)
            if ((w >> 6) == k) word = removed[k];
        word = __shfl(word, w & 63, 64);
        if (!((word >> (i & 63)) & 1ull)) {
            if (lane == 0) keep[cnt] = order[i];
            cnt++;
            const unsigned long long* row = mask + (size_t)i * words;
#pragma unroll
            for (int k = 0; k < WPL; ++k) {
                const int ww = lane + 64 * k;
                if (ww >= w && ww < words) removed[k] |= row[ww];
            }
        }
    }
    if (lane == 0) *n_keep = cnt;
}

template <typename T, int CONV>
int run_nms(pp_ctx* ctx, const T* d_boxes, const T* d_scores, int n, T thr, int32_t* d_order,
            unsigned long long* d_mask, int32_t* d_keep, int32_t* d_nkeep) {
    hipStream_t s = ctx->stream;
    int npow2 = 1;
    while (npow2 < n) npow2 <<= 1;
    const size_t lds = (size_t)npow2 * (sizeof(T) + sizeof(int32_t));
    static bool attr_set = false;
    if (!attr_set) {
        PP_HIP_CHECK(hipFuncSetAttribute((const void*)sort_desc_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         MAX_N * (sizeof(T) + sizeof(int32_t))));
        attr_set = true;
    }
    hipLaunchKernelGGL((sort_desc_kernel<T>), dim3(1), dim3(1024), lds, s, d_scores, n, npow2, d_order);
    const int words = (n + 63) / 64;
    PP_HIP_CHECK(hipMemsetAsync(d_mask, 0, (size_t)n * words * sizeof(unsigned long long), s));
    hipLaunchKernelGGL((nms_mask_kernel<T, CONV>), dim3(words, words), dim3(64), 0, s, d_boxes, d_order, n, thr, d_mask, words);
    hipLaunchKernelGGL(nms_sweep_kernel, dim3(1), dim3(64), 0, s, d_mask, d_order, n, words, d_keep, d_nkeep);
    PP_HIP_CHECK(hipGetLastError());
    return PP_OK;
}

}  // namespace

// device-pointer entry used by the detector: float32 boxes, convention 0, nothing synchronised.
// scratch must hold n int32 (order) + n*ceil(n/64) u64 (mask).
int pp_enqueue_nms_f32(pp_ctx* ctx, const float* boxes, const float* scores, int n, float thr, void* scratch,
                       int32_t* keep, int32_t* n_keep) {
    PP_REQUIRE(n <= MAX_N, "nms: n=%d exceeds %d", n, MAX_N);
    if (n <= 0) {
        PP_HIP_CHECK(hipMemsetAsync(n_keep, 0, sizeof(int32_t), ctx->stream));
        return PP_OK;
    }
    int32_t* order = static_cast<int32_t*>(scratch);
    unsigned long long* mask = reinterpret_cast<unsigned long long*>(static_cast<char*>(scratch) + ScratchCursor::align((size_t)n * 4));
    return run_nms<float, 0>(ctx, boxes, scores, n, thr, order, mask, keep, n_keep);
}

size_t pp_nms_scratch_bytes(int n) {
    return ScratchCursor::align((size_t)n * 4) + ScratchCursor::align((size_t)n * ((n + 63) / 64) * 8);
}

extern "C" int pp_nms(pp_ctx* ctx, const void* boxes, const void* scores, int n, double iou_thr, int convention,
                      int32_t* keep, int32_t* n_keep, int mem) {
    PP_REQUIRE(ctx && n_keep && (n == 0 || (boxes && scores && keep)), "pp_nms: NULL argument");
    PP_REQUIRE(convention == 0 || convention == 1, "pp_nms: convention must be 0 (mmcv, float32 xyxy) or 1 (deep_sort, float64 tlwh)");
    PP_REQUIRE(n >= 0 && n <= MAX_N, "pp_nms: n=%d not in [0,%d]", n, MAX_N);
    if (n == 0) {
        if (mem == PP_MEM_HOST) *n_keep = 0;
        else PP_HIP_CHECK(hipMemsetAsync(n_keep, 0, sizeof(int32_t), ctx->stream));
        return PP_OK;
    }
    const size_t esz = convention == 0 ? 4 : 8;
    const int words = (n + 63) / 64;
    size_t need = ScratchCursor::align((size_t)n * 4) + ScratchCursor::align((size_t)n * words * 8);
    if (mem == PP_MEM_HOST) need += ScratchCursor::align((size_t)n * 4 * esz) + ScratchCursor::align((size_t)n * esz) + 2 * ScratchCursor::align((size_t)n * 4 + 256);
    int rc = ctx->ensure_scratch(need);
    if (rc != PP_OK) return rc;
    ScratchCursor cur(ctx);
    hipStream_t s = ctx->stream;
    int32_t* d_order = cur.take<int32_t>(n);
    unsigned long long* d_mask = cur.take<unsigned long long>((size_t)n * words);
    const void* d_boxes = boxes;
    const void* d_scores = scores;
    int32_t* d_keep = keep;
    int32_t* d_nkeep = n_keep;
    if (mem == PP_MEM_HOST) {
        char* db = cur.take<char>((size_t)n * 4 * esz);
        char* ds = cur.take<char>((size_t)n * esz);
        d_keep = cur.take<int32_t>(n);
        d_nkeep = cur.take<int32_t>(1);
        PP_HIP_CHECK(hipMemcpyAsync(db, boxes, (size_t)n * 4 * esz, hipMemcpyHostToDevice, s));
        PP_HIP_CHECK(hipMemcpyAsync(ds, scores, (size_t)n * esz, hipMemcpyHostToDevice, s));
        d_boxes = db;
        d_scores = ds;
    }
    if (convention == 0)
        rc = run_nms<float, 0>(ctx, (const float*)d_boxes, (const float*)d_scores, n, (float)iou_thr, d_order, d_mask, d_keep, d_nkeep);
    else
        rc = run_nms<double, 1>(ctx, (const double*)d_boxes, (const double*)d_scores, n, iou_thr, d_order, d_mask, d_keep, d_nkeep);
    if (rc != PP_OK) return rc;
    if (mem == PP_MEM_HOST) {
        PP_HIP_CHECK(hipMemcpyAsync(n_keep, d_nkeep, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        PP_HIP_CHECK(hipMemcpyAsync(keep, d_keep, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    }
    PP_HIP_CHECK(hipStreamSynchronize(s));
    return PP_OK;
}
