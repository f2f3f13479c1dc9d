// Non-maximum suppression on the device, batched over frames: score sort -> 64x64 bitmask tiles (one
// 64-bit word per lane, the wave64 analogue of the classic 32-bit mask kernel) -> sequential sweep by one
// wave per frame.  Box counts live on the device, so the detector chains these without a host sync.
//
// convention 0 -- mmcv-full `nms` / `batched_nms` used by the Faster-RCNN RPN and RoI head
//   (3rdparty/mmtracking/_base_/models/faster_rcnn_r50_fpn.py:101-109): float32 x1y1x2y2, area = w*h,
//   suppress when inter > thr * (Sa + Sb - inter) (mmcv's devIoU form), survivors in descending score order
//   (ties: lower index first).
// convention 1 -- the in-tree greedy NMS, wrappers/deep_sort_yolov4/deep_sort/preprocessing.py:5-70:
//   float64 (x, y, w, h) boxes, +1-pixel areas (:47), overlap = intersection / area of the OTHER box (:66),
//   suppress overlap > thr, survivors in descending score order.
// convention 2 -- tf.image.non_max_suppression as called by wrappers/deep_sort_yolov4/yolo4/model.py:278-281:
//   float32 (y1, x1, y2, x2) with corners in any order, IoU = inter / (Sa + Sb - inter) (0 when an area is <= 0),
//   suppress IoU > thr, survivors in descending score order (ties: lower index first); the caller truncates to
//   max_output_size.
#include "pp_internal.h"
#include "det_internal.h"

namespace {

constexpr int MAX_N = 8192;

template <typename T>
__global__ __launch_bounds__(1024) void sort_desc_kernel(const T* __restrict__ scores, const int32_t* __restrict__ n_ptr,
                                                         int max_n, int32_t* __restrict__ order) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int f = blockIdx.x;
    const int n = n_ptr[f];
    int npow2 = 1;
    while (npow2 < n) npow2 <<= 1;
    T* key = reinterpret_cast<T*>(smem_raw);
    int32_t* idx = reinterpret_cast<int32_t*>(smem_raw + (size_t)MAX_N * sizeof(T));
    const T* sc = scores + (size_t)f * max_n;
    for (int i = threadIdx.x; i < npow2; i += blockDim.x) {
        key[i] = i < n ? sc[i] : (T)-INFINITY;
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
    int32_t* o = order + (size_t)f * max_n;
    for (int i = threadIdx.x; i < n; i += blockDim.x) o[i] = idx[i];
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
        // mmcv devIoU: interS > threshold * (Sa + Sb - interS), no division (-ffp-contract=off: no fma)
        return inter > thr * ((area_a + area_b) - inter);
    } else if (CONV == 2) {
        const T ay1 = min(a[0], a[2]), ay2 = max(a[0], a[2]), ax1 = min(a[1], a[3]), ax2 = max(a[1], a[3]);
        const T by1 = min(b[0], b[2]), by2 = max(b[0], b[2]), bx1 = min(b[1], b[3]), bx2 = max(b[1], b[3]);
        const T area_a = (ay2 - ay1) * (ax2 - ax1), area_b = (by2 - by1) * (bx2 - bx1);
        if (area_a <= (T)0 || area_b <= (T)0) return false;
        const T h = max(min(ay2, by2) - max(ay1, by1), (T)0);
        const T w = max(min(ax2, bx2) - max(ax1, bx1), (T)0);
        const T inter = h * w;
        return inter / ((area_a + area_b) - inter) > thr;
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
                                                      const int32_t* __restrict__ n_ptr, int max_n, T thr,
                                                      unsigned long long* __restrict__ mask, int words) {
    const int f = blockIdx.z;
    const int n = n_ptr[f];
    const int row0 = blockIdx.y * 64, col0 = blockIdx.x * 64;
    if (row0 >= n || col0 + 63 < row0) return;            // only j > i matters; the sweep never reads the rest
    const T* bx = boxes + (size_t)f * max_n * 4;
    const int32_t* ord = order + (size_t)f * max_n;
    __shared__ T cb[64 * 4];
    const int cj = col0 + threadIdx.x;
    if (cj < n) {
        const T* b = bx + (size_t)ord[cj] * 4;
        cb[threadIdx.x * 4 + 0] = b[0]; cb[threadIdx.x * 4 + 1] = b[1];
        cb[threadIdx.x * 4 + 2] = b[2]; cb[threadIdx.x * 4 + 3] = b[3];
    }
    __syncthreads();
    const int i = row0 + threadIdx.x;
    if (i >= n) return;
    T a[4];
    const T* ap = bx + (size_t)ord[i] * 4;
    a[0] = ap[0]; a[1] = ap[1]; a[2] = ap[2]; a[3] = ap[3];
    unsigned long long bits = 0;
    const int jmax = min(64, n - col0);
    for (int j = 0; j < jmax; ++j) {
        if (col0 + j > i && suppresses<T, CONV>(a, cb + j * 4, thr)) bits |= 1ull << j;
    }
    mask[((size_t)f * max_n + i) * words + blockIdx.x] = bits;
}

// one wave per frame: lane l owns removed-words l, l+64
__global__ __launch_bounds__(64) void nms_sweep_kernel(const unsigned long long* __restrict__ mask,
                                                       const int32_t* __restrict__ order, const int32_t* __restrict__ n_ptr,
                                                       int max_n, int words, int32_t* __restrict__ keep,
                                                       int32_t* __restrict__ n_keep) {
    constexpr int WPL = MAX_N / 64 / 64;     // words per lane (2)
    const int f = blockIdx.x;
    const int n = n_ptr[f];
    const int32_t* ord = order + (size_t)f * max_n;
    int32_t* kp = keep + (size_t)f * max_n;
    unsigned long long removed[WPL];
#pragma unroll
    for (int k = 0; k < WPL; ++k) removed[k] = 0;
    const int lane = threadIdx.x;
    int cnt = 0;
    // 64 boxes (= one mask word) at a time: resolve the chunk against itself with the diagonal words held in
    // registers (no memory on the sequential path), then OR the surviving rows into the later words with
    // independent, batched loads.
    const int nchunks = (n + 63) >> 6;
    for (int c = 0; c < nchunks; ++c) {
        unsigned long long cur = 0;
#pragma unroll
        for (int k = 0; k < WPL; ++k)
            if ((c >> 6) == k) cur = removed[k];
        cur = __shfl(cur, c & 63, 64);                         // removed bits of this chunk (uniform)
        const int i = (c << 6) + lane;
        const unsigned long long diag = i < n ? mask[((size_t)f * max_n + i) * words + c] : 0ull;
        const int nb = min(64, n - (c << 6));
        unsigned long long keepmask = 0;
        for (int b = 0; b < nb; ++b) {
            const unsigned long long d = __shfl(diag, b, 64);
            if (!((cur >> b) & 1ull)) {
                keepmask |= 1ull << b;
                cur |= d;
            }
        }
        if ((keepmask >> lane) & 1ull) kp[cnt + __popcll(keepmask & ((1ull << lane) - 1ull))] = ord[i];
        cnt += __popcll(keepmask);
        // later words: lane owns words lane and lane + 64
        unsigned long long km = keepmask;
        while (km) {
            unsigned long long v[4][WPL];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int k = 0; k < WPL; ++k) v[u][k] = 0;
                if (km) {
                    const int b = __ffsll((long long)km) - 1;
                    km &= km - 1;
                    const unsigned long long* row = mask + ((size_t)f * max_n + (c << 6) + b) * words;
#pragma unroll
                    for (int k = 0; k < WPL; ++k) {
                        const int ww = lane + 64 * k;
                        if (ww > c && ww < words) v[u][k] = row[ww];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < WPL; ++k) removed[k] |= v[u][k];
        }
    }
    if (lane == 0) n_keep[f] = cnt;
}

template <typename T, int CONV>
int run_nms(hipStream_t s, const T* d_boxes, const T* d_scores, const int32_t* d_n, int max_n, int n_frames, T thr,
            int32_t* d_order, unsigned long long* d_mask, int32_t* d_keep, int32_t* d_nkeep) {
    const size_t lds = (size_t)MAX_N * (sizeof(T) + sizeof(int32_t));
    static PpPerDeviceOnce attr_set;
    attr_set.run([&] { (void)hipFuncSetAttribute((const void*)sort_desc_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
    hipLaunchKernelGGL((sort_desc_kernel<T>), dim3(n_frames), dim3(1024), lds, s, d_scores, d_n, max_n, d_order);
    const int words = (max_n + 63) / 64;
    hipLaunchKernelGGL((nms_mask_kernel<T, CONV>), dim3(words, words, n_frames), dim3(64), 0, s, d_boxes, d_order, d_n, max_n,
                       thr, d_mask, words);
    hipLaunchKernelGGL(nms_sweep_kernel, dim3(n_frames), dim3(64), 0, s, d_mask, d_order, d_n, max_n, words, d_keep, d_nkeep);
    PP_HIP_CHECK(hipGetLastError());
    return PP_OK;
}

size_t scratch_bytes(int max_n, int n_frames) {
    const int words = (max_n + 63) / 64;
    return ScratchCursor::align((size_t)n_frames * max_n * 4) + ScratchCursor::align((size_t)n_frames * max_n * words * 8);
}

}  // namespace

size_t pp_nms_batched_scratch_bytes(int max_n, int n_frames) { return scratch_bytes(max_n, n_frames); }

int pp_enqueue_nms_batched(hipStream_t s, const float* boxes, const float* scores, const int32_t* n, int max_n,
                           int n_frames, float thr, void* scratch, int32_t* keep, int32_t* n_keep) {
    PP_REQUIRE(max_n > 0 && max_n <= MAX_N, "nms: max_n=%d not in (0,%d]", max_n, MAX_N);
    int32_t* order = static_cast<int32_t*>(scratch);
    unsigned long long* mask = reinterpret_cast<unsigned long long*>(
        static_cast<char*>(scratch) + ScratchCursor::align((size_t)n_frames * max_n * 4));
    return run_nms<float, 0>(s, boxes, scores, n, max_n, n_frames, thr, order, mask, keep, n_keep);
}

extern "C" int pp_nms(pp_ctx* ctx, const void* boxes, const void* scores, int n, double iou_thr, int convention,
                      int32_t* keep, int32_t* n_keep, int mem) {
    PP_REQUIRE(ctx && n_keep && (n == 0 || (boxes && scores && keep)), "pp_nms: NULL argument");
    PP_REQUIRE(convention >= 0 && convention <= 2,
               "pp_nms: convention must be 0 (mmcv, float32 xyxy), 1 (deep_sort, float64 tlwh) or 2 (TensorFlow, float32 yxyx)");
    PP_REQUIRE(n >= 0 && n <= MAX_N, "pp_nms: n=%d not in [0,%d]", n, MAX_N);
    hipStream_t s = ctx->stream;
    if (n == 0) {
        if (mem == PP_MEM_HOST) *n_keep = 0;
        else PP_HIP_CHECK(hipMemsetAsync(n_keep, 0, sizeof(int32_t), s));
        return PP_OK;
    }
    const size_t esz = convention == 1 ? 8 : 4;
    size_t need = scratch_bytes(n, 1) + ScratchCursor::align(4);
    if (mem == PP_MEM_HOST) need += ScratchCursor::align((size_t)n * 4 * esz) + ScratchCursor::align((size_t)n * esz) + 2 * ScratchCursor::align((size_t)n * 4 + 256);
    int rc = ctx->ensure_scratch(need);
    if (rc != PP_OK) return rc;
    ScratchCursor cur(ctx);
    const int words = (n + 63) / 64;
    int32_t* d_order = cur.take<int32_t>(n);
    unsigned long long* d_mask = cur.take<unsigned long long>((size_t)n * words);
    int32_t* d_n = cur.take<int32_t>(1);
    PP_HIP_CHECK(hipMemcpyAsync(d_n, &n, sizeof(int32_t), hipMemcpyHostToDevice, s));
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
        rc = run_nms<float, 0>(s, (const float*)d_boxes, (const float*)d_scores, d_n, n, 1, (float)iou_thr, d_order, d_mask, d_keep, d_nkeep);
    else if (convention == 2)
        rc = run_nms<float, 2>(s, (const float*)d_boxes, (const float*)d_scores, d_n, n, 1, (float)iou_thr, d_order, d_mask, d_keep, d_nkeep);
    else
        rc = run_nms<double, 1>(s, (const double*)d_boxes, (const double*)d_scores, d_n, n, 1, iou_thr, d_order, d_mask, d_keep, d_nkeep);
    if (rc != PP_OK) return rc;
    if (mem == PP_MEM_HOST) {
        PP_HIP_CHECK(hipMemcpyAsync(n_keep, d_nkeep, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        PP_HIP_CHECK(hipMemcpyAsync(keep, d_keep, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    }
    PP_HIP_CHECK(hipStreamSynchronize(s));   // &n and the scratch staging must outlive the copies
    return PP_OK;
}
