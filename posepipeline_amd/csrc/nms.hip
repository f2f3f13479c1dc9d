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
                                                       int32_t* __restrict__ n_keep, unsigned char* __restrict__ kflag) {
    // kflag (may be null; decision margins): [frame][max_n], 1 where the box at that SORTED position is kept
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
        if (kflag && lane < nb) kflag[(size_t)f * max_n + i] = (unsigned char)((keepmask >> lane) & 1ull);
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

// ---- decision margin of a finished NMS (convention 0) ----------------------------------------------------------------------------
// Which perturbation of the inputs leaves the kept LIST as it is?  Walk the boxes in score order and assume every earlier decision
// stands.  A KEPT box j stays kept while every kept predecessor i stays under the threshold: margin thr - IoU(i, j).  A SUPPRESSED box j
// stays suppressed while ONE kept predecessor keeps suppressing it: the best over its suppressors of min(IoU - thr, w (s_i - s_j)) --
// the second term because a suppressor only counts while it stays AHEAD of j in the order (w converts a score lead into IoU units:
// pp_detector_enable_margins).  A later box that would overtake j and suppress it is itself suppressed or kept today and contributes its
// own margin.  The frame's figure is the minimum over all boxes; by induction over the order, inputs perturbed by less than it (IoU) --
// and scores by less than it / (2 w) -- give the same kept list in the same order of decisions.
// grid (ceil(max_n / 256), frames), 256 threads: thread = box j, tiles of 256 predecessors through LDS; only kept predecessors cost.
__global__ __launch_bounds__(256) void nms_margin_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                         const int32_t* __restrict__ order, const unsigned char* __restrict__ kflag,
                                                         const int32_t* __restrict__ n_ptr, int max_n, float thr, float score_weight,
                                                         unsigned* __restrict__ margin, int margin_stride) {
    const int f = blockIdx.y;
    const int n = n_ptr[f];
    const int j0 = blockIdx.x * 256;
    if (j0 >= n) return;
    __shared__ float4 s_box[256];
    __shared__ float s_sc[256];
    __shared__ unsigned char s_kf[256];
    __shared__ float s_red[4];
    const float* bx = boxes + (size_t)f * max_n * 4;
    const float* sc = scores + (size_t)f * max_n;
    const int32_t* ord = order + (size_t)f * max_n;
    const unsigned char* kf = kflag + (size_t)f * max_n;
    const int j = j0 + threadIdx.x;
    float4 bj = make_float4(0.f, 0.f, 0.f, 0.f);
    float sj = 0.f, area_j = 0.f;
    bool kept_j = false;
    if (j < n) {
        const int oj = ord[j];
        bj = *reinterpret_cast<const float4*>(bx + (size_t)oj * 4);
        sj = sc[oj];
        kept_j = kf[j] != 0;
        area_j = (bj.z - bj.x) * (bj.w - bj.y);
    }
    float m_keep = INFINITY, m_sup = -1.f;
    const int jend = min(j0 + 256, n);
    for (int i0 = 0; i0 < jend; i0 += 256) {
        __syncthreads();
        const int i = i0 + threadIdx.x;
        if (i < n) {
            const int oi = ord[i];
            s_box[threadIdx.x] = *reinterpret_cast<const float4*>(bx + (size_t)oi * 4);
            s_sc[threadIdx.x] = sc[oi];
            s_kf[threadIdx.x] = kf[i];
        } else {
            s_kf[threadIdx.x] = 0;
        }
        __syncthreads();
        const int cnt = min(256, jend - i0);
        for (int t = 0; t < cnt; ++t) {
            if (!s_kf[t]) continue;                    // (uniform: every lane looks at the same predecessor)
            if (i0 + t >= j) continue;
            const float4 a = s_box[t];
            const float area_a = (a.z - a.x) * (a.w - a.y);
            const float w = fmaxf(fminf(a.z, bj.z) - fmaxf(a.x, bj.x), 0.f);
            const float h = fmaxf(fminf(a.w, bj.w) - fmaxf(a.y, bj.y), 0.f);
            const float inter = w * h;
            const float uni = (area_a + area_j) - inter;
            const float iou = uni > 0.f ? inter / uni : 0.f;
            const float d = iou - thr;
            if (kept_j) m_keep = fminf(m_keep, -d);
            else if (d > 0.f) m_sup = fmaxf(m_sup, fminf(d, score_weight * (s_sc[t] - sj)));
        }
    }
    // (the division above and the product form of the decision can disagree by an ulp at the threshold: clamp at zero)
    float mj = INFINITY;
    if (j < n) mj = kept_j ? fmaxf(m_keep, 0.f) : fmaxf(m_sup, 0.f);
    for (int off = 32; off > 0; off >>= 1) mj = fminf(mj, __shfl_down(mj, off, 64));
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = mj;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float r = fminf(fminf(s_red[0], s_red[1]), fminf(s_red[2], s_red[3]));
        atomicMin(margin + (size_t)f * margin_stride, __float_as_uint(r));     // non-negative floats order like their bit patterns
    }
}

__global__ void fill_u32_strided_kernel(unsigned* p, int stride, int n, unsigned v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[(size_t)i * stride] = v;
}

template <typename T, int CONV>
int run_nms(hipStream_t s, const T* d_boxes, const T* d_scores, const int32_t* d_n, int max_n, int n_frames, T thr,
            int32_t* d_order, unsigned long long* d_mask, int32_t* d_keep, int32_t* d_nkeep, unsigned char* d_kflag = nullptr) {
    const size_t lds = (size_t)MAX_N * (sizeof(T) + sizeof(int32_t));
    static PpPerDeviceOnce attr_set;
    attr_set.run([&] { (void)hipFuncSetAttribute((const void*)sort_desc_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
    hipLaunchKernelGGL((sort_desc_kernel<T>), dim3(n_frames), dim3(1024), lds, s, d_scores, d_n, max_n, d_order);
    const int words = (max_n + 63) / 64;
    hipLaunchKernelGGL((nms_mask_kernel<T, CONV>), dim3(words, words, n_frames), dim3(64), 0, s, d_boxes, d_order, d_n, max_n,
                       thr, d_mask, words);
    hipLaunchKernelGGL(nms_sweep_kernel, dim3(n_frames), dim3(64), 0, s, d_mask, d_order, d_n, max_n, words, d_keep, d_nkeep, d_kflag);
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
                           int n_frames, float thr, void* scratch, int32_t* keep, int32_t* n_keep, float* margin,
                           int margin_stride, float score_weight, unsigned char* kflag_scratch) {
    PP_REQUIRE(max_n > 0 && max_n <= MAX_N, "nms: max_n=%d not in (0,%d]", max_n, MAX_N);
    PP_REQUIRE(!margin || kflag_scratch, "nms: the decision margin needs the kept-flag scratch");
    int32_t* order = static_cast<int32_t*>(scratch);
    unsigned long long* mask = reinterpret_cast<unsigned long long*>(
        static_cast<char*>(scratch) + ScratchCursor::align((size_t)n_frames * max_n * 4));
    int rc = run_nms<float, 0>(s, boxes, scores, n, max_n, n_frames, thr, order, mask, keep, n_keep, margin ? kflag_scratch : nullptr);
    if (rc != PP_OK || !margin) return rc;
    hipLaunchKernelGGL(fill_u32_strided_kernel, dim3((n_frames + 63) / 64), dim3(64), 0, s, reinterpret_cast<unsigned*>(margin),
                       margin_stride, n_frames, 0x7f800000u);
    hipLaunchKernelGGL(nms_margin_kernel, dim3((max_n + 255) / 256, n_frames), dim3(256), 0, s, boxes, scores, order, kflag_scratch, n,
                       max_n, thr, score_weight, reinterpret_cast<unsigned*>(margin), margin_stride);
    PP_HIP_CHECK(hipGetLastError());
    return PP_OK;
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
