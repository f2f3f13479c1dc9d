// Device-side pre/post-processing of the Faster-RCNN detector, batched over frames, no host round trips.
//
// Replaces, for the detection stage reached from pose_pipeline/wrappers/mmtrack.py:45
// (mmtrack `inference_mot` -> mmdet FasterRCNN.simple_test; model spec
// 3rdparty/mmtracking/_base_/models/faster_rcnn_r50_fpn.py:1-112, test pipeline
// 3rdparty/mmtracking/_base_/datasets/mot_challenge.py:3-4,33-47):
//   det_preprocess   mmcv imrescale (cv2.resize INTER_LINEAR, 8-bit fixed point) + imnormalize(to_rgb) + Pad(/32)
//   rpn_select       per level: sigmoid, top-1000 by score, AnchorGenerator grid anchor + delta2bbox
//   rpn_compact      concat levels, drop empty boxes, batched_nms coordinate offsets
//   (nms.hip)        NMS 0.7 -> top 1000 proposals;  NMS 0.5 -> top 100 detections
//   roi_align        SingleRoIExtractor.map_roi_levels + mmcv RoIAlign(7x7, aligned, adaptive sampling, avg)
//   final_decode     softmax, delta2bbox(.1,.1,.2,.2), /scale_factor, score > .05
// Every step follows oracle/detector.py line by line (same float32 operation order; sigmoid / exp /
// softmax evaluated in double and rounded once; sort ties break towards the lower index), so the
// integer outputs (selected indices, kept boxes) are bit-exact and the float boxes equal.
#include "pp_internal.h"
#include "det_internal.h"
#include "pp_amax.h"

namespace {

// ---- block-wide helpers (1024 threads = 16 waves) ---------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// exclusive prefix of a 0/1 flag over the block, in thread order; *total = sum.  s_w: >= 17 ints of LDS.
__device__ int block_rank(bool flag, int* s_w, int* total) {
    const unsigned long long b = __ballot(flag);
    const int r = __popcll(b & ((1ull << lane_id()) - 1ull));
    __syncthreads();
    if (lane_id() == 0) s_w[wave_id()] = __popcll(b);
    __syncthreads();
    int base = 0, tot = 0;
    const int nw = blockDim.x >> 6;
    for (int w = 0; w < nw; ++w) {
        const int c = s_w[w];
        if (w < wave_id()) base += c;
        tot += c;
    }
    *total = tot;
    return base + r;
}

__device__ __forceinline__ float sigmoid_f32(float x) { return (float)(1.0 / (1.0 + exp(-(double)x))); }
__device__ __forceinline__ float exp_f32(float x) { return (float)exp((double)x); }

// DeltaXYWHBBoxCoder.decode, means 0, clip_border False
__device__ __forceinline__ void delta2bbox(const float roi[4], const float d_in[4], const float stds[4], float out[4]) {
    const float dx = d_in[0] * stds[0], dy = d_in[1] * stds[1];
    float dw = d_in[2] * stds[2], dh = d_in[3] * stds[3];
    const float px = (roi[0] + roi[2]) * 0.5f, py = (roi[1] + roi[3]) * 0.5f;
    const float pw = roi[2] - roi[0], ph = roi[3] - roi[1];
    const float max_ratio = (float)4.135166556742356;     // |log(16/1000)|
    dw = fminf(fmaxf(dw, -max_ratio), max_ratio);
    dh = fminf(fmaxf(dh, -max_ratio), max_ratio);
    const float gx = px + pw * dx, gy = py + ph * dy;
    const float gw = pw * exp_f32(dw), gh = ph * exp_f32(dh);
    const float hw = gw * 0.5f, hh = gh * 0.5f;
    out[0] = gx - hw; out[1] = gy - hh; out[2] = gx + hw; out[3] = gy + hh;
}

// ---- det_preprocess -----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void det_preprocess_kernel(const uint8_t* __restrict__ frames, int H, int W, int nh,
                                                             int nw, int Hp, int Wp, const int32_t* __restrict__ xtab,
                                                             const int32_t* __restrict__ ytab, const float* __restrict__ lut,
                                                             float pad_val, float* __restrict__ out, unsigned* __restrict__ amax) {
    // xtab: [nw][3] = (sx, a0, a1);  ytab: [nh][3] = (sy, b0, b1)   (cv::resize 8-bit linear tables, *2048)
    __shared__ float s_lut[768];
    __shared__ float s_red[4];
    float vmax = 0.f;             // amax: max |v| of this frame's tensor (pp_amax.h; the fp16-form stem scales by it), one atomic per workgroup
    for (int i = threadIdx.x; i < 768; i += blockDim.x) s_lut[i] = lut[i];
    __syncthreads();
    const int f = blockIdx.y;
    const uint8_t* img = frames + (size_t)f * H * W * 3;
    float* o = out + (size_t)f * Hp * Wp * 4;
    const int total = Hp * Wp;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < total; p += gridDim.x * blockDim.x) {
        const int y = p / Wp, x = p - y * Wp;
        float4 v = make_float4(pad_val, pad_val, pad_val, 0.f);       // mmdet Pad(pad_val): 0 for Faster-RCNN, 114 for YOLOX
        if (y < nh && x < nw) {
            const int sx = xtab[3 * x], a0 = xtab[3 * x + 1], a1 = xtab[3 * x + 2];
            const int sy = ytab[3 * y], b0 = ytab[3 * y + 1], b1 = ytab[3 * y + 2];
            const int sx1 = min(sx + 1, W - 1), sy1 = min(sy + 1, H - 1);
            const uint8_t* r0 = img + (size_t)sy * W * 3;
            const uint8_t* r1 = img + (size_t)sy1 * W * 3;
            uint8_t c3[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int h0 = r0[sx * 3 + c] * a0 + r0[sx1 * 3 + c] * a1;
                const int h1 = r1[sx * 3 + c] * a0 + r1[sx1 * 3 + c] * a1;
                int val = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
                c3[c] = (uint8_t)min(max(val, 0), 255);
            }
            // to_rgb swap of the (already swapped by the wrapper) frame: tensor channel c <- frame channel 2-c
            // of the wrapper's RGB array == channel c of the decoded BGR frame; `frames` here IS the BGR frame.
            v.x = s_lut[0 * 256 + c3[0]];
            v.y = s_lut[1 * 256 + c3[1]];
            v.z = s_lut[2 * 256 + c3[2]];
        }
        *reinterpret_cast<float4*>(o + (size_t)p * 4) = v;
        vmax = fmaxf(vmax, pp_abs4max(v));
    }
    if (amax) {
        vmax = pp_wave_max(vmax);
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = vmax;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float r = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
            if (r > 0.f) atomicMax(amax + f, __float_as_uint(r));
        }
    }
}

// ---- rpn_select: one block per (level, frame) --------------------------------------------------------------
struct RpnLevels {
    const float* cls[5];      // [frame][H][W][3], or channels 0 - 2 of a 16-channel map (pitch 16)
    const float* reg[5];      // [frame][H][W][12], or channels 3 - 14 of it
    int pitch;
    int h[5], w[5], stride[5];
    float base[5][3][4];      // base anchors
};

// cut_gap (may be null; decision margins, pp_detector_enable_margins): per (frame, level) the score gap between the LAST anchor the
// top-k cut takes and the FIRST it leaves out -- +inf when the level has no cut (N <= nms_pre), 0 when the cut falls inside a tie
__global__ __launch_bounds__(1024) void rpn_select_kernel(RpnLevels L, int nms_pre, float* __restrict__ score_scratch,
                                                          int scratch_stride, float* __restrict__ cand_box,
                                                          float* __restrict__ cand_score, int32_t* __restrict__ cand_cnt,
                                                          float* __restrict__ cut_gap) {
    // outputs per frame: cand_box [5*nms_pre][4], cand_score [5*nms_pre], cand_cnt [5]
    const int lvl = blockIdx.x, f = blockIdx.y;
    const int N = L.h[lvl] * L.w[lvl] * 3;
    // anchor i = (pixel i / 3, anchor i % 3): logit at cls[i], deltas at reg[4 i ..] of the two-map layout; pixel * 16 + a and
    // pixel * 16 + 3 + 4 a of the fused map
    const bool fused = L.pitch == 16;
    const float* cls = L.cls[lvl] + (size_t)f * (fused ? (size_t)(N / 3) * 16 : (size_t)N);
    const float* reg = L.reg[lvl] + (size_t)f * (fused ? (size_t)(N / 3) * 16 : (size_t)N * 4);
    int lvl_off = 0;
    for (int l = 0; l < lvl; ++l) lvl_off += L.h[l] * L.w[l] * 3;
    float* sc = score_scratch + (size_t)f * scratch_stride + lvl_off;
    __shared__ int s_hist[2048];
    __shared__ int s_w[17];
    __shared__ float s_key[1024];
    __shared__ int s_idx[1024];
    __shared__ int s_sel_cnt;
    __shared__ unsigned s_prefix;
    __shared__ int s_need;
    __shared__ float s_below[16];

    for (int i = threadIdx.x; i < N; i += blockDim.x) sc[i] = sigmoid_f32(fused ? cls[(i / 3) * 16 + i % 3] : cls[i]);
    __syncthreads();
    const int k = min(nms_pre, N);
    int n_sel;
    if (N <= nms_pre) {
        // no top-k: natural index order (RPNHead only sorts when nms_pre < N)
        for (int i = threadIdx.x; i < 1024; i += blockDim.x) {
            s_key[i] = i < N ? sc[i] : -1.f;
            s_idx[i] = i < N ? i : 0x7fffffff;
        }
        n_sel = N;
        if (cut_gap && threadIdx.x == 0) cut_gap[f * 5 + lvl] = INFINITY;
        __syncthreads();
    } else {
        // radix select of the k-th largest score (scores >= 0: uint order == float order)
        unsigned prefix = 0;
        int need = k;            // how many still to take from the current prefix bucket
        const int shifts[3] = {21, 10, 0};
        const int bits[3] = {11, 11, 10};
        for (int pass = 0; pass < 3; ++pass) {
            for (int i = threadIdx.x; i < 2048; i += blockDim.x) s_hist[i] = 0;
            __syncthreads();
            const int sh = shifts[pass], nb = 1 << bits[pass];
            const unsigned hi_mask = pass == 0 ? 0u : (0xffffffffu << (sh + bits[pass]));
            for (int i = threadIdx.x; i < N; i += blockDim.x) {
                const unsigned u = __float_as_uint(sc[i]);
                if ((u & hi_mask) == (prefix & hi_mask)) atomicAdd(&s_hist[(u >> sh) & (nb - 1)], 1);
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                int acc = 0, b = nb - 1;
                for (; b >= 0; --b) {
                    if (acc + s_hist[b] >= need) break;
                    acc += s_hist[b];
                }
                s_prefix = prefix | ((unsigned)b << sh);
                s_need = need - acc;
            }
            __syncthreads();
            prefix = s_prefix;
            need = s_need;
            __syncthreads();
        }
        const float T = __uint_as_float(prefix);     // k-th largest value; `need` ties (lowest indices) are taken
        if (threadIdx.x == 0) s_sel_cnt = 0;
        __syncthreads();
        int eq_seen = 0;
        float below = -1.f;          // largest score under the cut value (scores are >= 0)
        for (int base = 0; base < N; base += blockDim.x) {
            const int i = base + threadIdx.x;
            const float v = i < N ? sc[i] : -1.f;
            const bool gt = v > T;
            const bool eq = (i < N) && (v == T);
            if (v < T) below = fmaxf(below, v);
            int tot;
            const int r = block_rank(eq, s_w, &tot);
            const bool take = gt || (eq && (eq_seen + r) < need);
            if (take) {
                const int pos = atomicAdd(&s_sel_cnt, 1);
                s_key[pos] = v;
                s_idx[pos] = i;
            }
            eq_seen += tot;
        }
        if (cut_gap) {
            for (int off = 32; off > 0; off >>= 1) below = fmaxf(below, __shfl_down(below, off, 64));
            if (lane_id() == 0) s_below[wave_id()] = below;
            __syncthreads();
            if (threadIdx.x == 0) {
                float b = s_below[0];
                for (int w = 1; w < (int)(blockDim.x >> 6); ++w) b = fmaxf(b, s_below[w]);
                // eq_seen = all anchors AT the cut value, `need` of them were taken
                cut_gap[f * 5 + lvl] = eq_seen > need ? 0.f : (b < 0.f ? INFINITY : T - b);
            }
        }
        __syncthreads();
        n_sel = s_sel_cnt;          // == k
        for (int i = n_sel + threadIdx.x; i < 1024; i += blockDim.x) {
            s_key[i] = -1.f;
            s_idx[i] = 0x7fffffff;
        }
        __syncthreads();
        // bitonic sort by (score desc, index asc)
        for (int kk = 2; kk <= 1024; kk <<= 1) {
            for (int j = kk >> 1; j > 0; j >>= 1) {
                const int i = threadIdx.x, l = i ^ j;
                if (l > i) {
                    const bool up = (i & kk) == 0;
                    const float ka = s_key[i], kb = s_key[l];
                    const int ia = s_idx[i], ib = s_idx[l];
                    const bool a_first = ka > kb || (ka == kb && ia < ib);
                    if (a_first != up) {
                        s_key[i] = kb; s_key[l] = ka;
                        s_idx[i] = ib; s_idx[l] = ia;
                    }
                }
                __syncthreads();
            }
        }
    }
    // decode the selected anchors
    float* ob = cand_box + ((size_t)f * 5 + lvl) * nms_pre * 4;
    float* os = cand_score + ((size_t)f * 5 + lvl) * nms_pre;
    const int Wl = L.w[lvl];
    const float stride = (float)L.stride[lvl];
    const float ones[4] = {1.f, 1.f, 1.f, 1.f};
    for (int r = threadIdx.x; r < n_sel; r += blockDim.x) {
        const int idx = s_idx[r];
        const int a = idx % 3, pos = idx / 3;
        const int x = pos % Wl, y = pos / Wl;
        const float sx = (float)x * stride, sy = (float)y * stride;
        float anchor[4] = {L.base[lvl][a][0] + sx, L.base[lvl][a][1] + sy, L.base[lvl][a][2] + sx, L.base[lvl][a][3] + sy};
        const float* d = fused ? reg + (size_t)(idx / 3) * 16 + (idx % 3) * 4 : reg + (size_t)idx * 4;
        float dd[4] = {d[0], d[1], d[2], d[3]};
        float box[4];
        delta2bbox(anchor, dd, ones, box);
        ob[4 * r] = box[0]; ob[4 * r + 1] = box[1]; ob[4 * r + 2] = box[2]; ob[4 * r + 3] = box[3];
        os[r] = s_key[r];
    }
    if (threadIdx.x == 0) cand_cnt[f * 5 + lvl] = n_sel;
}

// ---- rpn_compact: concat levels, drop empty boxes, add batched_nms offsets --------------------------------
__global__ __launch_bounds__(1024) void rpn_compact_kernel(const float* __restrict__ cand_box, const float* __restrict__ cand_score,
                                                           const int32_t* __restrict__ cand_cnt, int nms_pre, int max_n,
                                                           float* __restrict__ boxes, float* __restrict__ boxes_nms,
                                                           float* __restrict__ scores, int32_t* __restrict__ n_out) {
    const int f = blockIdx.x;
    __shared__ int s_w[17];
    __shared__ float s_max[16];
    const float* cb = cand_box + (size_t)f * 5 * nms_pre * 4;
    const float* cs = cand_score + (size_t)f * 5 * nms_pre;
    float* ob = boxes + (size_t)f * max_n * 4;
    float* obn = boxes_nms + (size_t)f * max_n * 4;
    float* os = scores + (size_t)f * max_n;
    // pass 1: max coordinate over valid boxes
    float mx = -INFINITY;
    for (int l = 0; l < 5; ++l) {
        const int c = cand_cnt[f * 5 + l];
        for (int r = threadIdx.x; r < c; r += blockDim.x) {
            const float* b = cb + ((size_t)l * nms_pre + r) * 4;
            if ((b[2] - b[0]) > 0.f && (b[3] - b[1]) > 0.f) mx = fmaxf(fmaxf(fmaxf(mx, b[0]), fmaxf(b[1], b[2])), b[3]);
        }
    }
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_down(mx, off, 64));
    if (lane_id() == 0) s_max[wave_id()] = mx;
    __syncthreads();
    mx = s_max[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) mx = fmaxf(mx, s_max[w]);
    const float step = mx + 1.0f;
    // pass 2: ordered compaction
    int written = 0;
    for (int l = 0; l < 5; ++l) {
        const int c = cand_cnt[f * 5 + l];
        const float offset = (float)l * step;
        for (int base = 0; base < c; base += blockDim.x) {
            const int r = base + threadIdx.x;
            bool ok = false;
            float b[4] = {0, 0, 0, 0};
            if (r < c) {
                const float* p = cb + ((size_t)l * nms_pre + r) * 4;
                b[0] = p[0]; b[1] = p[1]; b[2] = p[2]; b[3] = p[3];
                ok = (b[2] - b[0]) > 0.f && (b[3] - b[1]) > 0.f;
            }
            int tot;
            const int pos = written + block_rank(ok, s_w, &tot);
            if (ok) {
                ob[4 * pos] = b[0]; ob[4 * pos + 1] = b[1]; ob[4 * pos + 2] = b[2]; ob[4 * pos + 3] = b[3];
                obn[4 * pos] = b[0] + offset; obn[4 * pos + 1] = b[1] + offset;
                obn[4 * pos + 2] = b[2] + offset; obn[4 * pos + 3] = b[3] + offset;
                os[pos] = cs[(size_t)l * nms_pre + r];
            }
            written += tot;
        }
    }
    if (threadIdx.x == 0) n_out[f] = written;
}

// ---- gather the first `limit` kept boxes ---------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_kept_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                          int max_n, const int32_t* __restrict__ keep,
                                                          const int32_t* __restrict__ n_keep, int limit,
                                                          float* __restrict__ out_box, float* __restrict__ out_score,
                                                          int32_t* __restrict__ n_out, int out5, float* __restrict__ top_gap,
                                                          float* __restrict__ order_gap, int margin_stride) {
    // out5 == 0: out_box [f][limit][4] + out_score [f][limit];  out5 == 1: out_box [f][limit][5] (score in col 4)
    // top_gap / order_gap (may be null; decision margins): score gap across the `limit` cut of the kept list (+inf: no cut), smallest
    // gap between neighbours of the rows that are output (+inf: fewer than two)
    const int f = blockIdx.x;
    const int n = min(n_keep[f], limit);
    if (top_gap && threadIdx.x == 0) {
        const int32_t* kp = keep + (size_t)f * max_n;
        const float* sc = scores + (size_t)f * max_n;
        top_gap[(size_t)f * margin_stride] = n_keep[f] > limit ? sc[kp[limit - 1]] - sc[kp[limit]] : INFINITY;
    }
    if (order_gap && threadIdx.x == 64) {
        const int32_t* kp = keep + (size_t)f * max_n;
        const float* sc = scores + (size_t)f * max_n;
        float g = INFINITY;
        for (int r = 1; r < n; ++r) g = fminf(g, sc[kp[r - 1]] - sc[kp[r]]);
        order_gap[(size_t)f * margin_stride] = g;
    }
    for (int r = threadIdx.x; r < limit; r += blockDim.x) {
        float b[4] = {0, 0, 0, 0}, s = 0.f;
        if (r < n) {
            const int i = keep[(size_t)f * max_n + r];
            const float* p = boxes + ((size_t)f * max_n + i) * 4;
            b[0] = p[0]; b[1] = p[1]; b[2] = p[2]; b[3] = p[3];
            s = scores[(size_t)f * max_n + i];
        }
        if (out5) {
            float* o = out_box + ((size_t)f * limit + r) * 5;
            o[0] = b[0]; o[1] = b[1]; o[2] = b[2]; o[3] = b[3]; o[4] = s;
        } else {
            float* o = out_box + ((size_t)f * limit + r) * 4;
            o[0] = b[0]; o[1] = b[1]; o[2] = b[2]; o[3] = b[3];
            out_score[(size_t)f * limit + r] = s;
        }
    }
    if (threadIdx.x == 0) n_out[f] = n;
}

// ---- RoIAlign ---------------------------------------------------------------------------------------------------
struct FpnLevels {
    const float* feat[4];   // [frame][H][W][C]
    int h[4], w[4], stride[4];
};

__device__ __forceinline__ float bilinear(const float* feat, int H, int W, int C, int c, float y, float x) {
    if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) return 0.f;
    if (y <= 0.f) y = 0.f;
    if (x <= 0.f) x = 0.f;
    int y_low = (int)y, x_low = (int)x, y_high, x_high;
    if (y_low >= H - 1) { y_high = y_low = H - 1; y = (float)y_low; } else y_high = y_low + 1;
    if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else x_high = x_low + 1;
    const float ly = y - (float)y_low, lx = x - (float)x_low, hy = 1.f - ly, hx = 1.f - lx;
    const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
    const float v1 = feat[((size_t)y_low * W + x_low) * C + c], v2 = feat[((size_t)y_low * W + x_high) * C + c];
    const float v3 = feat[((size_t)y_high * W + x_low) * C + c], v4 = feat[((size_t)y_high * W + x_high) * C + c];
    return ((w1 * v1 + w2 * v2) + w3 * v3) + w4 * v4;
}

// 4 channels per lane (16-byte loads), one wave per bin: a 256-thread block walks the 49 bins of one RoI four
// at a time.  Same (iy, ix) accumulation order per output as the scalar statement -> identical results.
__device__ __forceinline__ float4 bilinear4(const float* feat, int H, int W, int C, int c, float y, float x) {
    if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) return make_float4(0.f, 0.f, 0.f, 0.f);
    if (y <= 0.f) y = 0.f;
    if (x <= 0.f) x = 0.f;
    int y_low = (int)y, x_low = (int)x, y_high, x_high;
    if (y_low >= H - 1) { y_high = y_low = H - 1; y = (float)y_low; } else y_high = y_low + 1;
    if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else x_high = x_low + 1;
    const float ly = y - (float)y_low, lx = x - (float)x_low, hy = 1.f - ly, hx = 1.f - lx;
    const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
    const float4 v1 = *reinterpret_cast<const float4*>(feat + ((size_t)y_low * W + x_low) * C + c);
    const float4 v2 = *reinterpret_cast<const float4*>(feat + ((size_t)y_low * W + x_high) * C + c);
    const float4 v3 = *reinterpret_cast<const float4*>(feat + ((size_t)y_high * W + x_low) * C + c);
    const float4 v4 = *reinterpret_cast<const float4*>(feat + ((size_t)y_high * W + x_high) * C + c);
    float4 o;
    o.x = ((w1 * v1.x + w2 * v2.x) + w3 * v3.x) + w4 * v4.x;
    o.y = ((w1 * v1.y + w2 * v2.y) + w3 * v3.y) + w4 * v4.y;
    o.z = ((w1 * v1.z + w2 * v2.z) + w3 * v3.z) + w4 * v4.z;
    o.w = ((w1 * v1.w + w2 * v2.w) + w3 * v3.w) + w4 * v4.w;
    return o;
}

__device__ __forceinline__ int roi_level(float x1, float y1, float x2, float y2) {
    const float scale = sqrtf((x2 - x1) * (y2 - y1));
    const int lvl = (int)floor(log2((double)(scale / PP_DET_FINEST_SCALE + 1e-6f)));
    return min(max(lvl, 0), 3);
}

// (Round 2 measured a locality order for the RoIs -- sorted by FPN level and Morton code of their centre, dealt out to the
// XCDs in contiguous eighths or in runs of 25: 6.1 - 7.1 ms against 5.5 ms for the score order NMS leaves them in.  L2 hit
// rate was not what bounds this kernel: with 58 % hits it already moves 63 GB through the L2s per launch (11 TB/s), and
// concentrating the requests on one region makes them collide on the same L2 channels.  profiles/r02_roi_pmc_before.txt.)
__global__ __launch_bounds__(256) void roi_align_kernel(FpnLevels L, int C, const float* __restrict__ rois,
                                                        const int32_t* __restrict__ n_rois, int max_rois, float* __restrict__ out) {
    // grid (max_rois, frames), block 256 = 4 waves x (C/4 = 64 lanes); out [frame*max_rois + r][7][7][C]
    const int f = blockIdx.y;
    const int r = blockIdx.x;
    float* o = out + ((size_t)f * max_rois + r) * 49 * C;
    if (r >= n_rois[f]) {
        for (int i = threadIdx.x; i < 49 * C / 4; i += blockDim.x)
            reinterpret_cast<float4*>(o)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const int c = (threadIdx.x & 63) * 4;
    const int wv = threadIdx.x >> 6;
    const float* roi = rois + ((size_t)f * max_rois + r) * 4;
    const float rx1 = roi[0], ry1 = roi[1], rx2 = roi[2], ry2 = roi[3];
    const int lvl = roi_level(rx1, ry1, rx2, ry2);
    const float ss = 1.0f / (float)L.stride[lvl];
    const int H = L.h[lvl], W = L.w[lvl];
    const float* feat = L.feat[lvl] + (size_t)f * H * W * C + c;
    const float x1 = rx1 * ss - 0.5f, y1 = ry1 * ss - 0.5f, x2 = rx2 * ss - 0.5f, y2 = ry2 * ss - 0.5f;
    const float rw = x2 - x1, rh = y2 - y1;
    const float bw = rw / 7.f, bh = rh / 7.f;
    const int gh = (int)ceilf(rh / 7.f), gw = (int)ceilf(rw / 7.f);
    const float count = (float)max(gh * gw, 1);
    for (int bin = wv; bin < 49; bin += 4) {
        const int ph = bin / 7, pw = bin - ph * 7;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        // (keeping the previous sample's two tap columns in registers -- consecutive samples are <= 1 px apart -- was measured
        // twice, rounds 1 and 2: the selects cost more than the loads they save, 5.5 -> 7.3 ms; the loads hit L1 / L2 anyway)
        for (int iy = 0; iy < gh; ++iy) {
            const float y = (y1 + (float)ph * bh) + (((float)iy + 0.5f) * bh) / (float)gh;
            for (int ix = 0; ix < gw; ++ix) {
                const float x = (x1 + (float)pw * bw) + (((float)ix + 0.5f) * bw) / (float)gw;
                const float4 v = bilinear4(feat, H, W, C, 0, y, x);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
        *reinterpret_cast<float4*>(o + bin * C + c) = make_float4(acc.x / count, acc.y / count, acc.z / count, acc.w / count);
    }
}

// ---- RoIAlign of the DEFAULT numerics: the same average of bilinear samples, evaluated separably --------------------------------
// profiles/r02_roi_pmc.txt: roi_align_kernel is VALU-bound (5.8e9 vector instructions per 64-frame launch = 9.5 ms of the SIMDs'
// time; ~200 instructions per sample and wave for 4 taps of address arithmetic and 16 multiply-adds), not memory-bound.
// A sample's four weights are products hy*hx, hy*lx, ly*hx, ly*lx and the sample grid of a bin is a tensor product (y depends on
// iy only, x on ix only), so
//     out[ph][pw][c] = 1/count * sum_iy sum_ix bilinear(y_iy, x_ix)[c] = sum_Y Wy[ph][Y] * ( sum_X Wx[pw][X] * F[Y][X][c] )
// with Wx[pw][X] = sum over the bin's valid x samples of (hx if x_low == X) + (lx if x_high == X), Wy likewise.  The row sums
// T[Y][pw] are computed once per feature row -- every pixel of the RoI's window is loaded ONCE -- and shared by all y samples
// and bins that touch the row.  RoIs far larger than the map (proposals are not clipped: SURVEY.md A5) cost what the map costs.
// Same sample positions, validity rule and clamping as mmcv's kernel; the float32 sum is taken in another ORDER, so this form
// belongs to the default numerics (tolerance-based parity, tests/test_gpu_split.py); programs created PP_NET_NUMERICS_EXACT
// keep roi_align_kernel, whose (iy, ix) accumulation order is the oracle's.
// One wave per RoI, lane = 4 channels (C = 256).  Wx is a dense table in LDS ([column of the window][8 floats], filled by the
// lanes in parallel, read back as broadcasts); the two most recent rows' T stay in registers while the y samples sweep down.
__device__ __forceinline__ float4 fma4(float w, const float4 v, const float4 a) {
    return make_float4(fmaf(w, v.x, a.x), fmaf(w, v.y, a.y), fmaf(w, v.z, a.z), fmaf(w, v.w, a.w));
}

constexpr int ROI_MAXW = 288;     // widest FPN map the table holds (level 0 of a 640 x 1088 input: 272 columns)

// amax (may be null): per RoI, the bit pattern of max |out| -- what the RoI head's fp16-form fc6 scales its input by
// (pp_net_input_amax; a wave owns its RoI, so the maximum is a plain store, no atomic and no zeroing)
__global__ __launch_bounds__(256) void roi_align_sep_kernel(FpnLevels L, int C, const float* __restrict__ rois,
                                                            const int32_t* __restrict__ n_rois, int max_rois, float* __restrict__ out,
                                                            unsigned* __restrict__ amax) {
    __shared__ __attribute__((aligned(16))) float s_wx[4][ROI_MAXW][8];
    const int f = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = blockIdx.x * 4 + wv;
    if (r >= max_rois) return;
    float* o = out + ((size_t)f * max_rois + r) * 49 * C + lane * 4;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float ymax = 0.f;
    auto put_amax = [&]() {
        if (!amax) return;
        const float m = pp_wave_max(ymax);
        if (lane == 0) amax[(size_t)f * max_rois + r] = __float_as_uint(m);
    };
    if (r >= n_rois[f]) {
#pragma unroll 7
        for (int b = 0; b < 49; ++b) *reinterpret_cast<float4*>(o + b * C) = zero4;
        put_amax();
        return;
    }
    const float* roi = rois + ((size_t)f * max_rois + r) * 4;
    const float rx1 = roi[0], ry1 = roi[1], rx2 = roi[2], ry2 = roi[3];
    const int lvl = roi_level(rx1, ry1, rx2, ry2);
    const float ss = 1.0f / (float)L.stride[lvl];
    const int H = L.h[lvl], W = L.w[lvl];
    const float* feat = L.feat[lvl] + (size_t)f * H * W * C + lane * 4;
    const float x1 = rx1 * ss - 0.5f, y1 = ry1 * ss - 0.5f, x2 = rx2 * ss - 0.5f, y2 = ry2 * ss - 0.5f;
    const float rw = x2 - x1, rh = y2 - y1;
    const float bw = rw / 7.f, bh = rh / 7.f;
    const int gh = __builtin_amdgcn_readfirstlane((int)ceilf(rh / 7.f)), gw = __builtin_amdgcn_readfirstlane((int)ceilf(rw / 7.f));
    const float count = (float)max(gh * gw, 1);
    // columns any valid x sample can touch, with one column of slack on either side (zero weights there)
    const float xs_first = x1 + (0.5f * bw) / (float)max(gw, 1), xs_last = (x1 + 6.f * bw) + (((float)gw - 0.5f) * bw) / (float)max(gw, 1);
    const bool empty = gw < 1 || gh < 1 || !(xs_last >= -1.0f) || !(xs_first <= (float)W) || W > ROI_MAXW;
    if (empty) {
        if (gw >= 1 && gh >= 1 && W > ROI_MAXW) {
            // map wider than the weight table: the sample loop (one wave for all 49 bins); not reached with a 640 x 1088 input
            for (int bin = 0; bin < 49; ++bin) {
                const int ph = bin / 7, pw = bin - ph * 7;
                float4 acc = zero4;
                for (int iy = 0; iy < gh; ++iy) {
                    const float y = (y1 + (float)ph * bh) + (((float)iy + 0.5f) * bh) / (float)gh;
                    for (int ix = 0; ix < gw; ++ix) {
                        const float x = (x1 + (float)pw * bw) + (((float)ix + 0.5f) * bw) / (float)gw;
                        const float4 v = bilinear4(feat, H, W, C, 0, y, x);
                        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
                    }
                }
                const float4 ov = make_float4(acc.x / count, acc.y / count, acc.z / count, acc.w / count);
                *reinterpret_cast<float4*>(o + bin * C) = ov;
                ymax = fmaxf(ymax, pp_abs4max(ov));
            }
            put_amax();
            return;
        }
#pragma unroll 7
        for (int b = 0; b < 49; ++b) *reinterpret_cast<float4*>(o + b * C) = zero4;        // no sample inside the map: mmcv's sum is 0
        put_amax();
        return;
    }
    const int Xmin = __builtin_amdgcn_readfirstlane(min(max((int)floorf(fmaxf(xs_first, -1.0f)) - 1, 0), W - 1));
    const int Xmax = __builtin_amdgcn_readfirstlane(min(max((int)floorf(fminf(xs_last, (float)W)) + 2, 0), W - 1));
    const int Wn = Xmax - Xmin + 1;
    // ---- Wx[pw][X], X = Xmin + column: lane-parallel over the columns ----------------------------------------------------------
    float (*wxt)[8] = s_wx[wv];
    for (int c0 = 0; c0 < Wn; c0 += 64) {
        const int col = c0 + lane;
        const int X = Xmin + col;
        float w[8];
#pragma unroll
        for (int pw = 0; pw < 7; ++pw) {
            float acc = 0.f;
            // the bin's samples lie in [xf, xf + bw): columns floor(xf) .. floor(xf + bw) + 1 (after clamping into the map)
            const float xf = x1 + (float)pw * bw;
            if ((float)X >= xf - 2.0f && (float)X <= xf + bw + 2.0f) {
                for (int ix = 0; ix < gw; ++ix) {
                    float x = (x1 + (float)pw * bw) + (((float)ix + 0.5f) * bw) / (float)gw;
                    if (x < -1.0f || x > (float)W) continue;
                    if (x <= 0.f) x = 0.f;
                    int xl = (int)x, xh;
                    if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else xh = xl + 1;
                    const float lx = x - (float)xl, hx = 1.f - lx;
                    if (xl == X) acc += hx;
                    if (xh == X) acc += lx;
                }
            }
            w[pw] = acc;
        }
        w[7] = 0.f;
        if (col < Wn) {
            *reinterpret_cast<float4*>(&wxt[col][0]) = make_float4(w[0], w[1], w[2], w[3]);
            *reinterpret_cast<float4*>(&wxt[col][4]) = make_float4(w[4], w[5], w[6], w[7]);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the table is private to this wave: LDS writes before the reads below
    __builtin_amdgcn_wave_barrier();
    // ---- rows: T[pw] = sum_X Wx[pw][X] F[row][X]; the two most recent rows stay in registers -----------------------------------
    float4 T0[7], T1[7];
    int R0 = -1000, R1 = -1000;
    auto load_row = [&](int row) {          // T1 <- T0, T0 <- row
#pragma unroll
        for (int pw = 0; pw < 7; ++pw) T1[pw] = T0[pw];
        R1 = R0;
#pragma unroll
        for (int pw = 0; pw < 7; ++pw) T0[pw] = zero4;
        const float* frow = feat + ((size_t)row * W + Xmin) * C;
        int c = 0;
        for (; c + 4 <= Wn; c += 4) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(frow + (size_t)(c + u) * C);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float4 wa = *reinterpret_cast<const float4*>(&wxt[c + u][0]), wb = *reinterpret_cast<const float4*>(&wxt[c + u][4]);
                T0[0] = fma4(wa.x, v[u], T0[0]); T0[1] = fma4(wa.y, v[u], T0[1]); T0[2] = fma4(wa.z, v[u], T0[2]); T0[3] = fma4(wa.w, v[u], T0[3]);
                T0[4] = fma4(wb.x, v[u], T0[4]); T0[5] = fma4(wb.y, v[u], T0[5]); T0[6] = fma4(wb.z, v[u], T0[6]);
            }
        }
        for (; c < Wn; ++c) {
            const float4 v = *reinterpret_cast<const float4*>(frow + (size_t)c * C);
            const float4 wa = *reinterpret_cast<const float4*>(&wxt[c][0]), wb = *reinterpret_cast<const float4*>(&wxt[c][4]);
            T0[0] = fma4(wa.x, v, T0[0]); T0[1] = fma4(wa.y, v, T0[1]); T0[2] = fma4(wa.z, v, T0[2]); T0[3] = fma4(wa.w, v, T0[3]);
            T0[4] = fma4(wb.x, v, T0[4]); T0[5] = fma4(wb.y, v, T0[5]); T0[6] = fma4(wb.z, v, T0[6]);
        }
        R0 = row;
    };
    const float inv = 1.0f / count;
    for (int ph = 0; ph < 7; ++ph) {
        float4 acc[7];
#pragma unroll
        for (int pw = 0; pw < 7; ++pw) acc[pw] = zero4;
        // the bin's y samples lie in [y1 + ph bh, y1 + (ph + 1) bh): skip the part of the loop that is outside the map
        const float yb = y1 + (float)ph * bh;
        int iy0 = 0, iy1 = gh;
        if (bh > 0.f) {
            iy0 = __builtin_amdgcn_readfirstlane(min(max((int)floorf((-1.0f - yb) * (float)gh / bh - 0.5f) - 1, 0), gh));
            iy1 = __builtin_amdgcn_readfirstlane(min(max((int)ceilf(((float)H - yb) * (float)gh / bh - 0.5f) + 2, 0), gh));
        }
        for (int iy = iy0; iy < iy1; ++iy) {
            float y = (y1 + (float)ph * bh) + (((float)iy + 0.5f) * bh) / (float)gh;
            if (y < -1.0f || y > (float)H) continue;
            if (y <= 0.f) y = 0.f;
            int yl = (int)y, yh;
            if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else yh = yl + 1;
            const float ly = y - (float)yl, hy = 1.f - ly;
            yl = __builtin_amdgcn_readfirstlane(yl);
            yh = __builtin_amdgcn_readfirstlane(yh);
            if (yl != R0 && yl != R1) load_row(yl);
            if (yh != R0 && yh != R1) load_row(yh);
            if (yl != R0 && yl != R1) load_row(yl);            // (cannot happen for yh in {yl, yl + 1}; keeps the cache logic closed)
            const float wl = hy * inv, wh = ly * inv;
            if (yl == R0) {
#pragma unroll
                for (int pw = 0; pw < 7; ++pw) acc[pw] = fma4(wl, T0[pw], acc[pw]);
            } else {
#pragma unroll
                for (int pw = 0; pw < 7; ++pw) acc[pw] = fma4(wl, T1[pw], acc[pw]);
            }
            if (yh == R0) {
#pragma unroll
                for (int pw = 0; pw < 7; ++pw) acc[pw] = fma4(wh, T0[pw], acc[pw]);
            } else {
#pragma unroll
                for (int pw = 0; pw < 7; ++pw) acc[pw] = fma4(wh, T1[pw], acc[pw]);
            }
        }
#pragma unroll
        for (int pw = 0; pw < 7; ++pw) {
            *reinterpret_cast<float4*>(o + (ph * 7 + pw) * C) = acc[pw];
            ymax = fmaxf(ymax, pp_abs4max(acc[pw]));
        }
    }
    put_amax();
}

// scalar statement of the same kernel (one thread per channel), kept as the readable reference of the arithmetic
__global__ __launch_bounds__(256) void roi_align_kernel_scalar(FpnLevels L, int C, const float* __restrict__ rois,
                                                               const int32_t* __restrict__ n_rois, int max_rois,
                                                               float* __restrict__ out) {
    const int r = blockIdx.x, f = blockIdx.y, c = threadIdx.x;
    float* o = out + ((size_t)f * max_rois + r) * 49 * C;
    if (r >= n_rois[f]) {
        for (int b = 0; b < 49; ++b) o[b * C + c] = 0.f;
        return;
    }
    const float* roi = rois + ((size_t)f * max_rois + r) * 4;
    const float rx1 = roi[0], ry1 = roi[1], rx2 = roi[2], ry2 = roi[3];
    // SingleRoIExtractor.map_roi_levels (finest_scale 56)
    const float scale = sqrtf((rx2 - rx1) * (ry2 - ry1));
    int lvl = (int)floor(log2((double)(scale / PP_DET_FINEST_SCALE + 1e-6f)));
    lvl = min(max(lvl, 0), 3);
    const float ss = 1.0f / (float)L.stride[lvl];
    const int H = L.h[lvl], W = L.w[lvl];
    const float* feat = L.feat[lvl] + (size_t)f * H * W * C;
    const float x1 = rx1 * ss - 0.5f, y1 = ry1 * ss - 0.5f, x2 = rx2 * ss - 0.5f, y2 = ry2 * ss - 0.5f;
    const float rw = x2 - x1, rh = y2 - y1;
    const float bw = rw / 7.f, bh = rh / 7.f;
    const int gh = (int)ceilf(rh / 7.f), gw = (int)ceilf(rw / 7.f);
    const float count = (float)max(gh * gw, 1);
    for (int ph = 0; ph < 7; ++ph) {
        for (int pw = 0; pw < 7; ++pw) {
            float acc = 0.f;
            for (int iy = 0; iy < gh; ++iy) {
                const float y = (y1 + (float)ph * bh) + (((float)iy + 0.5f) * bh) / (float)gh;
                for (int ix = 0; ix < gw; ++ix) {
                    const float x = (x1 + (float)pw * bw) + (((float)ix + 0.5f) * bw) / (float)gw;
                    acc += bilinear(feat, H, W, C, c, y, x);
                }
            }
            o[(ph * 7 + pw) * C + c] = acc / count;
        }
    }
}

// ---- final_decode: softmax, delta2bbox, rescale, score threshold, ordered compaction ---------------------------
__global__ __launch_bounds__(1024) void final_decode_kernel(const float* __restrict__ rois, const int32_t* __restrict__ n_rois,
                                                            int max_rois, const float* __restrict__ cls,
                                                            const float* __restrict__ reg, float sfx, float sfy,
                                                            float score_thr, float* __restrict__ boxes,
                                                            float* __restrict__ scores, int32_t* __restrict__ n_out,
                                                            float* __restrict__ level_margin, float* __restrict__ thr_margin,
                                                            int margin_stride) {
    // level_margin / thr_margin (may be null; decision margins): smallest distance of a RoI's log2(scale / 56 + 1e-6) to a level
    // boundary (1, 2, 3) and smallest |score - score_thr| over the frame's RoIs
    const int f = blockIdx.x;
    __shared__ int s_w[17];
    __shared__ float s_m[2][16];
    const int n = n_rois[f];
    const float stds[4] = PP_DET_RCNN_STDS;
    int written = 0;
    float m_lvl = INFINITY, m_thr = INFINITY;
    for (int base = 0; base < n; base += blockDim.x) {
        const int r = base + threadIdx.x;
        bool ok = false;
        float b[4] = {0, 0, 0, 0}, s = 0.f;
        if (r < n) {
            const size_t g = (size_t)f * max_rois + r;
            const double z0 = cls[2 * g], z1 = cls[2 * g + 1];
            const double m = fmax(z0, z1);
            const double e0 = exp(z0 - m), e1 = exp(z1 - m);
            s = (float)(e0 / (e0 + e1));
            const float roi[4] = {rois[4 * g], rois[4 * g + 1], rois[4 * g + 2], rois[4 * g + 3]};
            const float d[4] = {reg[4 * g], reg[4 * g + 1], reg[4 * g + 2], reg[4 * g + 3]};
            delta2bbox(roi, d, stds, b);
            b[0] = b[0] / sfx; b[1] = b[1] / sfy; b[2] = b[2] / sfx; b[3] = b[3] / sfy;
            ok = s > score_thr;
            if (level_margin) {
                m_thr = fminf(m_thr, fabsf(s - score_thr));
                const float scale = sqrtf((roi[2] - roi[0]) * (roi[3] - roi[1]));
                const double v = log2((double)(scale / PP_DET_FINEST_SCALE + 1e-6f));
                m_lvl = fminf(m_lvl, (float)fmin(fmin(fabs(v - 1.0), fabs(v - 2.0)), fabs(v - 3.0)));
            }
        }
        int tot;
        const int pos = written + block_rank(ok, s_w, &tot);
        if (ok) {
            float* ob = boxes + ((size_t)f * max_rois + pos) * 4;
            ob[0] = b[0]; ob[1] = b[1]; ob[2] = b[2]; ob[3] = b[3];
            scores[(size_t)f * max_rois + pos] = s;
        }
        written += tot;
    }
    if (threadIdx.x == 0) n_out[f] = written;
    if (level_margin) {
        for (int off = 32; off > 0; off >>= 1) {
            m_lvl = fminf(m_lvl, __shfl_down(m_lvl, off, 64));
            m_thr = fminf(m_thr, __shfl_down(m_thr, off, 64));
        }
        if (lane_id() == 0) { s_m[0][wave_id()] = m_lvl; s_m[1][wave_id()] = m_thr; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < (int)(blockDim.x >> 6); ++w) { m_lvl = fminf(m_lvl, s_m[0][w]); m_thr = fminf(m_thr, s_m[1][w]); }
            level_margin[(size_t)f * margin_stride] = m_lvl;
            thr_margin[(size_t)f * margin_stride] = m_thr;
        }
    }
}

// min over (frame, level) of the per-level cut gaps -> one figure per frame
__global__ void rpn_cut_margin_kernel(const float* __restrict__ cut_gap, float* __restrict__ out, int margin_stride, int n_frames) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_frames) return;
    float g = INFINITY;
    for (int l = 0; l < 5; ++l) g = fminf(g, cut_gap[f * 5 + l]);
    out[(size_t)f * margin_stride] = g;
}

}  // namespace

// ---- launchers ------------------------------------------------------------------------------------------------------
int det_enqueue_preprocess(hipStream_t s, const uint8_t* frames, int n_frames, int H, int W, int nh, int nw, int Hp,
                           int Wp, const int32_t* xtab, const int32_t* ytab, const float* lut, float pad_val, float* out, unsigned* amax) {
    // with maxima: 128 workgroups per frame from 8 frames on (one same-address atomic each: ~0.6 us apiece at the memory side)
    dim3 grid(std::min((Hp * Wp + 255) / 256, amax && n_frames >= 8 ? 128 : 512), n_frames);
    hipLaunchKernelGGL(det_preprocess_kernel, grid, dim3(256), 0, s, frames, H, W, nh, nw, Hp, Wp, xtab, ytab, lut, pad_val, out, amax);
    PP_HIP_CHECK(hipGetLastError());
    return PP_OK;
}

int det_enqueue_rpn(hipStream_t s, const DetRpnArgs& a, int n_frames) {
    RpnLevels L;
    for (int l = 0; l < 5; ++l) {
        L.pitch = a.pitch;
        L.cls[l] = a.cls[l]; L.reg[l] = a.reg[l]; L.h[l] = a.h[l]; L.w[l] = a.w[l]; L.stride[l] = a.stride[l];
        memcpy(L.base[l], a.base[l], sizeof(L.base[l]));
    }
    hipLaunchKernelGGL(rpn_select_kernel, dim3(5, n_frames), dim3(1024), 0, s, L, a.nms_pre, a.score_scratch,
                       a.scratch_stride, a.cand_box, a.cand_score, a.cand_cnt, a.cut_gap);
    if (a.cut_gap)
        hipLaunchKernelGGL(rpn_cut_margin_kernel, dim3((n_frames + 63) / 64), dim3(64), 0, s, a.cut_gap, a.cut_margin, a.margin_stride, n_frames);
    hipLaunchKernelGGL(rpn_compact_kernel, dim3(n_frames), dim3(1024), 0, s, a.cand_box, a.cand_score, a.cand_cnt, a.nms_pre,
                       a.max_n, a.boxes, a.boxes_nms, a.scores, a.n_boxes);
    PP_HIP_CHECK(hipGetLastError());
    return PP_OK;
}

int det_enqueue_gather(hipStream_t s, const float* boxes, const float* scores, int max_n, const int32_t* keep,
                       const int32_t* n_keep, int limit, float* out_box, float* out_score, int32_t* n_out, int out5,
                       int n_frames, float* top_gap, float* order_gap, int margin_stride) {
    hipLaunchKernelGGL(gather_kept_kernel, dim3(n_frames), dim3(256), 0, s, boxes, scores, max_n, keep, n_keep, limit,
                       out_box, out_score, n_out, out5, top_gap, order_gap, margin_stride);
    PP_HIP_CHECK(hipGetLastError());
    return PP_OK;
}

// which kernel det_enqueue_roi_align launches for a program of these numerics (POSEPIPE_ROI_SEPARABLE: A/B knob, read once)
bool det_roi_align_separable(int separable) {
    static const int sep_env = getenv("POSEPIPE_ROI_SEPARABLE") ? atoi(getenv("POSEPIPE_ROI_SEPARABLE")) : -1;
    return sep_env >= 0 ? sep_env != 0 : separable != 0;
}

int det_enqueue_roi_align(hipStream_t s, const DetFpnArgs& a, const float* rois, const int32_t* n_rois, int max_rois,
                          float* out, int n_frames, int separable, unsigned* amax) {
    PP_REQUIRE(a.c == 256, "roi_align: C=%d (kernel launches one thread per channel, C must be 256)", a.c);
    FpnLevels L;
    for (int l = 0; l < 4; ++l) {
        L.feat[l] = a.feat[l]; L.h[l] = a.h[l]; L.w[l] = a.w[l]; L.stride[l] = a.stride[l];
    }
    const bool sep = det_roi_align_separable(separable);
    PP_REQUIRE(!amax || sep, "roi_align: only the separable kernel writes the per-RoI maxima");
    if (sep)
        hipLaunchKernelGGL(roi_align_sep_kernel, dim3((max_rois + 3) / 4, n_frames), dim3(256), 0, s, L, a.c, rois, n_rois, max_rois, out, amax);
    else
        hipLaunchKernelGGL(roi_align_kernel, dim3(max_rois, n_frames), dim3(a.c), 0, s, L, a.c, rois, n_rois, max_rois, out);
    PP_HIP_CHECK(hipGetLastError());
    return PP_OK;
}

int det_enqueue_final_decode(hipStream_t s, const float* rois, const int32_t* n_rois, int max_rois, const float* cls,
                             const float* reg, float sfx, float sfy, float score_thr, float* boxes, float* scores,
                             int32_t* n_out, int n_frames, float* level_margin, float* thr_margin, int margin_stride) {
    hipLaunchKernelGGL(final_decode_kernel, dim3(n_frames), dim3(1024), 0, s, rois, n_rois, max_rois, cls, reg, sfx, sfy,
                       score_thr, boxes, scores, n_out, level_margin, thr_margin, margin_stride);
    PP_HIP_CHECK(hipGetLastError());
    return PP_OK;
}
