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
#include <atomic>
#include <mutex>
#include "pp_internal.h"

std::atomic<int> g_decode_generic{-1};      // pp_debug_knob("decode_generic"): -1 = POSEPIPE_DECODE_GENERIC

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


// ---- round 4: the same decode at HBM speed --------------------------------------------------------------------------------------
// flip_merge_decode_kernel above spends its time in the blur, not in memory: every output of the two separable passes re-reads its
// 17 inputs from LDS behind a bounds test (2 x 17 ds_read_b32 + selects per pixel; 238 us per 64 persons = 0.25 TB/s).  Same
// arithmetic, same order of operations per output (row pass: g0 v0, then + g_j v_j for j = 1 .. ks-1; column pass: g_r c, then
// + g_{r+j} (down_j + up_j); no contraction) -- identical bits -- but organised for the hardware:
//   * the merged map sits in LDS with an 8-pixel zero (or reflected) frame, so no tap is ever tested, and ROW PAIRS interleaved
//     ([y / 2][x][y & 1]): a thread computes 8 consecutive outputs of TWO rows from 24 x 2 inputs held in registers
//     (12 ds_read_b128 for 16 outputs instead of 272 ds_read_b32) with packed float32 arithmetic (v_pk_mul_f32 / v_pk_add_f32: the two
//     rows are the two halves of every operand);
//   * the row pass writes its result COLUMN-PAIR interleaved ([x / 2][y][x & 1]) so that the column pass is the same kernel body
//     along y; the blurred map is never stored: the column pass keeps the running maximum (the rescale factor) and drops the 5 x 5
//     neighbourhood of the arg-max -- all the Taylor stencil reads -- into a 25-float table;
//   * loads: a thread requests all its lines of both maps (float4 straight, the mirrored map element-wise) before it consumes one.
// Maps with odd sizes, w % 4 != 0, sizes over 128 or blur kernels other than 11 / 17 keep the generic kernel.
typedef float v2f __attribute__((ext_vector_type(2)));

struct FastGeom {
    int wpa, hpb;            // padded pitches: A rows hold wpa pixel pairs, B columns hpb
    int load_tasks, c4;      // (row pair, 4-pixel chunk) tasks; chunks per row
    int row_tasks, nseg;     // (row pair, 8-pixel segment)
    int col_tasks, nsegy;    // (column pair, 8-row segment)
    float inv_c4, inv_nseg, inv_nsegy;
};

template <int KS>
__global__ __launch_bounds__(512) void flip_merge_decode_fast_kernel(DecodeArgs a, FastGeom g) {
    // 512 threads: the 432 row / column tasks of a 96 x 72 map are ONE round (with 256 threads: two, the second 69 % full, and the
    // load, row and column phases of a CU's two workgroups ran in lock-step instead of overlapping -- measured, n = 256: 152 us of
    // which loads 59, the two passes 63, frames 9); two workgroups per CU = 4 waves per SIMD
    constexpr int R = KS / 2, NTH = 512, LU = 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H = a.h, W = a.w;
    float* A = smem;                                            // [H / 2][wpa][2]
    float* B = A + (H / 2) * g.wpa * 2;                         // [W8 / 2][hpb][2]
    float* ST = B + (KS ? ((W + 7) / 8 * 4) * g.hpb * 2 : 0);   // 5 x 5 blurred values around the arg-max
    __shared__ float s_v[8];
    __shared__ int s_i[8];
    const int tid = threadIdx.x;
    const int k = blockIdx.x % a.k;
    const int n = blockIdx.x / a.k;
    const int HW = H * W;
    const float* src = a.hm + ((size_t)n * a.k + k) * HW;
    const float* fsrc = a.hm_flip ? a.hm_flip + ((size_t)n * a.k + a.flip_perm[k]) * HW : nullptr;
    float* mdst = a.merged ? a.merged + ((size_t)n * a.k + k) * HW : nullptr;

    // ---- 0. the frames: zero (the reflected frame of post 2 is filled from the merged map below) -----------------------------------
    if (KS) {
        // A: per row pair 8 pixel pairs left of the map and wpa - W - 8 right of it (16 B = 2 pixel pairs); B: 8 rows above, hpb - H - 8 below
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        const int aq = (g.wpa - W) / 2;                         // float4 units per row pair: 4 left + the rest right
        for (int yp = tid / 16; yp < H / 2; yp += NTH / 16) {
            const int q = tid & 15;
            if (q < aq) *reinterpret_cast<float4*>(A + ((yp * g.wpa) + (q < 4 ? 2 * q : 8 + W + 2 * (q - 4))) * 2) = z;
        }
        const int ncp = (W + 7) / 8 * 4, bq = (g.hpb - H) / 2;  // column pairs the row pass writes; units per column pair
        for (int xp = tid / 16; xp < ncp; xp += NTH / 16) {
            const int q = tid & 15;
            if (q < bq) *reinterpret_cast<float4*>(B + ((xp * g.hpb) + (q < 4 ? 2 * q : 8 + H + 2 * (q - 4))) * 2) = z;
        }
    }

    // ---- 1. flip-merge into LDS + arg-max; four tasks' loads in flight per thread ---------------------------------------------------
    float best = -INFINITY;
    int besti = 0x7fffffff;
    for (int t0 = tid; t0 < g.load_tasks; t0 += NTH * LU) {
        float4 s[LU][2];
        float f[LU][2][4];
        int yp_[LU], c_[LU];
#pragma unroll
        for (int u = 0; u < LU; ++u) {
            const int t = t0 + NTH * u;
            const int tc = t < g.load_tasks ? t : 0;                       // clamped: the loads of a task past the end are valid and ignored
            yp_[u] = (int)(((float)tc + 0.5f) * g.inv_c4);
            c_[u] = tc - yp_[u] * g.c4;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int y = 2 * yp_[u] + r;
                s[u][r] = *reinterpret_cast<const float4*>(src + y * W + 4 * c_[u]);
                if (fsrc) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int x = 4 * c_[u] + e;
                        // flipped back: column W-1-x'; shifted right by one: x' = max(x-1, 0)
                        const int xs = a.shift_heatmap ? (x > 0 ? x - 1 : 0) : x;
                        f[u][r][e] = fsrc[y * W + (W - 1 - xs)];
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < LU; ++u) {
            if (t0 + NTH * u >= g.load_tasks) continue;
            float v[2][4];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const float sv[4] = {s[u][r].x, s[u][r].y, s[u][r].z, s[u][r].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[r][e] = fsrc ? __fmul_rn(__fadd_rn(sv[e], f[u][r][e]), 0.5f) : sv[e];
                    argmax_combine(best, besti, v[r][e], (2 * yp_[u] + r) * W + 4 * c_[u] + e);
                }
                if (mdst) *reinterpret_cast<float4*>(mdst + (2 * yp_[u] + r) * W + 4 * c_[u]) = make_float4(v[r][0], v[r][1], v[r][2], v[r][3]);
            }
            float* d = A + ((yp_[u] * g.wpa) + 4 * c_[u] + 8) * 2;
            *reinterpret_cast<float4*>(d) = make_float4(v[0][0], v[1][0], v[0][1], v[1][1]);
            *reinterpret_cast<float4*>(d + 4) = make_float4(v[0][2], v[1][2], v[0][3], v[1][3]);
        }
    }
    block_argmax(best, besti, s_v, s_i);            // (its barriers also publish A)
    const float maxval = best;
    const int idx = besti;
    int px = idx % W, py = idx / W;
    const bool has_peak = maxval > 0.0f;
    float cxp = has_peak ? (float)px : -1.0f;       // mmpose: preds = -1 where maxval <= 0
    float cyp = has_peak ? (float)py : -1.0f;
    if (!has_peak) {
        px = -1;
        py = -1;
    }
    auto Aat = [&](int y, int x) -> float { return A[(((y >> 1) * g.wpa) + x + 8) * 2 + (y & 1)]; };

    if (KS && (a.post == 1 || a.post == 2)) {
        const bool reflect = a.post == 2;
        if (reflect) {
            // BORDER_REFLECT_101 frame of the merged map: A(y, -j) = A(y, j), A(y, W-1+j) = A(y, W-1-j), j = 1 .. R
            for (int i = tid; i < H * 2 * R; i += NTH) {
                const int y = i / (2 * R), r = i - y * (2 * R), j = (r >> 1) + 1;
                const int xd = (r & 1) ? W - 1 + j : -j, xs = (r & 1) ? W - 1 - j : j;
                A[(((y >> 1) * g.wpa) + xd + 8) * 2 + (y & 1)] = Aat(y, xs);
            }
            __syncthreads();
        }
        // ---- 2. row pass: 8 outputs x 2 rows per task ----------------------------------------------------------------------------------
        for (int t = tid; t < g.row_tasks; t += NTH) {
            const int yp = (int)(((float)t + 0.5f) * g.inv_nseg), seg = t - yp * g.nseg;
            const float4* in4 = reinterpret_cast<const float4*>(A + ((yp * g.wpa) + 8 * seg) * 2);
            v2f in[24];
#pragma unroll
            for (int q = 0; q < 12; ++q) {
                const float4 v = in4[q];
                in[2 * q] = v2f{v.x, v.y};
                in[2 * q + 1] = v2f{v.z, v.w};
            }
            v2f out[8];
#pragma unroll
            for (int xi = 0; xi < 8; ++xi) {
                v2f sacc = in[xi + 8 - R] * a.gk[0];
#pragma unroll
                for (int j = 1; j < KS; ++j) sacc = sacc + in[xi + j + 8 - R] * a.gk[j];
                out[xi] = sacc;
            }
            const int y0 = 2 * yp;
#pragma unroll
            for (int xq = 0; xq < 4; ++xq) {
                const int xpair = 4 * seg + xq;
                *reinterpret_cast<float4*>(B + ((xpair * g.hpb) + y0 + 8) * 2) = make_float4(out[2 * xq].x, out[2 * xq + 1].x, out[2 * xq].y, out[2 * xq + 1].y);
            }
        }
        __syncthreads();
        if (reflect) {
            const int ncp2 = W;                        // columns
            for (int i = tid; i < ncp2 * 2 * R; i += NTH) {
                const int x = i / (2 * R), r = i - x * (2 * R), j = (r >> 1) + 1;
                const int yd = (r & 1) ? H - 1 + j : -j, ys = (r & 1) ? H - 1 - j : j;
                B[(((x >> 1) * g.hpb) + yd + 8) * 2 + (x & 1)] = B[(((x >> 1) * g.hpb) + ys + 8) * 2 + (x & 1)];
            }
            __syncthreads();
        }
        // ---- 3. column pass: 8 outputs x 2 columns per task; running maximum + the 5 x 5 neighbourhood of the arg-max --------------------
        float bmax = -INFINITY;
        for (int t = tid; t < g.col_tasks; t += NTH) {
            const int xp = (int)(((float)t + 0.5f) * g.inv_nsegy), sy = t - xp * g.nsegy;
            const float4* in4 = reinterpret_cast<const float4*>(B + ((xp * g.hpb) + 8 * sy) * 2);
            v2f in[24];
#pragma unroll
            for (int q = 0; q < 12; ++q) {
                const float4 v = in4[q];
                in[2 * q] = v2f{v.x, v.y};
                in[2 * q + 1] = v2f{v.z, v.w};
            }
            const bool near_peak = has_peak && 2 * xp + 1 >= px - 2 && 2 * xp <= px + 2 && 8 * sy + 7 >= py - 2 && 8 * sy <= py + 2;
#pragma unroll
            for (int yi = 0; yi < 8; ++yi) {
                v2f sacc = in[yi + 8] * a.gk[R];
#pragma unroll
                for (int j = 1; j <= R; ++j) sacc = sacc + (in[yi + 8 + j] + in[yi + 8 - j]) * a.gk[R + j];
                const int y = 8 * sy + yi;
                if (y < H) {
                    bmax = fmaxf(bmax, fmaxf(sacc.x, sacc.y));
                    if (near_peak && y >= py - 2 && y <= py + 2) {
                        const int dx0 = 2 * xp - px + 2;
                        if (dx0 >= 0 && dx0 < 5) ST[(y - py + 2) * 5 + dx0] = sacc.x;
                        if (dx0 + 1 >= 0 && dx0 + 1 < 5) ST[(y - py + 2) * 5 + dx0 + 1] = sacc.y;
                    }
                }
            }
        }
        int dummy = tid;
        block_argmax(bmax, dummy, s_v, s_i);       // maximum of the blurred map (the index is not used); publishes ST
        auto Bl = [&](int yy, int xx) -> float { return ST[(yy - py + 2) * 5 + (xx - px + 2)]; };
        // The logarithms of the stencil (correctly rounded float32 log = a double-precision log each, ~150 dependent instructions) are
        // taken by one LANE EACH and handed over through LDS: with all of them on thread 0 they were the longest stretch of the whole
        // workgroup (13 in a row, ~3 us, while 511 threads waited)
        float* LT = ST + 32;
        if (a.post == 2) {
            if (has_peak) {
                if (tid < 9) {
                    const int yy = min(max(py + tid / 3 - 1, 0), H - 1), xx = min(max(px + tid % 3 - 1, 0), W - 1);
                    const float v = fminf(fmaxf(Bl(yy, xx), 0.001f), 50.0f);
                    LT[tid] = (float)log((double)v);
                }
                __syncthreads();
            }
            if (tid == 0 && has_peak) {
                auto L = [&](int dy, int dx) -> float { return LT[(dy + 1) * 3 + dx + 1]; };   // (clamped coordinates were applied above)
                const float i_ = L(0, 0), ix1 = L(0, 1), iy1 = L(1, 0), ix1y1 = L(1, 1);
                const float ix1_y1_ = L(-1, -1), ix1_ = L(0, -1), iy1_ = L(-1, 0);
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
        } else if (1 < px && px < W - 2 && 1 < py && py < H - 2) {      // (uniform: px, py are the workgroup's arg-max)
            const float scale = maxval / bmax;     // float32 / float32
            if (tid < 25) {
                float v = __fmul_rn(ST[tid], scale);
                v = fmaxf(v, 1e-10f);
                LT[tid] = (float)log((double)v);   // correctly rounded float32 log
            }
            __syncthreads();
            if (tid == 0) {
                auto L = [&](int yy, int xx) -> float { return LT[(yy - py + 2) * 5 + (xx - px + 2)]; };
                const float l00 = L(py, px);
                const float lxp = L(py, px + 1), lxm = L(py, px - 1), lyp = L(py + 1, px), lym = L(py - 1, px);
                const float lxpp = L(py, px + 2), lxmm = L(py, px - 2), lypp = L(py + 2, px), lymm = L(py - 2, px);
                const float lpp = L(py + 1, px + 1), lmp = L(py - 1, px + 1), lpm = L(py + 1, px - 1), lmm = L(py - 1, px - 1);
                const double dx = 0.5 * (double)__fsub_rn(lxp, lxm);
                const double dy = 0.5 * (double)__fsub_rn(lyp, lym);
                const double dxx = 0.25 * (((double)lxpp - 2.0 * (double)l00) + (double)lxmm);
                const double dyy = 0.25 * (((double)lypp - 2.0 * (double)l00) + (double)lymm);
                const double dxy = 0.25 * (double)__fadd_rn(__fsub_rn(__fsub_rn(lpp, lmp), lpm), lmm);
                const double det = dxx * dyy - dxy * dxy;
                if (det != 0.0) {
                    const double i00 = dyy / det, i01 = -dxy / det, i11 = dxx / det;
                    const double ox = -(i00 * dx + i01 * dy);
                    const double oy = -(i01 * dx + i11 * dy);
                    cxp = (float)((double)cxp + ox);
                    cyp = (float)((double)cyp + oy);
                }
            }
        }
    } else if (a.post == 0) {
        if (tid == 0 && 1 < px && px < W - 1 && 1 < py && py < H - 1) {
            const float ddx = __fsub_rn(Aat(py, px + 1), Aat(py, px - 1));
            const float ddy = __fsub_rn(Aat(py + 1, px), Aat(py - 1, px));
            const float sx = ddx > 0.f ? 1.f : (ddx < 0.f ? -1.f : 0.f);
            const float sy = ddy > 0.f ? 1.f : (ddy < 0.f ? -1.f : 0.f);
            cxp = __fadd_rn(cxp, sx * 0.25f);
            cyp = __fadd_rn(cyp, sy * 0.25f);
        }
    }

    // ---- transform_preds (float32 array arithmetic with float64 scalars cast to float32), as the generic kernel -----------------------
    if (tid == 0) {
        const float* cs = a.center_scale + 4 * n;
        const float s200x = __fmul_rn(cs[2], 200.0f), s200y = __fmul_rn(cs[3], 200.0f);
        const float scale_x = (float)((double)s200x / (a.post == 2 ? (double)W - 1.0 : (double)W));
        const float scale_y = (float)((double)s200y / (a.post == 2 ? (double)H - 1.0 : (double)H));
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
    // the fast form (see flip_merge_decode_fast_kernel): even maps with w % 4 == 0 up to 128 x 128, blur kernels 11 / 17 (or no blur)
    // A/B and test knob (both kernels give identical bits: tests/test_gpu_decode_fast.py): pp_debug_knob("decode_generic", 1), else
    // POSEPIPE_DECODE_GENERIC read ONCE -- no environment scan per launch
    static const int gen_env = [] { const char* e = getenv("POSEPIPE_DECODE_GENERIC"); return e ? atoi(e) : 0; }();
    const int gen_knob = g_decode_generic.load(std::memory_order_relaxed);
    const bool fast_off = (gen_knob >= 0 ? gen_knob : gen_env) != 0;
    const bool blur = p.post >= 1;
    // UDP (post 2) fills its frame by REFLECTION (BORDER_REFLECT_101): column / row j of the frame mirrors index j, which must be a
    // map cell -- an 8-pixel map with the 17-tap kernel (radius 8) would read index 8, a frame cell other threads are writing
    // (ADVICE r4); such maps take the generic kernel, whose mirror() folds the index back
    const bool reflect_ok = p.post != 2 || (p.h > p.blur_kernel / 2 && p.w > p.blur_kernel / 2);
    const bool fast_ok = !fast_off && p.h % 2 == 0 && p.w % 4 == 0 && p.h <= 128 && p.w <= 128 && p.h >= 8 && p.w >= 8 &&
                         (!blur || p.blur_kernel == 11 || p.blur_kernel == 17) && reflect_ok;
    if (fast_ok) {
        FastGeom g{};
        const int w8 = (p.w + 7) / 8 * 8, h8 = (p.h + 7) / 8 * 8;
        g.wpa = w8 + 16;
        g.hpb = h8 + 16;
        g.c4 = p.w / 4;
        g.load_tasks = (p.h / 2) * g.c4;
        g.nseg = w8 / 8;
        g.row_tasks = (p.h / 2) * g.nseg;
        g.nsegy = h8 / 8;
        g.col_tasks = (p.w / 2) * g.nsegy;
        g.inv_c4 = 1.0f / (float)g.c4;
        g.inv_nseg = 1.0f / (float)g.nseg;
        g.inv_nsegy = 1.0f / (float)g.nsegy;
        const size_t lds_fast = ((size_t)(p.h / 2) * g.wpa * 2 + (blur ? (size_t)(w8 / 2) * g.hpb * 2 + 64 : 0)) * sizeof(float);
        static PpPerDeviceOnce fast_attr;      // (once per device, thread-safe)
        fast_attr.run([] {
            (void)hipFuncSetAttribute((const void*)flip_merge_decode_fast_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
            (void)hipFuncSetAttribute((const void*)flip_merge_decode_fast_kernel<11>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
            (void)hipFuncSetAttribute((const void*)flip_merge_decode_fast_kernel<17>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
        });
        if (!blur) hipLaunchKernelGGL(flip_merge_decode_fast_kernel<0>, dim3(p.n * p.k), dim3(512), lds_fast, s, a, g);
        else if (p.blur_kernel == 11) hipLaunchKernelGGL(flip_merge_decode_fast_kernel<11>, dim3(p.n * p.k), dim3(512), lds_fast, s, a, g);
        else hipLaunchKernelGGL(flip_merge_decode_fast_kernel<17>, dim3(p.n * p.k), dim3(512), lds_fast, s, a, g);
        PP_HIP_CHECK(hipGetLastError());
        return PP_OK;
    }
    static PpPerDeviceOnce attr_set;
    if (lds > 64 * 1024)
        attr_set.run([] { (void)hipFuncSetAttribute((const void*)flip_merge_decode_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64); });
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
