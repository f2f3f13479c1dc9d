// Implicit-GEMM convolution on the gfx950 fp32 matrix cores (v_mfma_f32_16x16x4_f32).
//
// Replaces the cuDNN conv + BatchNorm + ReLU (+ residual add, + nearest-upsample add) stacks that
// the reference reaches through mmpose HRNet / mmdet ResNet-FPN / VideoPose3D (arch
// specs: 3rdparty/mmpose/config/top_down/darkpose/coco/hrnet_w48_coco_384x288_dark.py:44-79,
// 3rdparty/mmtracking/_base_/models/faster_rcnn_r50_fpn.py:1-112, wrappers/videopose3d.py:46-50).
//
// GEMM view:  D[cout][pixel] = sum_k  W[k][cout] * X[k][pixel],  k = (kh*KW + kw)*Cin + cin,
// pixel = (n, ho, wo).  MFMA "A" rows are output channels, "B" columns are pixels, so a lane ends
// up holding 4 consecutive output channels of one pixel = one 16-byte NHWC store.
//
// Numerics: the f32 MFMA is bit-for-bit a k-ordered fmaf chain; the K loop below feeds k in
// natural order, so every output equals  fmaf(x_{K-1}, w_{K-1}, ... fmaf(x_0, w_0, 0))  exactly,
// which is what oracle/conv_ref.c computes.  Zero padding contributes fmaf(0, w, acc) == acc.
//
// Layout: activations NHWC fp32 (Cin % 4 == 0); bias [CoutPad16].  K is cut into chunks of 32.
// Weight blob: [Kpad32/32][CoutPad16][32] with the 32 k's of a chunk stored per output channel in
// "operand order": element 8*g + s holds k = 4*s + g.  Lane (row, g = lane>>4) of an MFMA consumes
// k = 4*s + g at step s, so its 8 operands of a chunk are 8 contiguous floats = two ds_read_b128.
//
// LDS: both tiles are [row][32] (row = output channel / pixel, 128-byte rows, no padding) in operand
// order, with the 16-byte slot index XOR-swizzled by a function of (row & 15) chosen (exhaustive
// search over GF(2)-linear maps) so that every ds_read_b128 lane group AND the transposing
// ds_write_b32 pattern of the pixel loader are bank-conflict free.
// Pixel loader: lanes run along k first (8 lanes x 16 B = one 128-byte line of a pixel's channels), so a
// wave-load touches 8 cache lines instead of 64.
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

#include "pp_internal.h"
#include "pp_amax.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BK = 32;

__device__ __forceinline__ int swz(int j) {
    // GF(2)-linear map of the row index (bit masks 2, 1, 6): conflict-free for the b128 lane groups
    return ((j >> 1) & 1) | ((j & 1) << 1) | ((((j >> 1) ^ (j >> 2)) & 1) << 2);
}

// unsigned division by an invariant divisor (Granlund & Montgomery 1994, Fig. 4.1): exact for all 32-bit n
__device__ __forceinline__ unsigned udiv(unsigned n, unsigned m, unsigned s1, unsigned s2) {
    const unsigned t = __umulhi(m, n);
    return (t + ((n - t) >> s1)) >> s2;
}

typedef short i16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

// Activations of the DeepSortYOLOv4 path.  The reference evaluates them as separate float32 TensorFlow ops
// (yolo4/model.py:48 `inputs * K.tanh(K.softplus(inputs))`, LeakyReLU(0.1), tf.nn.elu); here every transcendental op is
// evaluated in double precision and rounded to float once, which is what oracle/yolo.py restates.
__device__ __forceinline__ float activate(float x, int act) {
    if (act == PP_ACT_LEAKY) return x >= 0.f ? x : 0.1f * x;
    if (act == PP_ACT_MISH) {
        // tanh(log(1 + e^x)) = n(n + 2) / (n(n + 2) + 2) with n = e^x: one exp and one division in fp64, rounded to float
        // once; 1.0f beyond x = 20 (the true value is 1 - 2e-18) keeps n*n finite
        if (x > 20.f) return x;
        const double n = exp((double)x);
        const double t = n * (n + 2.0);
        const float th = (float)(t / (t + 2.0));
        return x * th;
    }
    if (act == PP_ACT_ELU) return x > 0.f ? x : (float)expm1((double)x);
    if (act == PP_ACT_SWISH) {   // mmcv Swish: x * torch.sigmoid(x) -- sigmoid in fp64 rounded once, then a float product
        const float s = (float)(1.0 / (1.0 + exp(-(double)x)));
        return x * s;
    }
    return x;
}

// WALK: Cin >= 32, so a 32-k chunk spans at most two kernel taps and the tap bookkeeping can live on the scalar unit
// (see load_chunk).  The generic variant decodes every lane's tap with two magic divisions per chunk.
template <int CT, int PT, bool WALK>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs a) {
    constexpr int BC = 16 * CT;          // output channels per block
    constexpr int BP = 64 * PT;          // pixels per block (4 waves x PT x 16)
    constexpr int NQ = 2 * PT;           // pixel quads per thread and chunk
    constexpr int NW = (BC * 8 + 255) / 256;   // weight float4s per thread and chunk
    constexpr int WROWS = NW * 32;       // weight rows staged per chunk (>= BC; the extra rows are never read)
    __shared__ __attribute__((aligned(16))) float smem[(WROWS + BP) * BK];
    float* Ws = smem;
    float* Xs = smem + WROWS * BK;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    // XCD-aware tile order: the dispatcher places linear workgroup id L on XCD L % 8, each XCD has its own L2.
    // Give every XCD a contiguous range of tiles, channel tile fastest, so that the channel tiles of one pixel
    // tile (same im2col data) and vertically adjacent pixel tiles (shared halo rows) meet in the same L2.
    int tile_x = blockIdx.x, tile_y = blockIdx.y;
    if (a.xcd_remap) {
        const unsigned total = gridDim.x * gridDim.y;
        const unsigned L = blockIdx.x + blockIdx.y * gridDim.x;
        const unsigned xcd = L & 7u, j = L >> 3;
        const unsigned q = total >> 3, r = total & 7u;
        const unsigned Lp = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
        tile_y = (int)(Lp % gridDim.y);
        tile_x = (int)(Lp / gridDim.y);
    }
    const int m0 = tile_x * BP;
    const int c0 = tile_y * BC;

    // ---- loader role: k-quad kq of pixel rows prow0 + 32*i, weight float4s tid + 256*j ------------------
    // Everything after the second barrier of a K step is ONE basic block (out-of-range buffer offsets and clamped
    // weight rows instead of branches), so loads, address arithmetic and MFMAs can be ordered freely (see the loop).
    const int kq = tid & 7;
    const int prow0 = tid >> 3;
    const int wsw = swz(prow0 & 15);     // rows prow0 + 32*i share (row & 15)
    unsigned pbase[NQ];                  // element offset of the pixel's image
    int phw[NQ];                         // (hi0 << 16) | (wi0 & 0xffff); hi0 = -32768 for rows past M (never in range)
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const int m = m0 + prow0 + 32 * i;
        const bool mok = m < a.M;
        const unsigned mm = mok ? (unsigned)m : 0u;
        const int n = (int)udiv(mm, a.div_hw_m, a.div_hw_s1, a.div_hw_s2);
        const int rem = (int)mm - n * a.HWout;
        const int ho = (int)udiv((unsigned)rem, a.div_w_m, a.div_w_s1, a.div_w_s2);
        const int wo = rem - ho * a.Wout;
        const int hi0 = ho * a.stride - a.pad_h, wi0 = wo * a.stride - a.pad_w;
        // WALK: byte offset of (n, hi0, wi0, channel 0) modulo 2^32 (hi0 / wi0 may be negative); else: the image's
        // element offset
        pbase[i] = WALK ? (unsigned)(((n * a.Hin + hi0) * a.Win + wi0) * a.Cin) * 4u
                        : (unsigned)n * (unsigned)(a.Hin * a.Win * a.Cin);
        phw[i] = mok ? ((hi0 << 16) | (wi0 & 0xffff)) : (int)0x80000000;
    }
    static_assert(NW <= 2, "weight loader handles at most 64 rows");
    // float offset of this thread's weight float4s inside a chunk of the blob (rows past the blob re-read row 0)
    const unsigned wofs0 = (unsigned)(((c0 + prow0 < a.CoutPad) ? prow0 : 0) * BK + kq * 4);
    const unsigned wofs1 = (unsigned)(((c0 + prow0 + 32 < a.CoutPad) ? prow0 + 32 : 0) * BK + kq * 4);
    const float* wblob = a.w + (size_t)c0 * BK;

    float4 xr[NQ];
    float4 wr0, wr1;
    // raw buffer descriptor over the whole input (x_bytes < 4 GiB: the launcher splits larger batches)
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)a.x_bytes, 0x00020000);

    // WALK state, all wave-uniform (SGPRs): the chunk starts c0s channels into tap (kh0, kw0)
    int c0s = 0, kh0 = 0, kw0 = 0;
    const int kq16 = 16 * kq;
    const unsigned lim = ((unsigned)(a.Hin - 1) << 16) | (unsigned)(a.Win - 1);

    auto load_chunk = [&](int k0) {
        const bool kok = k0 + 4 * kq < a.K;
        if constexpr (WALK) {
            // VALU instructions take matrix-pipe time on this chip (~2.5 cycles each, tools/mfma_peak.hip), so the tap
            // walk is scalar: lanes whose quad lies before the tap boundary of this chunk use (tap0, d0), the others
            // (tap1, d1); per pixel that leaves one packed 16-bit add + max + compare for the bounds check and one add
            // + select for the byte offset.
            int kw1 = kw0 + 1, kh1 = kh0;
            if (kw1 == a.KW) {
                kw1 = 0;
                kh1 = kh0 + 1;
            }
            const unsigned d0 = (unsigned)((kh0 * a.dil_h * a.Win + kw0 * a.dil_w) * a.Cin + c0s) * 4u;
            const unsigned d1 = (unsigned)((kh1 * a.dil_h * a.Win + kw1 * a.dil_w) * a.Cin + c0s - a.Cin) * 4u;
            const unsigned p0 = ((unsigned)(kh0 * a.dil_h) << 16) | (unsigned)(kw0 * a.dil_w);
            const unsigned p1 = ((unsigned)(kh1 * a.dil_h) << 16) | (unsigned)(kw1 * a.dil_w);
            const bool first = 4 * kq < a.Cin - c0s;
            // k >= K (zero-padded tail of the last chunk): a tap offset that no pixel can satisfy
            const unsigned dpk = kok ? (first ? p0 : p1) : 0x7fff7fffu;
            const unsigned delta = (first ? d0 : d1) + (unsigned)kq16;
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                // (hi, wi) as two u16 halves; in range iff max(hi, Hin-1) == Hin-1 and max(wi, Win-1) == Win-1
                const u16x2 hw = __builtin_bit_cast(u16x2, __builtin_bit_cast(i16x2, phw[i]) + __builtin_bit_cast(i16x2, dpk));
                const bool ok = __builtin_bit_cast(unsigned, __builtin_elementwise_max(hw, __builtin_bit_cast(u16x2, lim))) == lim;
                const unsigned off = ok ? pbase[i] + delta : 0xffffffffu;
                xr[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)off, 0, 0));
            }
            c0s += BK;
            if (c0s >= a.Cin) {
                c0s -= a.Cin;
                kh0 = kh1;
                kw0 = kw1;
            }
        } else {
            // (kh, kw, c) of this thread's quad: k = k0 + 4*kq = (kh*KW + kw)*Cin + c
            const unsigned k4 = (unsigned)(k0 + 4 * kq);
            const unsigned tap = udiv(k4, a.div_c_m, a.div_c_s1, a.div_c_s2);
            const int qc = (int)(k4 - tap * (unsigned)a.Cin);
            const int qkh = (int)udiv(tap, a.div_kw_m, a.div_kw_s1, a.div_kw_s2);
            const int qkw = (int)tap - qkh * a.KW;
            const int dh = qkh * a.dil_h, dw = qkw * a.dil_w;
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                const int hi = (phw[i] >> 16) + dh;
                const int wi = (int)(short)(phw[i] & 0xffff) + dw;
                const bool ok = kok && (unsigned)hi < (unsigned)a.Hin && (unsigned)wi < (unsigned)a.Win;
                // out-of-image taps / rows past M / k past K: an out-of-range byte offset, for which the buffer load
                // returns zeros -- no branch, no select
                const unsigned off = ok ? (pbase[i] + (unsigned)((hi * a.Win + wi) * a.Cin + qc)) * 4u : 0xffffffffu;
                xr[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)off, 0, 0));
            }
        }
        const float* wsrc = wblob + (size_t)(k0 / BK) * a.CoutPad * BK;
        wr0 = *reinterpret_cast<const float4*>(wsrc + wofs0);
        if (NW > 1) wr1 = *reinterpret_cast<const float4*>(wsrc + wofs1);
    };

    auto store_chunk = [&]() {
        // transpose: element r of the quad is k = 4*kq + r  ->  operand position 8*r + kq
        const int within = kq & 3, hi_slot = kq >> 2;
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            float* row = Xs + (prow0 + 32 * i) * BK + within;
            row[((0 + hi_slot) ^ wsw) * 4] = xr[i].x;
            row[((2 + hi_slot) ^ wsw) * 4] = xr[i].y;
            row[((4 + hi_slot) ^ wsw) * 4] = xr[i].z;
            row[((6 + hi_slot) ^ wsw) * 4] = xr[i].w;
        }
        *reinterpret_cast<float4*>(Ws + prow0 * BK + ((kq ^ wsw) * 4)) = wr0;
        if (NW > 1) *reinterpret_cast<float4*>(Ws + (prow0 + 32) * BK + ((kq ^ wsw) * 4)) = wr1;
    };

    f32x4 acc[CT][PT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) acc[ct][pt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int lrow = lane >> 4;   // g: k offset within an MFMA step
    const int lcol = lane & 15;   // cout (A) / pixel (B) within the 16-tile
    const int rsw = swz(lcol);
    const float* wrd = Ws + lcol * BK;
    const float* xrd = Xs + (wave * (16 * PT) + lcol) * BK;

    f32x4 av[CT], bv[PT];
    auto read_operands = [&](int h) {
        const int so = ((2 * lrow + h) ^ rsw) * 4;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) av[ct] = *reinterpret_cast<const f32x4*>(wrd + ct * 16 * BK + so);
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) bv[pt] = *reinterpret_cast<const f32x4*>(xrd + pt * 16 * BK + so);
    };
    auto mma_steps = [&](int s0, int s1) {
#pragma unroll
        for (int s = s0; s < s1; ++s)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int pt = 0; pt < PT; ++pt)
                    acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ct][s], bv[pt][s], acc[ct][pt], 0, 0, 0);
    };

    load_chunk(0);
    int k0 = 0;
    for (; k0 + BK < a.Kpad; k0 += BK) {
        __syncthreads();
        store_chunk();
        __syncthreads();
        // Order after the barrier: operand reads first (their LDS latency hides behind what follows), one MFMA step
        // to get the matrix pipe going, and only then the address arithmetic + issue of the next chunk's loads.
        read_operands(0);
        __builtin_amdgcn_sched_barrier(0);
        mma_steps(0, 1);
        __builtin_amdgcn_sched_barrier(0);
        load_chunk(k0 + BK);
        __builtin_amdgcn_sched_barrier(0);
        mma_steps(1, 4);
        read_operands(1);
        mma_steps(0, 4);
    }
    __syncthreads();
    store_chunk();
    __syncthreads();
    read_operands(0);
    mma_steps(0, 4);
    if (k0 + 16 < a.K) {   // the second half of the last chunk may hold no taps (e.g. K = 9*48)
        read_operands(1);
        mma_steps(0, 4);
    }

    // ---- epilogue: bias, residuals, ReLU, (upsampled / NCHW) store ----------------------------
    const int up = a.up_log2;
    const int f = 1 << up;
    const int Ho2 = a.Hout << up, Wo2 = a.Wout << up;   // dims of the out buffer
    const bool vec4 = ((a.Cout & 3) == 0) && !a.out_nchw;
    const bool res1_plain = (a.res1_shift == 0 && a.res1_off_w == 0 && a.res1_H == Ho2 && a.res1_W == Wo2);
    if (up == 0 && vec4 && (!a.res1 || res1_plain) && a.relu <= PP_RELU_FIRST && a.y_stride == a.Cout) {
        // common case (BasicBlock / Bottleneck / plain convs): the output pixel index IS m, no coordinate math.
        // All bias / residual loads are issued back to back from clamped (always valid) addresses before anything
        // consumes them: one memory round trip per phase instead of one per 16x16 tile.
        // Loads are batched PB pixel tiles at a time so the live set stays inside the main loop's register budget.
        constexpr int PB = (CT * PT <= 6) ? PT : 1;
        bool cok[CT];
        int cos[CT];
        float4 b4[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int co = c0 + ct * 16 + 4 * lrow;
            cok[ct] = co < a.Cout;
            cos[ct] = cok[ct] ? co : 0;
            b4[ct] = *reinterpret_cast<const float4*>(a.bias + cos[ct]);
        }
        float ymax[PT];
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) ymax[pt] = 0.f;
#pragma unroll
        for (int p0 = 0; p0 < PT; p0 += PB) {
            bool mok[PB];
            size_t moff[PB];
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) {
                const int m = m0 + wave * (16 * PT) + (p0 + pb) * 16 + lcol;
                mok[pb] = m < a.M;
                moff[pb] = (size_t)(mok[pb] ? m : 0) * (size_t)a.Cout;
            }
            float4 o[CT][PB];
            if (a.res1) {
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int pb = 0; pb < PB; ++pb)
                        o[ct][pb] = *reinterpret_cast<const float4*>(a.res1 + moff[pb] + cos[ct]);
            }
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int pb = 0; pb < PB; ++pb) {
                    const f32x4 c = acc[ct][p0 + pb];
                    float4 v = make_float4(c[0] + b4[ct].x, c[1] + b4[ct].y, c[2] + b4[ct].z, c[3] + b4[ct].w);
                    if (a.relu == PP_RELU_FIRST) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    if (a.res1) { v.x += o[ct][pb].x; v.y += o[ct][pb].y; v.z += o[ct][pb].z; v.w += o[ct][pb].w; }
                    o[ct][pb] = v;
                }
            if (a.res2) {
                float4 r2[CT][PB];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int pb = 0; pb < PB; ++pb)
                        r2[ct][pb] = *reinterpret_cast<const float4*>(a.res2 + moff[pb] + cos[ct]);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int pb = 0; pb < PB; ++pb) {
                        o[ct][pb].x += r2[ct][pb].x; o[ct][pb].y += r2[ct][pb].y;
                        o[ct][pb].z += r2[ct][pb].z; o[ct][pb].w += r2[ct][pb].w;
                    }
            }
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int pb = 0; pb < PB; ++pb) {
                    float4 v = o[ct][pb];
                    if (a.relu == PP_RELU_LAST) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    if (mok[pb] && cok[ct]) {
                        *reinterpret_cast<float4*>(a.y + moff[pb] + cos[ct]) = v;
                        ymax[p0 + pb] = fmaxf(ymax[p0 + pb], pp_abs4max(v));
                    }
                }
        }
        if (a.y_amax) {              // a fp16-form convolution reads this tensor: max |y| per sample of what was stored (pp_amax.h)
            __shared__ float amax_red[16];
            int yimg[PT];
#pragma unroll
            for (int pt = 0; pt < PT; ++pt)
                yimg[pt] = (int)udiv((unsigned)min(m0 + wave * (16 * PT) + pt * 16 + lcol, a.M - 1), a.div_hw_m, a.div_hw_s1, a.div_hw_s2);
            const unsigned wlast = (unsigned)min(m0 + 64 * PT - 1, a.M - 1);
            pp_amax_commit_wg<4, PT>(a.y_amax, yimg, ymax, (int)udiv((unsigned)min(m0, (int)wlast), a.div_hw_m, a.div_hw_s1, a.div_hw_s2),
                                     (int)udiv(wlast, a.div_hw_m, a.div_hw_s1, a.div_hw_s2), amax_red);
        }
        return;
    }
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const int m = m0 + wave * (16 * PT) + pt * 16 + lcol;
        if (m >= a.M) continue;
        const int n = (int)udiv((unsigned)m, a.div_hw_m, a.div_hw_s1, a.div_hw_s2);
        const int rem = m - n * a.HWout;
        const int ho = (int)udiv((unsigned)rem, a.div_w_m, a.div_w_s1, a.div_w_s2);
        const int wo = rem - ho * a.Wout;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int co = c0 + ct * 16 + 4 * lrow;
            if (co >= a.Cout) continue;
            const float4 b4 = *reinterpret_cast<const float4*>(a.bias + co);
            float v[4] = {acc[ct][pt][0] + b4.x, acc[ct][pt][1] + b4.y, acc[ct][pt][2] + b4.z,
                          acc[ct][pt][3] + b4.w};
            if (a.relu == PP_RELU_FIRST) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            } else if (a.relu >= PP_ACT_LEAKY) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = activate(v[r], a.relu);
            }
            for (int dy = 0; dy < f; ++dy) {
                for (int dx = 0; dx < f; ++dx) {
                    const int h2 = (ho << up) + dy, w2 = (wo << up) + dx;
                    const size_t opix = ((size_t)n * Ho2 + h2) * Wo2 + w2;
                    float o[4] = {v[0], v[1], v[2], v[3]};
                    if (a.res1) {
                        size_t rpix = opix;
                        if (!res1_plain) {
                            rpix = ((size_t)n * a.res1_H + (h2 >> a.res1_shift)) * a.res1_W +
                                   (w2 >> a.res1_shift) + a.res1_off_w;
                        }
                        const float* rp = a.res1 + rpix * a.Cout + co;
                        if (vec4) {
                            const float4 r4 = *reinterpret_cast<const float4*>(rp);
                            o[0] += r4.x; o[1] += r4.y; o[2] += r4.z; o[3] += r4.w;
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (co + r < a.Cout) o[r] += rp[r];
                        }
                    }
                    if (a.res2) {
                        const float* rp = a.res2 + opix * a.Cout + co;
                        if (vec4) {
                            const float4 r4 = *reinterpret_cast<const float4*>(rp);
                            o[0] += r4.x; o[1] += r4.y; o[2] += r4.z; o[3] += r4.w;
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (co + r < a.Cout) o[r] += rp[r];
                        }
                    }
                    if (a.relu == PP_RELU_LAST) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = fmaxf(o[r], 0.f);
                    }
                    if (a.out_nchw) {
                        const size_t plane = (size_t)Ho2 * Wo2;
                        float* yp = a.y + ((size_t)n * a.Cout + co) * plane + (size_t)h2 * Wo2 + w2;
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (co + r < a.Cout) yp[r * plane] = o[r];
                    } else if (vec4) {
                        *reinterpret_cast<float4*>(a.y + opix * a.y_stride + a.y_coff + co) = make_float4(o[0], o[1], o[2], o[3]);
                    } else {
                        float* yp = a.y + opix * a.y_stride + a.y_coff + co;
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (co + r < a.Cout) yp[r] = o[r];
                    }
                }
            }
        }
    }
}

template <int CT, int PT>
int launch_t(const ConvArgs& a, hipStream_t stream) {
    dim3 grid((a.M + 64 * PT - 1) / (64 * PT), (a.CoutPad + 16 * CT - 1) / (16 * CT));
    if (a.Cin >= BK)
        hipLaunchKernelGGL((conv_igemm_kernel<CT, PT, true>), grid, dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL((conv_igemm_kernel<CT, PT, false>), grid, dim3(256), 0, stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        pp_set_error("conv_igemm launch failed: %s", hipGetErrorString(e));
        return PP_ERR_HIP;
    }
    return PP_OK;
}

template <int CT>
int launch_ct(const ConvArgs& a, int pt, hipStream_t stream) {
    // PT = 4 for the narrow channel tiles (CT <= 2, same 32 accumulator registers) measured 92.5 vs 94.3 TFLOP/s on the
    // 32 -> 32 3x3 layers of HRNet-W32: not instantiated
    return pt >= 2 ? launch_t<CT, 2>(a, stream) : launch_t<CT, 1>(a, stream);
}

// ---- max pool (NHWC, 4 channels per thread) ---------------------------------------------------
__global__ __launch_bounds__(256) void maxpool_kernel(PoolArgs a) {
    const int c4n = a.C >> 2;
    const size_t xs = a.x_stride ? a.x_stride : a.C, ys = a.y_stride ? a.y_stride : a.C;   // channel slices of wider buffers
    const size_t total = (size_t)a.N * a.Hout * a.Wout * c4n;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = i % c4n;
        size_t p = i / c4n;
        const int wo = p % a.Wout;
        p /= a.Wout;
        const int ho = p % a.Hout;
        const int n = p / a.Hout;
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        for (int kh = 0; kh < a.KH; ++kh) {
            const int hi = ho * a.stride - a.pad_h + kh;
            if ((unsigned)hi >= (unsigned)a.Hin) continue;
            for (int kw = 0; kw < a.KW; ++kw) {
                const int wi = wo * a.stride - a.pad_w + kw;
                if ((unsigned)wi >= (unsigned)a.Win) continue;
                const float4 v = *reinterpret_cast<const float4*>(
                    a.x + (((size_t)n * a.Hin + hi) * a.Win + wi) * xs + a.x_coff + 4 * c4);
                m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
            }
        }
        *reinterpret_cast<float4*>(a.y + (((size_t)n * a.Hout + ho) * a.Wout + wo) * ys + a.y_coff + 4 * c4) = m;
    }
}

int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

// ---- tile selection: measured table first, heuristic otherwise --------------------------------------------------------
// The best (channel tile, pixel tile) of a layer depends on how its grid quantises onto 256 CUs x 4 resident blocks, which
// no closed formula predicts well (HRNet's 24x18 / 12x9 layers launch 0.4-0.9 "rounds" of blocks).  tools/autotune_conv.py
// measures every instantiated (CT, PT) per layer shape on the GPU and writes posepipeline_amd/conv_tuning.txt:
//   Cin Cout KH KW stride dil_h dil_w M  CT PT [variant]     (one line per shape; M = batch * Hout * Wout)
// Results are identical for every choice (same k order per output); only the speed differs.
struct TuneKey {
    int cin, cout, kh, kw, stride, dil_h, dil_w;
    bool operator<(const TuneKey& o) const { return memcmp(this, &o, sizeof(TuneKey)) < 0; }
};
struct TuneTable {
    struct Choice { int m, ct, pt, variant; };
    std::map<TuneKey, std::vector<Choice>> best;     // per layer shape: the measured batch sizes (M), ascending
    TuneTable() {
        std::string path;
        if (const char* e = getenv("POSEPIPE_CONV_TUNING")) {
            path = e;                                    // "" / "0" disables the table
        } else {
            Dl_info info;
            if (dladdr((void*)&pp_conv_out_dim, &info) && info.dli_fname) {
                path = info.dli_fname;
                const size_t slash = path.rfind('/');
                path = (slash == std::string::npos ? std::string(".") : path.substr(0, slash)) + "/conv_tuning.txt";
            }
        }
        if (path.empty() || path == "0") return;
        FILE* f = fopen(path.c_str(), "r");
        if (!f) return;
        char line[256];
        while (fgets(line, sizeof(line), f)) {
            TuneKey k;
            memset(&k, 0, sizeof(k));
            Choice c{0, 0, 0, -1};
            if (line[0] == '#' || sscanf(line, "%d %d %d %d %d %d %d %d %d %d %d", &k.cin, &k.cout, &k.kh, &k.kw, &k.stride, &k.dil_h,
                                         &k.dil_w, &c.m, &c.ct, &c.pt, &c.variant) < 10)
                continue;
            if (c.ct >= 1 && c.ct <= 4 && (c.pt == 1 || c.pt == 2) && c.m > 0 && c.variant != 2) best[k].push_back(c);
        }
        fclose(f);
        for (auto& kv : best) std::sort(kv.second.begin(), kv.second.end(), [](const Choice& x, const Choice& y) { return x.m < y.m; });
    }
    // the entry measured at the nearest M (ratio-wise), if it is within a factor 1.5 (grid quantisation changes beyond that)
    const Choice* find(const TuneKey& k, int m) const {
        auto it = best.find(k);
        if (it == best.end()) return nullptr;
        const Choice* pick = nullptr;
        double best_r = 1.5;
        for (const Choice& c : it->second) {
            const double r = c.m > m ? (double)c.m / m : (double)m / c.m;
            if (r <= best_r) {
                best_r = r;
                pick = &c;
            }
        }
        return pick;
    }
};
// ---- tap tables of the pipelined kernel (variant 3) ---------------------------------------------------------------------------
struct TapKey {
    int dev, cin, kh, kw, dil_h, dil_w, win, k;
    bool operator<(const TapKey& o) const { return memcmp(this, &o, sizeof(TapKey)) < 0; }
};
std::mutex g_tap_mutex;
std::map<TapKey, uint2*> g_tap_tables;

const uint2* tap_table_for(const ConvArgs& a, bool create) {
    TapKey key;
    memset(&key, 0, sizeof(key));
    if (hipGetDevice(&key.dev) != hipSuccess) return nullptr;
    const int pitch = a.Win + a.x_pad;        // pixels per stored row
    key.cin = a.Cin; key.kh = a.KH; key.kw = a.KW; key.dil_h = a.dil_h; key.dil_w = a.dil_w; key.win = pitch; key.k = a.K;
    std::lock_guard<std::mutex> lock(g_tap_mutex);
    auto it = g_tap_tables.find(key);
    if (it != g_tap_tables.end()) return it->second;
    if (!create) return nullptr;
    const int nsteps = (a.K + 15) / 16;
    std::vector<uint2> h((size_t)(nsteps + 1) * 4);            // + one spare step: the kernel prefetches one entry ahead
    for (int s = 0; s <= nsteps; ++s)
        for (int q = 0; q < 4; ++q) {
            const int k = 16 * s + 4 * q;
            // k >= K: a tap no pixel can satisfy (checked kernel) and an offset beyond any tensor < 2 GiB (unchecked kernel):
            // either way the buffer load returns zeros
            uint2 e = make_uint2(0x7fff7fffu, 0x80000000u);
            if (k < a.K) {
                const int tap = k / a.Cin, c = k - tap * a.Cin;
                const int kh = tap / a.KW, kw = tap - kh * a.KW;
                e.x = ((unsigned)(kh * a.dil_h) << 16) | (unsigned)(kw * a.dil_w);
                e.y = (unsigned)(((kh * a.dil_h * pitch + kw * a.dil_w) * a.Cin + c) * 4);
            }
            h[(size_t)s * 4 + q] = e;
        }
    uint2* d = nullptr;
    if (hipMalloc((void**)&d, h.size() * sizeof(uint2)) != hipSuccess) return nullptr;
    if (hipMemcpy(d, h.data(), h.size() * sizeof(uint2), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(d);
        return nullptr;
    }
    g_tap_tables[key] = d;
    return d;
}

int g_force_ct = 0, g_force_pt = 0;   // pp_conv_force (autotuner, A/B experiments)
int g_variant = -1;                   // pp_conv_variant: -1 = default (env POSEPIPE_CONV_VARIANT, else table / built-in)

}  // namespace

extern "C" int pp_conv_variant(int variant) {
    if (variant < -1 || variant > 4) {      // 2: timing experiment with fake addresses (wrong results), tools/conv_probe.py only
        pp_set_error("pp_conv_variant: -1 (default), 0 (two-barrier kernel), 1 (three-stage pipelined kernel), 3 (pipelined + tap table: "
                     "the bit-exact default) or 4 (bf16-split kernel where eligible)");
        return PP_ERR_ARG;
    }
    g_variant = variant;
    return PP_OK;
}

extern "C" int pp_conv_force(int ct, int pt) {
    if (ct < 0 || ct > 4 || (pt != 0 && pt != 1 && pt != 2)) {
        pp_set_error("pp_conv_force: ct in 0..4, pt in {0, 1, 2} (0 = automatic)");
        return PP_ERR_ARG;
    }
    g_force_ct = ct;
    g_force_pt = pt;
    return PP_OK;
}

int pp_conv_out_dim(int in, int k, int stride, int pad, int dil) {
    return (in + 2 * pad - dil * (k - 1) - 1) / stride + 1;
}

static void magic_u32(unsigned d, unsigned* m, unsigned* s1, unsigned* s2) {
    unsigned l = 0;
    while ((1ull << l) < d) ++l;                               // ceil(log2 d)
    *m = (unsigned)((((1ull << l) - d) << 32) / d + 1);
    *s1 = l < 1 ? l : 1;
    *s2 = l > 0 ? l - 1 : 0;
}

// The split kernel is the default where a layer is eligible; POSEPIPE_CONV_EXACT=1 or an explicit variant 0 / 1 / 3 keeps every
// layer on the bit-exact fp32-MFMA kernels.
// Process-wide DEFAULT only (ABI 7): read once when a net is created and by the single-op API; a net's numerics are its own.
std::atomic<int> g_exact{-1};         // pp_conv_exact: -1 = POSEPIPE_CONV_EXACT
bool pp_conv_split_enabled() {
    static const int env_exact = env_int("POSEPIPE_CONV_EXACT", 0);
    static const int env_variant = env_int("POSEPIPE_CONV_VARIANT", -1);
    const int v = g_variant >= 0 ? g_variant : env_variant;
    const int ge = g_exact.load(std::memory_order_relaxed);
    const int exact = ge >= 0 ? ge : env_exact;
    return v == 4 || (v < 0 && !exact);
}

std::atomic<int> g_split_f16{-1};     // pp_conv_split_kind: -1 = POSEPIPE_SPLIT_F16
bool pp_conv_split_f16_default() {
    static const int env_f16 = env_int("POSEPIPE_SPLIT_F16", 1);     // round 5: the fp16 form is the default split form
    const int g = g_split_f16.load(std::memory_order_relaxed);
    return (g >= 0 ? g : env_f16) != 0;
}

extern "C" int pp_conv_split_kind(int f16) {
    if (f16 < -1 || f16 > 1) {
        pp_set_error("pp_conv_split_kind: 1 (two float16 terms, three products), 0 (three bfloat16 terms, six products) or -1 (POSEPIPE_SPLIT_F16)");
        return PP_ERR_ARG;
    }
    g_split_f16.store(f16, std::memory_order_relaxed);
    return PP_OK;
}

extern "C" int pp_conv_exact(int exact) {
    if (exact < -1 || exact > 1) {
        pp_set_error("pp_conv_exact: 1 (bit-exact fp32 MFMA kernels), 0 (bf16-split kernels where eligible) or -1 (POSEPIPE_CONV_EXACT)");
        return PP_ERR_ARG;
    }
    g_exact.store(exact, std::memory_order_relaxed);
    return PP_OK;
}

// (the float32 kernels' condition is the one their "common case" epilogue tests)
bool pp_conv_tracks_amax(const ConvArgs& a, bool split) {
    // samples of a few pixels (the RoI head's FC layers: ONE pixel per sample): a workgroup's 256 pixels span hundreds of samples, the
    // fused epilogue would issue an atomic per pixel and channel block (measured: fc6 at 64 000 RoIs 5.3 -> 8.4 ms) where a pass over
    // the small output costs 0.1 ms
    if (a.HWout < 64) return false;
    if (split) return true;
    const int Ho2 = a.Hout << a.up_log2, Wo2 = a.Wout << a.up_log2;
    const bool res1_plain = a.res1_shift == 0 && a.res1_off_w == 0 && a.res1_H == Ho2 && a.res1_W == Wo2;
    return a.up_log2 == 0 && (a.Cout & 3) == 0 && !a.out_nchw && (!a.res1 || res1_plain) && a.relu <= PP_RELU_FIRST &&
           (a.y_stride == 0 || a.y_stride == a.Cout);
}

int pp_conv_prepare(const ConvArgs& a) {
    if (a.Cin % 4 != 0 || a.K <= 0) return PP_OK;
    return tap_table_for(a, true) ? PP_OK : PP_ERR_HIP;
}

int pp_launch_conv(const ConvArgs& a_in, hipStream_t stream) {
    ConvArgs a = a_in;
    a.tap_table = nullptr;
    magic_u32((unsigned)a.HWout, &a.div_hw_m, &a.div_hw_s1, &a.div_hw_s2);
    magic_u32((unsigned)a.Wout, &a.div_w_m, &a.div_w_s1, &a.div_w_s2);
    magic_u32((unsigned)a.Cin, &a.div_c_m, &a.div_c_s1, &a.div_c_s2);
    magic_u32((unsigned)a.KW, &a.div_kw_m, &a.div_kw_s1, &a.div_kw_s2);
    static const int xcd_remap = env_int("POSEPIPE_CONV_XCD", 1);
    a.xcd_remap = xcd_remap;
    if (a.Cin % 4 != 0) {
        pp_set_error("conv: Cin=%d must be a multiple of 4 (pad the input channels)", a.Cin);
        return PP_ERR_ARG;
    }
    if (a.Hin + a.pad_h >= 32768 || a.Win + a.pad_w >= 32768) {
        pp_set_error("conv: spatial dims must be < 32768");
        return PP_ERR_ARG;
    }
    if (a.M <= 0) return PP_OK;
    // The pixel loader addresses the input through one raw buffer descriptor (32-bit byte offsets, out-of-range
    // offsets read as zero), so one launch covers < 4 GiB of input: larger batches are cut into image ranges.
    const size_t img_bytes = (size_t)(a.Hin + a.x_pad) * (a.Win + a.x_pad) * a.Cin * sizeof(float);
    static const size_t max_bytes_env = (size_t)env_int("POSEPIPE_CONV_MAX_MB", 0) << 20;   // test knob for the split below
    const size_t max_bytes = max_bytes_env ? max_bytes_env : 0xfffffff0u;
    if (img_bytes > max_bytes) {
        pp_set_error("conv: one input image of %zu bytes exceeds the 4 GiB buffer range", img_bytes);
        return PP_ERR_ARG;
    }
    if ((size_t)a.N * img_bytes > max_bytes) {
        const int per = (int)(max_bytes / img_bytes);
        const size_t y_img = (size_t)((a.Hout << a.up_log2) + a.y_pad) * ((a.Wout << a.up_log2) + a.y_pad) * (a.y_stride ? a.y_stride : a.Cout);
        const size_t r1_img = (size_t)(a.res1_H + a.r1_pad) * (a.res1_W + a.r1_pad) * a.Cout;
        const size_t r2_img = (size_t)((a.Hout << a.up_log2) + a.r2_pad) * ((a.Wout << a.up_log2) + a.r2_pad) * a.Cout;
        for (int n0 = 0; n0 < a.N; n0 += per) {
            ConvArgs p = a_in;
            p.N = std::min(per, a.N - n0);
            p.M = p.N * a.HWout;
            p.x = a.x + (size_t)n0 * (img_bytes / sizeof(float));
            p.y = a.y + (size_t)n0 * y_img;
            if (a.res1) p.res1 = a.res1 + (size_t)n0 * r1_img;
            if (a.res2) p.res2 = a.res2 + (size_t)n0 * r2_img;
            if (a.x_amax) p.x_amax = a.x_amax + n0;
            if (a.y_amax) p.y_amax = a.y_amax + n0;
            const int rc = pp_launch_conv(p, stream);
            if (rc != PP_OK) return rc;
        }
        return PP_OK;
    }
    a.x_bytes = (unsigned)((size_t)a.N * img_bytes);
    if (a.y_stride == 0) a.y_stride = a.Cout;
    // an op of a net: the split kernel exactly where the net built split weights at creation (nothing is decided at launch time);
    // single-op API (numerics 0): the process-wide setting, with a temporary split copy of the weights
    if (a.numerics == PP_NET_NUMERICS_SPLIT && a.wsplit) return pp_launch_conv_split(a, stream);
    if (a.numerics == 0 && pp_conv_split_enabled() && pp_conv_split_eligible(a)) {
        if (a.wsplit) return pp_launch_conv_split(a, stream);
        a.split_f16 = pp_conv_split_f16_default();
        void* tmp = nullptr;                 // single-op API: split the weights for this call
        const size_t wbytes = (pp_conv_split_bytes(a) + 255) / 256 * 256;
        PP_HIP_CHECK(hipMalloc(&tmp, wbytes + (a.split_f16 ? (size_t)a.N * sizeof(unsigned) : 0)));
        a.wsplit = tmp;
        int rc = pp_conv_split_weights(a, tmp, stream);
        if (rc == PP_OK && a.split_f16) {    // ... and take the per-sample maxima of the input (a net tracks them where the tensor is produced)
            unsigned* am = reinterpret_cast<unsigned*>(static_cast<unsigned char*>(tmp) + wbytes);
            if (hipMemsetAsync(am, 0, (size_t)a.N * sizeof(unsigned), stream) != hipSuccess) rc = PP_ERR_HIP;
            if (rc == PP_OK) rc = pp_launch_amax(a.x, a.N, img_bytes / sizeof(float), am, stream);
            a.x_amax = am;
        }
        if (rc == PP_OK) rc = pp_launch_conv_split(a, stream);
        (void)hipStreamSynchronize(stream);
        (void)hipFree(tmp);
        return rc;
    }
    if (a.y_amax && !pp_conv_tracks_amax(a, false)) {
        pp_set_error("conv: this layer's epilogue does not track the output maximum (ConvArgs::y_amax; see pp_conv_tracks_amax)");
        return PP_ERR_STATE;
    }
    static const int force_ct = env_int("POSEPIPE_CONV_CT", 0), force_pt = env_int("POSEPIPE_CONV_PT", 0),
                     min_blocks = env_int("POSEPIPE_CONV_MIN_BLOCKS", 512);
    const int tiles = a.CoutPad / 16;
    // pick the channel tile that wastes the fewest 16-wide tiles; ties go to the larger tile
    int best_ct = 1, best_waste = 1 << 30;
    for (int ct = 4; ct >= 1; --ct) {
        const int waste = ((tiles + ct - 1) / ct) * ct - tiles;
        if (waste < best_waste) {
            best_waste = waste;
            best_ct = ct;
        }
    }
    static const TuneTable tuning;
    static const int env_variant = env_int("POSEPIPE_CONV_VARIANT", -1);
    int tuned_pt = 0, variant = 0;       // built-in default: the two-barrier kernel
    if (!tuning.best.empty()) {
        TuneKey k;
        memset(&k, 0, sizeof(k));
        k.cin = a.Cin; k.cout = a.Cout; k.kh = a.KH; k.kw = a.KW; k.stride = a.stride; k.dil_h = a.dil_h; k.dil_w = a.dil_w;
        if (const TuneTable::Choice* c = tuning.find(k, a.M)) {
            best_ct = c->ct;
            tuned_pt = c->pt;
            if (c->variant >= 0) variant = c->variant;
        }
    }
    if (env_variant >= 0) variant = env_variant;
    if (g_variant >= 0) variant = g_variant;
    if (variant == 4) variant = 0;           // not eligible for the split kernel: built-in exact default
    if (force_ct) best_ct = force_ct;
    if (g_force_ct) best_ct = g_force_ct;
    const int cblocks = (tiles + best_ct - 1) / best_ct;
    // pixel tile: 128 pixels (PT=2: 4 waves/SIMD by registers) when that still gives >= ~2 blocks per CU,
    // else 64.  PT=4 halves the occupancy and measured 20 % slower on HRNet-W48, so it is never chosen.
    int pt = 2;
    while (pt > 1 && (long)((a.M + 64 * pt - 1) / (64 * pt)) * cblocks < min_blocks) pt >>= 1;
    if (tuned_pt) pt = tuned_pt;
    if (force_pt) pt = force_pt;
    if (g_force_pt) pt = g_force_pt;
    {
        // experiment knob: POSEPIPE_CONV_PT_CT<ct>=<pt> forces the pixel tile for one channel-tile class
        static const int pt_by_ct[5] = {0, env_int("POSEPIPE_CONV_PT_CT1", 0), env_int("POSEPIPE_CONV_PT_CT2", 0),
                                        env_int("POSEPIPE_CONV_PT_CT3", 0), env_int("POSEPIPE_CONV_PT_CT4", 0)};
        if (pt_by_ct[best_ct] && (long)((a.M + 64 * pt_by_ct[best_ct] - 1) / (64 * pt_by_ct[best_ct])) * cblocks >= min_blocks)
            pt = pt_by_ct[best_ct];
    }
    a.no_bounds = 0;
    const bool halo = (a.x_pad | a.y_pad | a.r1_pad | a.r2_pad) != 0;
    if (halo) variant = 3;                   // only the pipelined tap-table kernel knows the halo layout
    if (variant == 3) {
        // every tap of every output stays inside the image or inside its zero halo (x_pad columns / rows at the right / below;
        // left / top overshoot lands in the previous row's / image's halo)
        a.no_bounds = a.pad_h <= a.x_pad && a.pad_w <= a.x_pad &&
                      (a.Hout - 1) * a.stride - a.pad_h + (a.KH - 1) * a.dil_h <= a.Hin - 1 + a.x_pad &&
                      (a.Wout - 1) * a.stride - a.pad_w + (a.KW - 1) * a.dil_w <= a.Win - 1 + a.x_pad && a.x_bytes < 0x7fffff00u;
        a.tap_table = tap_table_for(a, true);
        if (!a.tap_table) {
            pp_set_error("conv: could not build the tap table");
            return PP_ERR_HIP;
        }
    }
    if (variant >= 1) return pp_launch_conv_p3(a, best_ct, pt, stream, variant == 2);
    switch (best_ct) {
        case 4: return launch_ct<4>(a, pt, stream);
        case 3: return launch_ct<3>(a, pt, stream);
        case 2: return launch_ct<2>(a, pt, stream);
        default: return launch_ct<1>(a, pt, stream);
    }
}

int pp_launch_maxpool(const PoolArgs& a, hipStream_t stream) {
    if (a.C % 4 != 0) {
        pp_set_error("maxpool: C=%d must be a multiple of 4", a.C);
        return PP_ERR_ARG;
    }
    const size_t total = (size_t)a.N * a.Hout * a.Wout * (a.C / 4);
    if (total == 0) return PP_OK;
    const int blocks = (int)std::min<size_t>((total + 255) / 256, 256 * 8);
    hipLaunchKernelGGL(maxpool_kernel, dim3(blocks), dim3(256), 0, stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        pp_set_error("maxpool launch failed: %s", hipGetErrorString(e));
        return PP_ERR_HIP;
    }
    return PP_OK;
}
