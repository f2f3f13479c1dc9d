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
// Layout: activations NHWC fp32 (Cin % 4 == 0), weights [Kpad][CoutPad] (both padded to 16 with
// zeros), bias [CoutPad].  LDS tiles are k-major ([k][cout], [k][pixel]) with a row stride that
// shifts consecutive k rows by 16 banks, so the ds_read_b32 operand fetches (lanes 0-15: k,
// lanes 16-31: k+1, ...) are conflict-free.
#include "pp_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BK = 16;

template <int CT, int PT>
struct Tile {
    static constexpr int BC = 16 * CT;           // output channels per block
    static constexpr int BP = 64 * PT;           // pixels per block (4 waves x PT x 16)
    static constexpr int WS = BC + ((BC % 32 == 16) ? 0 : 16);
    static constexpr int XS = BP + 16;
    static constexpr int LDS_FLOATS = BK * WS + BK * XS;
};

template <int CT, int PT>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs a) {
    using T = Tile<CT, PT>;
    constexpr int BC = T::BC, BP = T::BP, WS = T::WS, XS = T::XS;
    __shared__ __attribute__((aligned(16))) float smem[T::LDS_FLOATS];
    float* Ws = smem;
    float* Xs = smem + BK * WS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int m0 = blockIdx.x * BP;
    const int c0 = blockIdx.y * BC;

    // ---- X loader role: one pixel, PT k-quads per thread ---------------------------------
    const int lp = tid % BP;
    const int kq0 = (tid / BP) * PT;
    const int lm = m0 + lp;
    const bool lvalid = lm < a.M;
    int ln = 0, lho = 0, lwo = 0;
    if (lvalid) {
        ln = lm / a.HWout;
        const int rem = lm - ln * a.HWout;
        lho = rem / a.Wout;
        lwo = rem - lho * a.Wout;
    }
    const int hi0 = lho * a.stride - a.pad_h;
    const int wi0 = lwo * a.stride - a.pad_w;
    const float* xn = a.x + (size_t)ln * a.Hin * a.Win * a.Cin;

    // (kh, kw, c) of each of this thread's k-quads for the current chunk
    int qkh[PT], qkw[PT], qc[PT];
#pragma unroll
    for (int i = 0; i < PT; ++i) {
        int k4 = 4 * (kq0 + i);
        int tap = k4 / a.Cin;
        qc[i] = k4 - tap * a.Cin;
        qkh[i] = tap / a.KW;
        qkw[i] = tap - qkh[i] * a.KW;
    }

    // ---- W loader role: one float4 per thread (threads < 4*BC active) ----------------------
    const int wk = tid / (BC / 4);
    const int wc4 = tid % (BC / 4);
    const bool wactive = (wk < BK) && (c0 + 4 * wc4 < a.CoutPad);
    const float* wsrc = a.w + (size_t)wk * a.CoutPad + c0 + 4 * wc4;

    float4 xr[PT];
    float4 wr;

    auto load_chunk = [&](int k0) {
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            const int hi = hi0 + qkh[i] * a.dil_h;
            const int wi = wi0 + qkw[i] * a.dil_w;
            const bool ok = lvalid && (k0 + 4 * (kq0 + i) < a.K) && (unsigned)hi < (unsigned)a.Hin &&
                            (unsigned)wi < (unsigned)a.Win;
            if (ok) {
                xr[i] = *reinterpret_cast<const float4*>(xn + ((size_t)hi * a.Win + wi) * a.Cin + qc[i]);
            } else {
                xr[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            // advance this quad by BK for the next chunk
            qc[i] += BK;
            while (qc[i] >= a.Cin) {
                qc[i] -= a.Cin;
                if (++qkw[i] == a.KW) {
                    qkw[i] = 0;
                    ++qkh[i];
                }
            }
        }
        if (wactive) {
            wr = *reinterpret_cast<const float4*>(wsrc + (size_t)k0 * a.CoutPad);
        } else {
            wr = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };

    auto store_chunk = [&]() {
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            float* dst = Xs + (4 * (kq0 + i)) * XS + lp;
            dst[0] = xr[i].x;
            dst[XS] = xr[i].y;
            dst[2 * XS] = xr[i].z;
            dst[3 * XS] = xr[i].w;
        }
        if (wk < BK) {
            *reinterpret_cast<float4*>(Ws + wk * WS + 4 * wc4) = wr;
        }
    };

    f32x4 acc[CT][PT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) acc[ct][pt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int lrow = lane >> 4;   // k within the MFMA step
    const int lcol = lane & 15;   // cout (A) / pixel (B) within the 16-tile
    const float* wrd = Ws + lrow * WS + lcol;
    const float* xrd = Xs + lrow * XS + wave * (16 * PT) + lcol;

    load_chunk(0);
    for (int k0 = 0; k0 < a.Kpad; k0 += BK) {
        __syncthreads();
        store_chunk();
        __syncthreads();
        if (k0 + BK < a.Kpad) load_chunk(k0 + BK);
#pragma unroll
        for (int s = 0; s < BK / 4; ++s) {
            float av[CT], bv[PT];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) av[ct] = wrd[(4 * s) * WS + ct * 16];
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) bv[pt] = xrd[(4 * s) * XS + pt * 16];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int pt = 0; pt < PT; ++pt)
                    acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ct], bv[pt], acc[ct][pt], 0, 0, 0);
        }
    }

    // ---- epilogue: bias, residuals, ReLU, (upsampled / NCHW) store ----------------------------
    const int up = a.up_log2;
    const int f = 1 << up;
    const int Ho2 = a.Hout << up, Wo2 = a.Wout << up;   // dims of the out buffer
    const bool vec4 = ((a.Cout & 3) == 0) && !a.out_nchw;
    const bool res1_plain = (a.res1_shift == 0 && a.res1_off_w == 0 && a.res1_H == Ho2 && a.res1_W == Wo2);
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const int m = m0 + wave * (16 * PT) + pt * 16 + lcol;
        if (m >= a.M) continue;
        const int n = m / a.HWout;
        const int rem = m - n * a.HWout;
        const int ho = rem / a.Wout;
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
                        *reinterpret_cast<float4*>(a.y + opix * a.Cout + co) = make_float4(o[0], o[1], o[2], o[3]);
                    } else {
                        float* yp = a.y + opix * a.Cout + co;
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
    using T = Tile<CT, PT>;
    dim3 grid((a.M + T::BP - 1) / T::BP, (a.CoutPad + T::BC - 1) / T::BC);
    hipLaunchKernelGGL((conv_igemm_kernel<CT, PT>), grid, dim3(256), 0, stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        pp_set_error("conv_igemm launch failed: %s", hipGetErrorString(e));
        return PP_ERR_HIP;
    }
    return PP_OK;
}

template <int CT>
int launch_ct(const ConvArgs& a, int pt, hipStream_t stream) {
    switch (pt) {
        case 4: return launch_t<CT, 4>(a, stream);
        case 2: return launch_t<CT, 2>(a, stream);
        default: return launch_t<CT, 1>(a, stream);
    }
}

// ---- max pool (NHWC, 4 channels per thread) ---------------------------------------------------
__global__ __launch_bounds__(256) void maxpool_kernel(PoolArgs a) {
    const int c4n = a.C >> 2;
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
                    a.x + (((size_t)n * a.Hin + hi) * a.Win + wi) * a.C + 4 * c4);
                m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
            }
        }
        *reinterpret_cast<float4*>(a.y + (((size_t)n * a.Hout + ho) * a.Wout + wo) * a.C + 4 * c4) = m;
    }
}

}  // namespace

int pp_conv_out_dim(int in, int k, int stride, int pad, int dil) {
    return (in + 2 * pad - dil * (k - 1) - 1) / stride + 1;
}

int pp_launch_conv(const ConvArgs& a, hipStream_t stream) {
    if (a.Cin % 4 != 0) {
        pp_set_error("conv: Cin=%d must be a multiple of 4 (pad the input channels)", a.Cin);
        return PP_ERR_ARG;
    }
    if (a.M <= 0) return PP_OK;
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
    const int cblocks = (tiles + best_ct - 1) / best_ct;
    // pixel tile: keep >= ~2 blocks per CU in flight when the problem allows it
    int pt = 4;
    while (pt > 1 && (long)((a.M + 64 * pt - 1) / (64 * pt)) * cblocks < 512) pt >>= 1;
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
