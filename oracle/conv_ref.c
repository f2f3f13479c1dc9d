/*
 * TEST INFRASTRUCTURE -- CPU oracle, never shipped, never on the product path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Plain-C restatement of the convolution / pooling arithmetic that the reference reaches in its
 * un-vendored dependencies (torch.nn.Conv2d / Conv1d / MaxPool2d inside mmpose HRNet, mmdet
 * ResNet-50-FPN and VideoPose3D; arch specs
 *   3rdparty/mmpose/config/top_down/darkpose/coco/hrnet_w48_coco_384x288_dark.py:44-79,
 *   3rdparty/mmtracking/_base_/models/faster_rcnn_r50_fpn.py:1-112,
 *   pose_pipeline/wrappers/videopose3d.py:46-50).
 * The reference has no tests or golden vectors for these stages and torch-CPU's own conv uses a
 * blocked summation order, so this file fixes ONE order -- for every output element
 *      acc = 0;  for kh, kw, cin (in that order, zero-padding taps skipped):  acc = fmaf(x, w, acc)
 * -- which is also the order the gfx950 fp32 MFMA accumulates in; the HIP kernel must match it
 * bit for bit.  tests/test_oracle_conv.py pins this file against torch.nn.functional.conv2d on
 * CPU (tolerance, different summation order).
 *
 * Layouts: activations NHWC fp32; weights W[K][cout] with k = (kh*KW + kw)*cin_total + cin;
 * bias[cout].  No BatchNorm here: callers fold it (oracle/nets.py) or apply it separately.
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* number of OpenMP threads the oracle may use (callers pass the usable core count: a container's CPU
 * quota can be far below the number of logical CPUs it sees) */
void oracle_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

static int out_dim(int in, int k, int stride, int pad, int dil) {
    return (in + 2 * pad - dil * (k - 1) - 1) / stride + 1;
}

/* y[n][ho][wo][cout] = bias[co] + sum_k x * w, accumulated as an fmaf chain in k order, bias added
 * last (acc + bias), matching the epilogue of the product kernel. */
void oracle_conv2d_nhwc(const float* x, int n, int hin, int win, int cin, const float* w, const float* bias,
                        int cout, int w_stride, int kh, int kw, int stride, int pad_h, int pad_w, int dil_h,
                        int dil_w, float* y) {
    const int hout = out_dim(hin, kh, stride, pad_h, dil_h);
    const int wout = out_dim(win, kw, stride, pad_w, dil_w);
#pragma omp parallel for collapse(2) schedule(static)
    for (int in_ = 0; in_ < n; ++in_) {
        for (int ho = 0; ho < hout; ++ho) {
            float* acc = (float*)malloc(sizeof(float) * (size_t)cout);
            for (int wo = 0; wo < wout; ++wo) {
                for (int co = 0; co < cout; ++co) acc[co] = 0.0f;
                for (int ikh = 0; ikh < kh; ++ikh) {
                    const int hi = ho * stride - pad_h + ikh * dil_h;
                    if (hi < 0 || hi >= hin) continue;
                    for (int ikw = 0; ikw < kw; ++ikw) {
                        const int wi = wo * stride - pad_w + ikw * dil_w;
                        if (wi < 0 || wi >= win) continue;
                        const float* xp = x + (((size_t)in_ * hin + hi) * win + wi) * cin;
                        const float* wp = w + (size_t)((ikh * kw + ikw) * cin) * w_stride;
                        for (int ci = 0; ci < cin; ++ci) {
                            const float xv = xp[ci];
                            const float* wr = wp + (size_t)ci * w_stride;
                            for (int co = 0; co < cout; ++co) acc[co] = fmaf(xv, wr[co], acc[co]);
                        }
                    }
                }
                float* yp = y + (((size_t)in_ * hout + ho) * wout + wo) * cout;
                for (int co = 0; co < cout; ++co) yp[co] = bias ? acc[co] + bias[co] : acc[co];
            }
            free(acc);
        }
    }
}

/* torch.nn.MaxPool2d semantics (padding behaves as -inf), NHWC */
void oracle_maxpool2d_nhwc(const float* x, int n, int hin, int win, int c, int kh, int kw, int stride, int pad_h,
                           int pad_w, float* y) {
    const int hout = out_dim(hin, kh, stride, pad_h, 1);
    const int wout = out_dim(win, kw, stride, pad_w, 1);
#pragma omp parallel for collapse(2) schedule(static)
    for (int in_ = 0; in_ < n; ++in_) {
        for (int ho = 0; ho < hout; ++ho) {
            for (int wo = 0; wo < wout; ++wo) {
                float* yp = y + (((size_t)in_ * hout + ho) * wout + wo) * c;
                for (int ci = 0; ci < c; ++ci) yp[ci] = -INFINITY;
                for (int ikh = 0; ikh < kh; ++ikh) {
                    const int hi = ho * stride - pad_h + ikh;
                    if (hi < 0 || hi >= hin) continue;
                    for (int ikw = 0; ikw < kw; ++ikw) {
                        const int wi = wo * stride - pad_w + ikw;
                        if (wi < 0 || wi >= win) continue;
                        const float* xp = x + (((size_t)in_ * hin + hi) * win + wi) * c;
                        for (int ci = 0; ci < c; ++ci) yp[ci] = xp[ci] > yp[ci] ? xp[ci] : yp[ci];
                    }
                }
            }
        }
    }
}
