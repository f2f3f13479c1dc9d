// PP_OP_DECONV_BF16: ConvTranspose2d(kernel 4, stride 2, padding 1) + bias (folded BN) + ReLU as ONE bf16 MFMA GEMM.
//
// The first deconvolution of the ViTPose head (1280 -> 256 channels on the 16 x 12 token map) is 1.0 GMAC per pass; as four
// fp32 2x2 convolutions it cost 2.2 ms of a 44.5 ms ViTPose-H step.  A transposed convolution is linear in the input
// pixels: output pixel (2j + a, 2i + b) is the sum over the 2 x 2 taps (r, s) of  X[j + a - 1 + r][i + b - 1 + s] . W[a][b][r][s],
// with W[a][b][r][s] = w[:, :, 3 - a - 2r, 3 - b - 2s].  So
//     Y[pixel][(a, b, r, s)][cout] = X[pixel][cin] . W_all[(a, b, r, s)][cout][cin]^T       -- one GEMM, N = 16 cout, no im2col,
//     out[2j + a][2i + b]          = relu((((y00 + y01) + y10) + y11) + bias)                 -- a gather over <= 4 neighbours.
// No multiply is wasted (the 16 blocks are exactly the 16 kernel taps).  X and W are rounded to bf16, the GEMM accumulates
// in fp32, Y stays fp32, and the four partial sums are added in the fixed order above.
#include "pp_internal.h"

#include <memory>

namespace {

__global__ __launch_bounds__(256) void deconv_gather_kernel(const float4* __restrict__ Y, const float4* __restrict__ bias,
                                                            float4* __restrict__ out, size_t total, int H, int W, int c4, int relu) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int cc = (int)(i % c4);
    size_t p = i / c4;
    const int ox = (int)(p % (2 * W));
    p /= 2 * W;
    const int oy = (int)(p % (2 * H));
    const size_t n = p / (2 * H);
    const int a = oy & 1, b = ox & 1, j = oy >> 1, ii = ox >> 1;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    bool first = true;
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int y = j + a - 1 + r, x = ii + b - 1 + s;
            if ((unsigned)y >= (unsigned)H || (unsigned)x >= (unsigned)W) continue;   // zero padding: the term is absent
            const int q = ((a * 2 + b) * 2 + r) * 2 + s;
            const float4 v = Y[(((n * H + y) * W + x) * 16 + q) * c4 + cc];
            if (first) {
                acc = v;
                first = false;
            } else {
                acc.x = __fadd_rn(acc.x, v.x); acc.y = __fadd_rn(acc.y, v.y); acc.z = __fadd_rn(acc.z, v.z); acc.w = __fadd_rn(acc.w, v.w);
            }
        }
    const float4 bb = bias[cc];
    acc.x = __fadd_rn(acc.x, bb.x); acc.y = __fadd_rn(acc.y, bb.y); acc.z = __fadd_rn(acc.z, bb.z); acc.w = __fadd_rn(acc.w, bb.w);
    if (relu) {
        acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f);
    }
    out[i] = acc;
}

}  // namespace

struct pp_deconv_bf16 {
    int h = 0, w = 0, cin = 0, cout = 0, max_batch = 0;
    unsigned short* wbf = nullptr;   // [16 * cout][cin] bf16
    const float* bias = nullptr;     // device pointer into the program's blob
    unsigned short* xbf = nullptr;   // [max_batch * h * w][cin] bf16
    float* y = nullptr;              // [max_batch * h * w][16 * cout] fp32
    ~pp_deconv_bf16() {
        if (wbf) (void)hipFree(wbf);
        if (xbf) (void)hipFree(xbf);
        if (y) (void)hipFree(y);
    }
};

void pp_deconv_bf16_destroy(pp_deconv_bf16* d) { delete d; }

// weights: DEVICE pointer to fp32 W_all [16][cout][cin] (block q = ((a * 2 + b) * 2 + r) * 2 + s), bias: device [cout]
int pp_deconv_bf16_create(const float* weights, const float* bias, int h, int w, int cin, int cout, int max_batch,
                          hipStream_t stream, pp_deconv_bf16** out) {
    PP_REQUIRE(weights && bias && out, "deconv_bf16: NULL argument");
    PP_REQUIRE(cin % 64 == 0 && (16 * cout) % 128 == 0 && (cout & 3) == 0, "deconv_bf16: cin %d must be a multiple of 64, cout %d of 8", cin, cout);
    std::unique_ptr<pp_deconv_bf16> d(new pp_deconv_bf16());
    d->h = h; d->w = w; d->cin = cin; d->cout = cout; d->max_batch = max_batch; d->bias = bias;
    const size_t nw = (size_t)16 * cout * cin, m = (size_t)max_batch * h * w;
    PP_HIP_CHECK(hipMalloc((void**)&d->wbf, nw * 2));
    PP_HIP_CHECK(hipMalloc((void**)&d->xbf, m * cin * 2));
    PP_HIP_CHECK(hipMalloc((void**)&d->y, m * 16 * cout * 4));
    int rc = pp_launch_f32_to_bf16(weights, d->wbf, nw, stream);
    if (rc != PP_OK) return rc;
    if ((rc = pp_gemm_bf16_prepare()) != PP_OK) return rc;
    *out = d.release();
    return PP_OK;
}

// x: [batch][h][w][cin] fp32 -> out: [batch][2h][2w][cout] fp32
int pp_deconv_bf16_run(pp_deconv_bf16* d, const float* x, float* out, int batch, int relu, hipStream_t stream) {
    PP_REQUIRE(d && x && out && batch > 0 && batch <= d->max_batch, "deconv_bf16: bad batch %d", batch);
    const int m = batch * d->h * d->w;
    int rc = pp_launch_f32_to_bf16(x, d->xbf, (size_t)m * d->cin, stream);
    if (rc != PP_OK) return rc;
    GemmArgs g{};
    g.A = d->xbf; g.B = d->wbf; g.C = d->y; g.M = m; g.N = 16 * d->cout; g.K = d->cin;
    if ((rc = pp_launch_gemm_bf16(g, stream)) != PP_OK) return rc;
    const size_t total = (size_t)batch * 4 * d->h * d->w * (d->cout / 4);
    hipLaunchKernelGGL(deconv_gather_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                       reinterpret_cast<const float4*>(d->y), reinterpret_cast<const float4*>(d->bias),
                       reinterpret_cast<float4*>(out), total, d->h, d->w, d->cout / 4, relu);
    PP_HIP_CHECK(hipGetLastError());
    return PP_OK;
}
