// PP_OP_UPSAMPLE_ADD: out[n][y][x][c] = act((up(t)[n][y][x][c] + res1[n][y][x][c]) + res2[...]),  up = nearest 2^u.
//
// HRNet's fuse layers (mmpose HRModule: 1x1 conv + BN on the coarser branch, nearest upsample, `y += ...`) used to
// scatter each conv result over its 2^u x 2^u patch from inside the conv epilogue.  The conv of a coarse branch has few
// tiles (8x6 maps: 48 blocks for a batch of 128), and each of those blocks then read-modified-wrote up to 64x its own
// output: 0.29 ms for 50 MB.  Here the conv writes its small fp32 result once and this kernel -- one float4 per thread,
// every CU busy, HBM-bound -- does the upsample + add.  The additions are the same fp32 operations in the same order
// as the conv epilogue's ((acc + bias) + res1) + res2, so results are bit-identical.
#include "pp_internal.h"

namespace {

__global__ __launch_bounds__(256) void upsample_add_kernel(const float4* __restrict__ t, const float4* __restrict__ r1,
                                                           const float4* __restrict__ r2, float4* __restrict__ y, size_t total,
                                                           int H, int W, int c4, int up, int relu) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int cc = (int)(i % c4);
    size_t p = i / c4;
    const int x = (int)(p % W);
    p /= W;
    const int yy = (int)(p % H);
    const size_t n = p / H;
    const int hs = H >> up, ws = W >> up;
    float4 v = t[((n * hs + (yy >> up)) * ws + (x >> up)) * c4 + cc];
    if (r1) {
        const float4 a = r1[i];
        v.x = __fadd_rn(v.x, a.x); v.y = __fadd_rn(v.y, a.y); v.z = __fadd_rn(v.z, a.z); v.w = __fadd_rn(v.w, a.w);
    }
    if (r2) {
        const float4 a = r2[i];
        v.x = __fadd_rn(v.x, a.x); v.y = __fadd_rn(v.y, a.y); v.z = __fadd_rn(v.z, a.z); v.w = __fadd_rn(v.w, a.w);
    }
    if (relu) {
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    }
    y[i] = v;
}

}  // namespace

int pp_launch_upsample_add(const float* t, const float* res1, const float* res2, float* y, int n, int H, int W, int c,
                           int up_log2, int relu, hipStream_t stream) {
    PP_REQUIRE(n > 0 && H > 0 && W > 0 && c > 0 && (c & 3) == 0, "upsample_add: c = %d must be a multiple of 4", c);
    PP_REQUIRE(up_log2 >= 0 && up_log2 <= 5 && (H >> up_log2 << up_log2) == H && (W >> up_log2 << up_log2) == W,
               "upsample_add: %dx%d is not a multiple of 2^%d", H, W, up_log2);
    const size_t total = (size_t)n * H * W * (c / 4);
    hipLaunchKernelGGL(upsample_add_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                       reinterpret_cast<const float4*>(t), reinterpret_cast<const float4*>(res1),
                       reinterpret_cast<const float4*>(res2), reinterpret_cast<float4*>(y), total, H, W, c / 4, up_log2, relu);
    PP_HIP_CHECK(hipGetLastError());
    return PP_OK;
}
