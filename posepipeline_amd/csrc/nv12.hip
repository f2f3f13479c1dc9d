// NV12 (Y plane + interleaved half-resolution UV plane) -> BGR u8 on the device: the frame-source side of the hot path.
// The reference reads BGR frames out of cv2.VideoCapture (pose_pipeline/pipeline.py:47-87 get_robust_reader,
// wrappers/mmtrack.py:38-45, wrappers/mmpose.py:55-75); a decoder's native output is NV12, half the bytes of BGR, so the
// drop-in uploads NV12 and converts here, bit for bit as OpenCV's cvtColor(COLOR_YUV2BGR_NV12) does (imgproc color_yuv:
// ITU-R BT.601 limited range, 20-bit fixed point -- restated in oracle/nv12.py).
//
// HBM-bound byte work (1.5 B read + 3 B written per pixel): one thread converts a 4 x 2 pixel block -- two 4-byte Y loads, one
// 4-byte UV load (two chroma pairs), two 12-byte stores; consecutive threads cover consecutive 4-pixel groups of a row pair, so
// every load and store instruction of a wave touches one contiguous span.
#include "pp_internal.h"

namespace {

constexpr int CY = 1220542, CUB = 2116026, CUG = -409993, CVG = -852492, CVR = 1673527, SHIFT = 20;

__device__ __forceinline__ unsigned sat8(int v) { return (unsigned)min(max(v >> SHIFT, 0), 255); }

__global__ __launch_bounds__(256) void nv12_to_bgr_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                                          int frames, int h, int w) {
    const int groups = w >> 2;                                   // 4-pixel groups per row
    const long long total = (long long)frames * (h >> 1) * groups;
    const long long stride = (long long)gridDim.x * 256;
    const size_t frame_in = (size_t)h * w * 3 / 2, frame_out = (size_t)h * w * 3;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        const int g = (int)(i % groups);
        const long long r = i / groups;
        const int yp = (int)(r % (h >> 1)), f = (int)(r / (h >> 1));
        const unsigned char* fin = src + (size_t)f * frame_in;
        const unsigned uv = *reinterpret_cast<const unsigned*>(fin + (size_t)h * w + (size_t)yp * w + 4 * g);
        const unsigned y0 = *reinterpret_cast<const unsigned*>(fin + (size_t)(2 * yp) * w + 4 * g);
        const unsigned y1 = *reinterpret_cast<const unsigned*>(fin + (size_t)(2 * yp + 1) * w + 4 * g);
        int ruv[2], guv[2], buv[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int u = (int)((uv >> (16 * p)) & 255) - 128, v = (int)((uv >> (16 * p + 8)) & 255) - 128;
            ruv[p] = (1 << (SHIFT - 1)) + CVR * v;
            guv[p] = (1 << (SHIFT - 1)) + CVG * v + CUG * u;
            buv[p] = (1 << (SHIFT - 1)) + CUB * u;
        }
#pragma unroll
        for (int row = 0; row < 2; ++row) {
            const unsigned yy = row ? y1 : y0;
            unsigned char o[12];
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                const int y = max((int)((yy >> (8 * px)) & 255) - 16, 0) * CY, p = px >> 1;
                o[3 * px + 0] = (unsigned char)sat8(y + buv[p]);
                o[3 * px + 1] = (unsigned char)sat8(y + guv[p]);
                o[3 * px + 2] = (unsigned char)sat8(y + ruv[p]);
            }
            unsigned* d = reinterpret_cast<unsigned*>(dst + (size_t)f * frame_out + ((size_t)(2 * yp + row) * w + 4 * g) * 3);
            d[0] = o[0] | (o[1] << 8) | (o[2] << 16) | ((unsigned)o[3] << 24);
            d[1] = o[4] | (o[5] << 8) | (o[6] << 16) | ((unsigned)o[7] << 24);
            d[2] = o[8] | (o[9] << 8) | (o[10] << 16) | ((unsigned)o[11] << 24);
        }
    }
}

}  // namespace

int pp_launch_nv12_to_bgr(const unsigned char* nv12, unsigned char* bgr, int frames, int h, int w, hipStream_t stream) {
    PP_REQUIRE(frames >= 0 && h > 0 && w > 0, "nv12_to_bgr: bad shape %d x %d x %d", frames, h, w);
    PP_REQUIRE(h % 2 == 0 && w % 4 == 0, "nv12_to_bgr: height %d must be even and width %d a multiple of 4", h, w);
    if (frames == 0) return PP_OK;
    const long long total = (long long)frames * (h / 2) * (w / 4);
    const unsigned grid = (unsigned)std::min<long long>((total + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(nv12_to_bgr_kernel, dim3(grid), dim3(256), 0, stream, nv12, bgr, frames, h, w);
    PP_HIP_CHECK(hipGetLastError());
    return PP_OK;
}
