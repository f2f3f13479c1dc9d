// NV12 (Y plane + interleaved half-resolution UV plane) -> BGR u8 on the device: the frame-source side of the hot path.
// The reference reads BGR frames out of cv2.VideoCapture (pose_pipeline/pipeline.py:47-87 get_robust_reader,
// wrappers/mmtrack.py:38-45, wrappers/mmpose.py:55-75); a decoder's native output is NV12, half the bytes of BGR, so the
// drop-in uploads NV12 and converts here, bit for bit as OpenCV's cvtColor(COLOR_YUV2BGR_NV12) does (imgproc color_yuv:
// ITU-R BT.601 limited range, 20-bit fixed point -- restated in oracle/nv12.py).
//
// HBM-bound byte work (1.5 B read + 3 B written per pixel): one thread converts a 4 x 2 pixel block -- two 4-byte Y loads, one
// 4-byte UV load (two chroma pairs), two 12-byte stores; consecutive threads cover consecutive 4-pixel groups of a row pair, so
// every load and store instruction of a wave touches one contiguous span.  Widths with w % 4 == 2 (854 x 480) take 16-bit
// accesses (their rows are only 2-byte aligned) and a 2-pixel last group.
#include "pp_internal.h"

namespace {

constexpr int CY = 1220542, CUB = 2116026, CUG = -409993, CVG = -852492, CVR = 1673527, SHIFT = 20;

__device__ __forceinline__ unsigned sat8(int v) { return (unsigned)min(max(v >> SHIFT, 0), 255); }

__global__ __launch_bounds__(256) void nv12_to_bgr_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                                          int frames, int h, int w) {
    const int groups = (w + 3) >> 2;                             // 4-pixel groups per row; the last one has 2 pixels when w % 4 == 2
    const long long total = (long long)frames * (h >> 1) * groups;
    const long long stride = (long long)gridDim.x * 256;
    const size_t frame_in = (size_t)h * w * 3 / 2, frame_out = (size_t)h * w * 3;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        const int g = (int)(i % groups);
        const long long r = i / groups;
        const int yp = (int)(r % (h >> 1)), f = (int)(r / (h >> 1));
        const unsigned char* fin = src + (size_t)f * frame_in;
        const unsigned char* puv = fin + (size_t)h * w + (size_t)yp * w + 4 * g;
        const unsigned char* py0 = fin + (size_t)(2 * yp) * w + 4 * g;
        const unsigned char* py1 = py0 + w;
        const bool full = 4 * g + 4 <= w;                        // rows are 2-byte aligned only when w % 4 == 2: 16-bit loads there
        unsigned uv, y0, y1;
        if (full && (w & 3) == 0) {
            uv = *reinterpret_cast<const unsigned*>(puv);
            y0 = *reinterpret_cast<const unsigned*>(py0);
            y1 = *reinterpret_cast<const unsigned*>(py1);
        } else {
            uv = *reinterpret_cast<const unsigned short*>(puv);
            y0 = *reinterpret_cast<const unsigned short*>(py0);
            y1 = *reinterpret_cast<const unsigned short*>(py1);
            if (full) {
                uv |= (unsigned)*reinterpret_cast<const unsigned short*>(puv + 2) << 16;
                y0 |= (unsigned)*reinterpret_cast<const unsigned short*>(py0 + 2) << 16;
                y1 |= (unsigned)*reinterpret_cast<const unsigned short*>(py1 + 2) << 16;
            }
        }
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
            unsigned char* d8 = dst + (size_t)f * frame_out + ((size_t)(2 * yp + row) * w + 4 * g) * 3;
            if (full && (w & 3) == 0) {                          // 12 bytes at a 4-byte aligned address
                unsigned* d = reinterpret_cast<unsigned*>(d8);
                d[0] = o[0] | (o[1] << 8) | (o[2] << 16) | ((unsigned)o[3] << 24);
                d[1] = o[4] | (o[5] << 8) | (o[6] << 16) | ((unsigned)o[7] << 24);
                d[2] = o[8] | (o[9] << 8) | (o[10] << 16) | ((unsigned)o[11] << 24);
            } else {                                             // w % 4 == 2: rows start at 2-byte aligned addresses
                unsigned short* d = reinterpret_cast<unsigned short*>(d8);
                const int n16 = full ? 6 : 3;
                for (int q = 0; q < n16; ++q) d[q] = (unsigned short)(o[2 * q] | (o[2 * q + 1] << 8));
            }
        }
    }
}

}  // namespace

int pp_launch_nv12_to_bgr(const unsigned char* nv12, unsigned char* bgr, int frames, int h, int w, hipStream_t stream) {
    PP_REQUIRE(frames >= 0 && h > 0 && w > 0, "nv12_to_bgr: bad shape %d x %d x %d", frames, h, w);
    PP_REQUIRE(h % 2 == 0 && w % 2 == 0, "nv12_to_bgr: height %d and width %d must be even", h, w);
    if (frames == 0) return PP_OK;
    const long long total = (long long)frames * (h / 2) * ((w + 3) / 4);
    const unsigned grid = (unsigned)std::min<long long>((total + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(nv12_to_bgr_kernel, dim3(grid), dim3(256), 0, stream, nv12, bgr, frames, h, w);
    PP_HIP_CHECK(hipGetLastError());
    return PP_OK;
}
