// bf16 GEMM on the gfx950 matrix cores (v_mfma_f32_16x16x32_bf16), the contraction of the ViTPose encoder
// (BASELINE.json configs[4]: "ViTPose-H backbone (bf16 MFMA path)").
//
//   C[M][N] = epilogue( A[M][K] . W[N][K]^T )        A, W bf16 row-major (torch nn.Linear weight layout), fp32 accumulate
//   epilogue: + bias[N] -> GELU (erf form, optional) -> + res[m % res_mod][N] (fp32, optional) -> fp32 or bf16 store
//
// Structure: 256 threads = 2 x 2 waves, block tile (32 WM) x (32 WN), K step 64.  Both operand tiles go global -> LDS
// with global_load_lds_dwordx4 (no staging registers); the LDS image is lane-linear, so the bank-conflict swizzle
// (16-byte chunk index ^ ((row >> 1) & 7) inside each 128-byte row) is applied to the per-lane SOURCE address and
// again when the fragments are read back with ds_read_b128.  The weights are the MFMA "A" operand and the activations
// the "B" operand, so a lane ends up with 4 consecutive n of one row m: bias / residual / store are 16-byte (8-byte
// for bf16) vector accesses.  Tiles are ordered in groups of 8 tile rows x all tile columns per sweep so that the
// blocks resident on one XCD share operand tiles in its L2.
#include "pp_internal.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

namespace {

__device__ __forceinline__ unsigned short f32_to_bf16_rne(float f) {
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

template <int WM, int WN>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(GemmArgs a) {
    constexpr int BM = WM * 32, BN = WN * 32;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[(BM + BN) * 128];
    unsigned char* ldsA = lds;              // activations tile [BM][64] bf16 (swizzled)
    unsigned char* ldsB = lds + BM * 128;   // weights tile     [BN][64] bf16 (swizzled)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // tile order: XCD-contiguous ranges (blocks are dealt round-robin to the 8 XCDs), inside a range groups of 8 tile
    // rows sweep the tile columns
    const int tiles_n = a.N / BN, tiles_m = (a.M + BM - 1) / BM;
    int pid = blockIdx.x;
    {
        const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = pid & 7, loc = pid >> 3;
        pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    constexpr int GM = 8;
    const int per_group = GM * tiles_n;
    const int grp = pid / per_group, in_grp = pid - grp * per_group;
    const int rows_here = min(GM, tiles_m - grp * GM);
    const int tn = in_grp / rows_here, tm = grp * GM + (in_grp - tn * rows_here);
    const int m0 = tm * BM, n0 = tn * BN;

    // global_load_lds sources: instruction i of wave w fills the 1 KiB slab (i * 4 + w) = 8 tile rows
    const int lrow = lane >> 3, slot = lane & 7;
    const __bf16* gA[BM / 32];
    const __bf16* gB[BN / 32];
#pragma unroll
    for (int i = 0; i < BM / 32; ++i) {
        const int row = (i * 4 + wave) * 8 + lrow;
        const int chunk = slot ^ ((row >> 1) & 7);
        gA[i] = reinterpret_cast<const __bf16*>(a.A) + (size_t)min(m0 + row, a.M - 1) * a.K + chunk * 8;
    }
#pragma unroll
    for (int i = 0; i < BN / 32; ++i) {
        const int row = (i * 4 + wave) * 8 + lrow;
        const int chunk = slot ^ ((row >> 1) & 7);
        gB[i] = reinterpret_cast<const __bf16*>(a.B) + (size_t)(n0 + row) * a.K + chunk * 8;
    }

    f32x4_t acc[WM][WN];
#pragma unroll
    for (int mi = 0; mi < WM; ++mi)
#pragma unroll
        for (int ni = 0; ni < WN; ++ni) acc[mi][ni] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int r16 = lane & 15, kg = lane >> 4, sw = (r16 >> 1) & 7;
    const unsigned char* rdA = ldsA + (wm * WM * 16 + r16) * 128;
    const unsigned char* rdB = ldsB + (wn * WN * 16 + r16) * 128;
    const int c0 = ((0 + kg) ^ sw) << 4, c1 = ((4 + kg) ^ sw) << 4;

    const int nk = a.K >> 6;
    for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
        for (int i = 0; i < BM / 32; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gA[i],
                                             (__attribute__((address_space(3))) void*)(ldsA + (i * 4 + wave) * 1024), 16, 0, 0);
            gA[i] += 64;
        }
#pragma unroll
        for (int i = 0; i < BN / 32; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gB[i],
                                             (__attribute__((address_space(3))) void*)(ldsB + (i * 4 + wave) * 1024), 16, 0, 0);
            gB[i] += 64;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int co = kk ? c1 : c0;
            bf16x8_t fa[WM], fb[WN];
#pragma unroll
            for (int mi = 0; mi < WM; ++mi) fa[mi] = *reinterpret_cast<const bf16x8_t*>(rdA + mi * 2048 + co);
#pragma unroll
            for (int ni = 0; ni < WN; ++ni) fb[ni] = *reinterpret_cast<const bf16x8_t*>(rdB + ni * 2048 + co);
#pragma unroll
            for (int mi = 0; mi < WM; ++mi)
#pragma unroll
                for (int ni = 0; ni < WN; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[ni], fa[mi], acc[mi][ni], 0, 0, 0);
        }
        __syncthreads();
    }

    // epilogue: lane holds C[m][n .. n + 3], m = row16 index (lane & 15), n = 4 * (lane >> 4)
#pragma unroll
    for (int mi = 0; mi < WM; ++mi) {
        const int m = m0 + (wm * WM + mi) * 16 + r16;
        if (m >= a.M) continue;
        const int rrow = a.res_mod > 0 ? m % a.res_mod : m;
#pragma unroll
        for (int ni = 0; ni < WN; ++ni) {
            const int n = n0 + (wn * WN + ni) * 16 + kg * 4;
            f32x4_t v = acc[mi][ni];
            if (a.bias) {
                const float4 b = *reinterpret_cast<const float4*>(a.bias + n);
                v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
            }
            if (a.act == 1) {
                v[0] = gelu_erf(v[0]); v[1] = gelu_erf(v[1]); v[2] = gelu_erf(v[2]); v[3] = gelu_erf(v[3]);
            }
            if (a.res) {
                const float4 rr = *reinterpret_cast<const float4*>(a.res + (size_t)rrow * a.N + n);
                v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
            }
            if (a.out_bf16) {
                ushort4 o;
                o.x = f32_to_bf16_rne(v[0]); o.y = f32_to_bf16_rne(v[1]);
                o.z = f32_to_bf16_rne(v[2]); o.w = f32_to_bf16_rne(v[3]);
                *reinterpret_cast<ushort4*>(reinterpret_cast<unsigned short*>(a.C) + (size_t)m * a.N + n) = o;
            } else {
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.C) + (size_t)m * a.N + n) =
                    make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    }
}

__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ x, unsigned short* __restrict__ y, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) y[i] = f32_to_bf16_rne(x[i]);
}

}  // namespace

int pp_launch_f32_to_bf16(const float* x, void* y, size_t n, hipStream_t stream) {
    if (n == 0) return PP_OK;
    const unsigned grid = (unsigned)std::min<size_t>((n + 255) / 256, 65536);
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(grid), dim3(256), 0, stream, x, reinterpret_cast<unsigned short*>(y), n);
    PP_HIP_CHECK(hipGetLastError());
    return PP_OK;
}

int pp_launch_gemm_bf16(const GemmArgs& a, hipStream_t stream) {
    PP_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "gemm_bf16: empty problem");
    PP_REQUIRE(a.K % 64 == 0, "gemm_bf16: K = %d must be a multiple of 64", a.K);
    PP_REQUIRE(a.N % 128 == 0, "gemm_bf16: N = %d must be a multiple of 128", a.N);
    const int tiles_n = a.N / 128;
    const char* env_tile = getenv("POSEPIPE_GEMM_TILE");   // tests / A-B runs: force the 128- or 256-row tile
    const int variant = env_tile ? atoi(env_tile) : 0;
    // 256-row tiles when they still fill the chip twice over
    const bool big = variant == 256 || (variant == 0 && (long)((a.M + 255) / 256) * tiles_n >= 1024);
    if (big) {
        const int tiles_m = (a.M + 255) / 256;
        hipLaunchKernelGGL((gemm_bf16_kernel<8, 4>), dim3(tiles_m * tiles_n), dim3(256), 0, stream, a);
    } else {
        const int tiles_m = (a.M + 127) / 128;
        hipLaunchKernelGGL((gemm_bf16_kernel<4, 4>), dim3(tiles_m * tiles_n), dim3(256), 0, stream, a);
    }
    PP_HIP_CHECK(hipGetLastError());
    return PP_OK;
}
