// bf16 GEMM on the gfx950 matrix cores (v_mfma_f32_16x16x32_bf16), the contraction of the ViTPose encoder
// (BASELINE.json configs[4]: "ViTPose-H backbone (bf16 MFMA path)").
//
//   C[M][N] = epilogue( A[M][K] . W[N][K]^T )        A, W bf16 row-major (torch nn.Linear weight layout), fp32 accumulate
//   epilogue: + bias[N] -> GELU (erf form, optional) -> + res[m % res_mod][N] (fp32, optional) -> fp32 or bf16 store
//
// Structure: WAVES_M x WAVES_N waves per block, each owning a (16 WM) x (16 WN) sub-tile of the block tile, K step 64.
// Both operand tiles go global -> LDS with global_load_lds_dwordx4 (no staging registers); the LDS image is
// lane-linear, so the bank-conflict swizzle (16-byte chunk index ^ ((row >> 1) & 7) inside each 128-byte row) is
// applied to the per-lane SOURCE address and again when the fragments are read back with ds_read_b128 (measured: 0
// bank-conflict cycles).  The weights are the MFMA "A" operand and the activations the "B" operand, so a lane ends up
// with 4 consecutive n of one row m: bias / residual / store are 16-byte (8-byte for bf16) vector accesses.  Tiles are
// ordered in groups of group_m (4) tile rows x all tile columns per sweep, each XCD owning a contiguous range of that
// order, so that the blocks resident on one XCD share operand tiles in its L2.
// Default: 256 x 256 tiles, 16 waves, two LDS stages with one barrier per K step (launcher at the end of the file lists
// every configuration with its measured rate; DESIGN.md 5b has the bound analysis).
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

// GELU(v) = 0.5 v (1 + erf(v / sqrt 2)); erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below the bf16 rounding
// of the result): 1 rcp + 1 exp + 8 VALU instead of libm's branchy erff -- the epilogue of the fc1 GEMM applies it to
// 64 values per lane.
__device__ __forceinline__ float gelu_erf(float v) {
    const float x = v * 0.70710678118654752440f, ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = __expf(-ax * ax);
    const float erf_ax = fmaf(-p * t, e, 1.0f);
    return 0.5f * v * (1.0f + copysignf(erf_ax, x));
}

// epilogue of one wave tile: lane holds C[m][n .. n + 3], m = mbase + 16 mi + (lane & 15), n = nbase + 16 ni + 4 (lane >> 4)
template <int WM, int WN>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& a, f32x4_t (&acc)[WM][WN], int mbase, int nbase, int r16, int kg) {
#pragma unroll
    for (int mi = 0; mi < WM; ++mi) {
        const int m = mbase + mi * 16 + r16;
        if (m >= a.M) continue;
        const int rrow = a.res_mod > 0 ? m % a.res_mod : m;
#pragma unroll
        for (int ni = 0; ni < WN; ++ni) {
            const int n = nbase + ni * 16 + kg * 4;
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
                size_t off = (size_t)m * a.N + n;
                if (a.qkv_tokens > 0) {      // head-major q / k / v: [which][sample][head][token][d]; 4 | head_dim keeps the 4 values together
                    const int dim = a.N / 3, which = n / dim, rem = n - which * dim, head = rem / a.qkv_hd, d = rem - head * a.qkv_hd;
                    const int smp = m / a.qkv_tokens, tok = m - smp * a.qkv_tokens;
                    off = ((((size_t)which * (a.M / a.qkv_tokens) + smp) * (dim / a.qkv_hd) + head) * a.qkv_tokens + tok) * a.qkv_hd + d;
                }
                *reinterpret_cast<ushort4*>(reinterpret_cast<unsigned short*>(a.C) + off) = o;
            } else {
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.C) + (size_t)m * a.N + n) =
                    make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    }
}

// blocks the register allocator must leave room for on one CU (without it the 256-thread variants spread into AGPRs and
// lose occupancy, which is what hides the load latency here)
constexpr int min_blocks(int nw, int wm, int nstage) {
    return nw >= 16 ? 1 : nw >= 8 ? (wm > 4 ? 1 : 2) : (wm > 4 || nstage > 1) ? 2 : 4;
}

template <int WAVES_M, int WAVES_N, int WM, int WN, int NSTAGE, int BK>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, min_blocks(WAVES_M * WAVES_N, WM, NSTAGE * BK / 64)) void gemm_bf16_kernel(GemmArgs a) {
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int BM = WAVES_M * WM * 16, BN = WAVES_N * WN * 16;
    constexpr int ROWB = BK * 2;             // bytes per tile row in LDS (128 / 64)
    constexpr int CH = BK / 8;               // 16-byte chunks per row (8 / 4)
    constexpr int RS = 1024 / ROWB;          // tile rows per 1 KiB global_load_lds slab (8 / 16)
    constexpr int SH = BK == 64 ? 1 : 2;     // swizzle: chunk ^ ((row >> SH) & (CH - 1)) -> 16 rows x 16 B hit 16 distinct bank groups
    constexpr int NLA = BM / (RS * NW), NLB = BN / (RS * NW);   // global_load_lds per thread and tile
    static_assert(BK == 64 || BK == 32, "K step");
    static_assert(BM % (RS * NW) == 0 && BN % (RS * NW) == 0, "tile rows must split evenly over the waves");
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];   // NSTAGE x (BM + BN) x ROWB bytes
    unsigned char* ldsA = lds;               // activations tile [BM][BK] bf16 (swizzled)
    unsigned char* ldsB = lds + BM * ROWB;   // weights tile     [BN][BK] bf16 (swizzled)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave - wm * WAVES_N;

    // tile order: XCD-contiguous ranges (blocks are dealt round-robin to the 8 XCDs), inside a range groups of group_m
    // tile rows sweep the tile columns
    const int tiles_n = a.N / BN, tiles_m = (a.M + BM - 1) / BM;
    int pid = blockIdx.x;
    {
        const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = pid & 7, loc = pid >> 3;
        pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int GM = a.group_m;
    const int per_group = GM * tiles_n;
    const int grp = pid / per_group, in_grp = pid - grp * per_group;
    const int rows_here = min(GM, tiles_m - grp * GM);
    const int tn = in_grp / rows_here, tm = grp * GM + (in_grp - tn * rows_here);
    const int m0 = tm * BM, n0 = tn * BN;

    // global_load_lds sources: instruction i of wave w fills the 1 KiB slab (i * NW + w) = RS tile rows
    const int lrow = lane / CH, slot = lane & (CH - 1);
    const __bf16* gA[NLA];
    const __bf16* gB[NLB];
#pragma unroll
    for (int i = 0; i < NLA; ++i) {
        const int row = (i * NW + wave) * RS + lrow;
        const int chunk = slot ^ ((row >> SH) & (CH - 1));
        gA[i] = reinterpret_cast<const __bf16*>(a.A) + (size_t)min(m0 + row, a.M - 1) * a.K + chunk * 8;
    }
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
        const int row = (i * NW + wave) * RS + lrow;
        const int chunk = slot ^ ((row >> SH) & (CH - 1));
        gB[i] = reinterpret_cast<const __bf16*>(a.B) + (size_t)(n0 + row) * a.K + chunk * 8;
    }

    f32x4_t acc[WM][WN];
#pragma unroll
    for (int mi = 0; mi < WM; ++mi)
#pragma unroll
        for (int ni = 0; ni < WN; ++ni) acc[mi][ni] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int r16 = lane & 15, kg = lane >> 4, sw = (r16 >> SH) & (CH - 1);
    const unsigned char* rdA = ldsA + (wm * WM * 16 + r16) * ROWB;
    const unsigned char* rdB = ldsB + (wn * WN * 16 + r16) * ROWB;
    const int c0 = ((0 + kg) ^ sw) << 4, c1 = (((4 + kg) & (CH - 1)) ^ sw) << 4;

    // K loop.
    // NSTAGE 1: load, wait, barrier, multiply, barrier -- the load latency is hidden by the other block(s) on the CU.
    // NSTAGE 2: wait for tile kt, ONE barrier (it also says every wave is done reading the other stage), start the loads
    //           of tile kt + 1 into that stage, multiply tile kt while they fly.  Raw s_barrier + explicit s_waitcnt: a
    //           __syncthreads() would make the compiler drain the loads first.
    constexpr int STAGE = (BM + BN) * ROWB;
    auto issue = [&](int stage) {
#pragma unroll
        for (int i = 0; i < NLA; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gA[i],
                                             (__attribute__((address_space(3))) void*)(ldsA + stage * STAGE + (i * NW + wave) * 1024),
                                             16, 0, 0);
            gA[i] += BK;
        }
#pragma unroll
        for (int i = 0; i < NLB; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gB[i],
                                             (__attribute__((address_space(3))) void*)(ldsB + stage * STAGE + (i * NW + wave) * 1024),
                                             16, 0, 0);
            gB[i] += BK;
        }
    };
    const int nk = a.K / BK;
    if constexpr (NSTAGE == 2) issue(0);
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = NSTAGE == 2 ? (kt & 1) : 0;
        if constexpr (NSTAGE == 1) issue(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();          // every wave's part of tile kt has landed
        if constexpr (NSTAGE == 2) {
            if (kt + 1 < nk) issue(cur ^ 1);
        }
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk) {
            const int co = (kk ? c1 : c0) + cur * STAGE;
            bf16x8_t fa[WM], fb[WN];
#pragma unroll
            for (int mi = 0; mi < WM; ++mi) fa[mi] = *reinterpret_cast<const bf16x8_t*>(rdA + mi * 16 * ROWB + co);
#pragma unroll
            for (int ni = 0; ni < WN; ++ni) fb[ni] = *reinterpret_cast<const bf16x8_t*>(rdB + ni * 16 * ROWB + co);
#pragma unroll
            for (int mi = 0; mi < WM; ++mi)
#pragma unroll
                for (int ni = 0; ni < WN; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[ni], fa[mi], acc[mi][ni], 0, 0, 0);
        }
        if constexpr (NSTAGE == 1) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();      // the tile may be overwritten
        }
    }

    gemm_epilogue<WM, WN>(a, acc, m0 + wm * WM * 16, n0 + wn * WN * 16, r16, kg);
}

// Register-pipelined form (configuration 9): 8 waves of 128 x 64, 256 x 256 block tile, two LDS stages.  The fragments
// of the NEXT half K step are read from LDS before the 32 MFMAs of the current one are issued, so no MFMA ever waits for
// an LDS read; the barrier sits in the middle of a K step (after the reads of its second half), where it also frees the
// stage for the tile after next -- global loads fly for one and a half K steps.
template <int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, 1) void gemm_bf16_pipelined_kernel(GemmArgs a) {
    constexpr int WM = 8, WN = 4;
    constexpr int NW = WAVES_M * WAVES_N, BK = 64, ROWB = 128, RS = 8;
    constexpr int BM = WAVES_M * WM * 16, BN = WAVES_N * WN * 16;
    constexpr int NLA = BM / (RS * NW), NLB = BN / (RS * NW);
    constexpr int STAGE = (BM + BN) * ROWB;
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    unsigned char* ldsA = lds;
    unsigned char* ldsB = lds + BM * ROWB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave - wm * WAVES_N;
    const int tiles_n = a.N / BN, tiles_m = (a.M + BM - 1) / BM;
    int pid = blockIdx.x;
    {
        const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = pid & 7, loc = pid >> 3;
        pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int GM = a.group_m, per_group = GM * tiles_n;
    const int grp = pid / per_group, in_grp = pid - grp * per_group;
    const int rows_here = min(GM, tiles_m - grp * GM);
    const int tn = in_grp / rows_here, tm = grp * GM + (in_grp - tn * rows_here);
    const int m0 = tm * BM, n0 = tn * BN;
    const int lrow = lane >> 3, slot = lane & 7;
    const __bf16* gA[NLA];
    const __bf16* gB[NLB];
#pragma unroll
    for (int i = 0; i < NLA; ++i) {
        const int row = (i * NW + wave) * RS + lrow;
        gA[i] = reinterpret_cast<const __bf16*>(a.A) + (size_t)min(m0 + row, a.M - 1) * a.K + (slot ^ ((row >> 1) & 7)) * 8;
    }
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
        const int row = (i * NW + wave) * RS + lrow;
        gB[i] = reinterpret_cast<const __bf16*>(a.B) + (size_t)(n0 + row) * a.K + (slot ^ ((row >> 1) & 7)) * 8;
    }
    auto issue = [&](int stage) {
#pragma unroll
        for (int i = 0; i < NLA; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gA[i],
                                             (__attribute__((address_space(3))) void*)(ldsA + stage * STAGE + (i * NW + wave) * 1024),
                                             16, 0, 0);
            gA[i] += BK;
        }
#pragma unroll
        for (int i = 0; i < NLB; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gB[i],
                                             (__attribute__((address_space(3))) void*)(ldsB + stage * STAGE + (i * NW + wave) * 1024),
                                             16, 0, 0);
            gB[i] += BK;
        }
    };
    const int r16 = lane & 15, kg = lane >> 4, sw = (r16 >> 1) & 7;
    const unsigned char* rdA = ldsA + (wm * WM * 16 + r16) * ROWB;
    const unsigned char* rdB = ldsB + (wn * WN * 16 + r16) * ROWB;
    const int c0 = ((0 + kg) ^ sw) << 4, c1 = ((4 + kg) ^ sw) << 4;

    f32x4_t acc[WM][WN];
#pragma unroll
    for (int mi = 0; mi < WM; ++mi)
#pragma unroll
        for (int ni = 0; ni < WN; ++ni) acc[mi][ni] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    bf16x8_t fa0[WM], fb0[WN], fa1[WM], fb1[WN];
    auto read = [&](bf16x8_t (&fa)[WM], bf16x8_t (&fb)[WN], int off) {
#pragma unroll
        for (int mi = 0; mi < WM; ++mi) fa[mi] = *reinterpret_cast<const bf16x8_t*>(rdA + mi * 16 * ROWB + off);
#pragma unroll
        for (int ni = 0; ni < WN; ++ni) fb[ni] = *reinterpret_cast<const bf16x8_t*>(rdB + ni * 16 * ROWB + off);
    };
    auto mma = [&](bf16x8_t (&fa)[WM], bf16x8_t (&fb)[WN]) {
#pragma unroll
        for (int mi = 0; mi < WM; ++mi)
#pragma unroll
            for (int ni = 0; ni < WN; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[ni], fa[mi], acc[mi][ni], 0, 0, 0);
    };

    const int nk = a.K / BK;
    issue(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (nk > 1) issue(1);
    read(fa0, fb0, c0);                                   // (tile 0, first half)
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = (kt & 1) * STAGE, nxt = STAGE - cur;
        read(fa1, fb1, c1 + cur);                         // second half of this K step
        __builtin_amdgcn_sched_barrier(0);
        mma(fa0, fb0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // own part of tile kt + 1 landed; stage `cur` fully read
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < nk) issue(kt & 1);                   // tile kt + 2 into the stage just freed
        if (kt + 1 < nk) read(fa0, fb0, c0 + nxt);        // first half of the next K step
        __builtin_amdgcn_sched_barrier(0);
        mma(fa1, fb1);
    }
    gemm_epilogue<WM, WN>(a, acc, m0 + wm * WM * 16, n0 + wn * WN * 16, r16, kg);
}

// Persistent form of the two-stage kernel: one block per CU walks the tile list; the first K tile of the NEXT output tile
// is requested (global_load_lds is asynchronous and needs no registers) before the epilogue of the current one, so the
// epilogue's loads / math / stores overlap that fetch instead of being followed by a cold pipeline start.
template <int WAVES_M, int WAVES_N, int WM, int WN>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, 1) void gemm_bf16_persistent_kernel(GemmArgs a) {
    constexpr int NW = WAVES_M * WAVES_N, BK = 64, ROWB = 128, CH = 8, RS = 8;
    constexpr int BM = WAVES_M * WM * 16, BN = WAVES_N * WN * 16;
    constexpr int NLA = BM / (RS * NW), NLB = BN / (RS * NW);
    constexpr int STAGE = (BM + BN) * ROWB;
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];   // 2 x STAGE
    unsigned char* ldsA = lds;
    unsigned char* ldsB = lds + BM * ROWB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave - wm * WAVES_N;
    const int tiles_n = a.N / BN, tiles_m = (a.M + BM - 1) / BM, ntiles = tiles_m * tiles_n;
    const int lrow = lane >> 3, slot = lane & 7;
    const int r16 = lane & 15, kg = lane >> 4, sw = (r16 >> 1) & 7;
    const unsigned char* rdA = ldsA + (wm * WM * 16 + r16) * ROWB;
    const unsigned char* rdB = ldsB + (wn * WN * 16 + r16) * ROWB;
    const int c0 = ((0 + kg) ^ sw) << 4, c1 = ((4 + kg) ^ sw) << 4;
    const int nk = a.K / BK;

    const __bf16* gA[NLA];
    const __bf16* gB[NLB];
    // virtual tile id -> (m0, n0): the XCD-aware grouped order of the non-persistent kernel over the whole tile list; a
    // block's ids are congruent mod 8 (the grid is a multiple of 8), so it stays on its XCD's range
    auto locate = [&](int vp, int& m0, int& n0) {
        const int q = ntiles >> 3, r = ntiles & 7, xcd = vp & 7, loc = vp >> 3;
        const int pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
        const int GM = a.group_m, per_group = GM * tiles_n;
        const int grp = pid / per_group, in_grp = pid - grp * per_group;
        const int rows_here = min(GM, tiles_m - grp * GM);
        const int tn = in_grp / rows_here, tm = grp * GM + (in_grp - tn * rows_here);
        m0 = tm * BM;
        n0 = tn * BN;
    };
    auto point = [&](int m0, int n0) {
#pragma unroll
        for (int i = 0; i < NLA; ++i) {
            const int row = (i * NW + wave) * RS + lrow;
            gA[i] = reinterpret_cast<const __bf16*>(a.A) + (size_t)min(m0 + row, a.M - 1) * a.K + (slot ^ ((row >> 1) & 7)) * 8;
        }
#pragma unroll
        for (int i = 0; i < NLB; ++i) {
            const int row = (i * NW + wave) * RS + lrow;
            gB[i] = reinterpret_cast<const __bf16*>(a.B) + (size_t)(n0 + row) * a.K + (slot ^ ((row >> 1) & 7)) * 8;
        }
    };
    auto issue = [&](int stage) {
#pragma unroll
        for (int i = 0; i < NLA; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gA[i],
                                             (__attribute__((address_space(3))) void*)(ldsA + stage * STAGE + (i * NW + wave) * 1024),
                                             16, 0, 0);
            gA[i] += BK;
        }
#pragma unroll
        for (int i = 0; i < NLB; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gB[i],
                                             (__attribute__((address_space(3))) void*)(ldsB + stage * STAGE + (i * NW + wave) * 1024),
                                             16, 0, 0);
            gB[i] += BK;
        }
    };

    int vp = blockIdx.x;
    if (vp >= ntiles) return;
    int m0, n0, st = 0;             // st: the LDS stage that holds (or will hold) the K tile about to be multiplied
    locate(vp, m0, n0);
    point(m0, n0);
    issue(st);
    while (true) {
        f32x4_t acc[WM][WN];
#pragma unroll
        for (int mi = 0; mi < WM; ++mi)
#pragma unroll
            for (int ni = 0; ni < WN; ++ni) acc[mi][ni] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < nk; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();      // K tile kt has landed in stage st; nobody reads stage st ^ 1 any more
            if (kt + 1 < nk) issue(st ^ 1);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int co = (kk ? c1 : c0) + st * STAGE;
                bf16x8_t fa[WM], fb[WN];
#pragma unroll
                for (int mi = 0; mi < WM; ++mi) fa[mi] = *reinterpret_cast<const bf16x8_t*>(rdA + mi * 16 * ROWB + co);
#pragma unroll
                for (int ni = 0; ni < WN; ++ni) fb[ni] = *reinterpret_cast<const bf16x8_t*>(rdB + ni * 16 * ROWB + co);
#pragma unroll
                for (int mi = 0; mi < WM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < WN; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[ni], fa[mi], acc[mi][ni], 0, 0, 0);
            }
            st ^= 1;
        }
        // here st names the stage that was read at step nk - 2: every wave has passed the barrier of step nk - 1 since,
        // so it is free for the first K tile of the next output tile
        const int cm0 = m0, cn0 = n0;
        vp += gridDim.x;
        const bool more = vp < ntiles;
        if (more) {
            locate(vp, m0, n0);
            point(m0, n0);
            issue(st);
        }
        gemm_epilogue<WM, WN>(a, acc, cm0 + wm * WM * 16, cn0 + wn * WN * 16, r16, kg);
        if (!more) break;
    }
}

__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ x, unsigned short* __restrict__ y, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) y[i] = f32_to_bf16_rne(x[i]);
}

template <int WAVES_M, int WAVES_N, int WM, int WN, int NSTAGE, int BK = 64>
int launch_cfg(const GemmArgs& a, hipStream_t stream) {
    constexpr int BM = WAVES_M * WM * 16, BN = WAVES_N * WN * 16;
    constexpr int lds = NSTAGE * (BM + BN) * BK * 2;
    auto* kern = &gemm_bf16_kernel<WAVES_M, WAVES_N, WM, WN, NSTAGE, BK>;
    static bool configured = false;
    if (!configured && lds > 64 * 1024) {
        PP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        configured = true;
    }
    if (a.M <= 0) return PP_OK;        // configure-only call (pp_gemm_bf16_prepare)
    const int tiles = ((a.M + BM - 1) / BM) * (a.N / BN);
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(64 * WAVES_M * WAVES_N), lds, stream, a);
    PP_HIP_CHECK(hipGetLastError());
    return PP_OK;
}

template <int WAVES_M, int WAVES_N>
int launch_pipelined(const GemmArgs& a, hipStream_t stream) {
    constexpr int BM = WAVES_M * 128, BN = WAVES_N * 64;
    constexpr int lds = 2 * (BM + BN) * 128;
    auto* kern = &gemm_bf16_pipelined_kernel<WAVES_M, WAVES_N>;
    static bool configured = false;
    if (!configured) {
        PP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        configured = true;
    }
    const int tiles = ((a.M + BM - 1) / BM) * (a.N / BN);
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(64 * WAVES_M * WAVES_N), lds, stream, a);
    PP_HIP_CHECK(hipGetLastError());
    return PP_OK;
}

template <int WAVES_M, int WAVES_N, int WM, int WN>
int launch_persistent(const GemmArgs& a, hipStream_t stream) {
    constexpr int BM = WAVES_M * WM * 16, BN = WAVES_N * WN * 16;
    constexpr int lds = 2 * (BM + BN) * 128;
    auto* kern = &gemm_bf16_persistent_kernel<WAVES_M, WAVES_N, WM, WN>;
    static int n_cu = 0;
    if (n_cu == 0) {
        PP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        int dev = 0;
        PP_HIP_CHECK(hipGetDevice(&dev));
        PP_HIP_CHECK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
        n_cu = std::max(8, n_cu / 8 * 8);        // a multiple of 8 keeps a block's tile ids on one XCD
    }
    const int tiles = ((a.M + BM - 1) / BM) * (a.N / BN);
    hipLaunchKernelGGL(kern, dim3(std::min(tiles, n_cu)), dim3(64 * WAVES_M * WAVES_N), lds, stream, a);
    PP_HIP_CHECK(hipGetLastError());
    return PP_OK;
}

}  // namespace

int pp_launch_f32_to_bf16(const float* x, void* y, size_t n, hipStream_t stream) {
    if (n == 0) return PP_OK;
    const unsigned grid = (unsigned)std::min<size_t>((n + 255) / 256, 65536);
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(grid), dim3(256), 0, stream, x, reinterpret_cast<unsigned short*>(y), n);
    PP_HIP_CHECK(hipGetLastError());
    return PP_OK;
}

// Tile configurations (POSEPIPE_GEMM_CFG forces one; tests run all of them):
//   0  128 x 128, 4 waves (64 x 64 each), 1 stage,  32 KiB LDS -> 4 blocks / CU
//   1  256 x 128, 8 waves (64 x 64 each), 1 stage,  48 KiB LDS -> 2 blocks / CU, 3/4 of the L2 -> LDS bytes per FLOP
//   2  256 x 256, 16 waves (64 x 64 each), 2 stages, 128 KiB LDS -> 1 block / CU, 1/2 of the bytes, in-block prefetch
//   3  256 x 128, 4 waves (128 x 64 each), 1 stage,  48 KiB LDS -> 2 blocks / CU, 3/4 of the LDS reads per FLOP
//   4  128 x 128, 4 waves, 2 stages, 64 KiB LDS
//   5  256 x 256, 8 waves (128 x 64 each), 2 stages, 128 KiB LDS -> 1 block / CU
//   6  256 x 128, 8 waves, 2 stages of K step 32, 48 KiB LDS -> 2 blocks / CU with the in-block prefetch
//   7  128 x 128, 4 waves, 2 stages of K step 32, 32 KiB LDS -> 4 blocks / CU
//   8  configuration 2 as a persistent kernel (one block per CU, next tile's first loads issued before the epilogue)
//   9  256 x 256, 8 waves (128 x 64 each), register-pipelined fragments, barrier mid K step: 821 / 671 / 747 / 1020
// set the dynamic-LDS attribute of the default large-tile kernel outside any hipGraph capture (called when an encoder /
// deconvolution object is created)
int pp_gemm_bf16_prepare() {
    GemmArgs none{};
    return launch_cfg<4, 4, 4, 4, 2>(none, nullptr);
}

int pp_launch_gemm_bf16(const GemmArgs& a_in, hipStream_t stream) {
    GemmArgs a = a_in;
    if (a.group_m <= 0) {
        const char* env_gm = getenv("POSEPIPE_GEMM_GROUP_M");
        a.group_m = env_gm ? std::max(1, atoi(env_gm)) : 4;   // measured best of {1, 2, 4, 8, 16, 48} on the ViT-H shapes
    }
    PP_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "gemm_bf16: empty problem");
    PP_REQUIRE(a.K % 64 == 0, "gemm_bf16: K = %d must be a multiple of 64", a.K);
    PP_REQUIRE(a.N % 128 == 0, "gemm_bf16: N = %d must be a multiple of 128", a.N);
    const char* env_cfg = getenv("POSEPIPE_GEMM_CFG");
    // default (measured on the ViT-H shapes at M = 12288, qkv / proj / fc1 / fc2 in TFLOP/s): 256 x 256 two-stage tiles
    // 867 / 743 / 730 / 997, 256 x 128 tiles 831 / 721 / 720 / 931, 128 x 128 tiles 785 / 673 / 716 / 844; small problems
    // keep the smaller tiles for the larger grid
    const long rows256 = (a.M + 255) / 256;
    int cfg = env_cfg ? atoi(env_cfg)
                      : (a.N % 256 == 0 && rows256 * (a.N / 256) >= 128) ? 2 : (rows256 * (a.N / 128) >= 256 ? 1 : 0);
    if ((cfg == 2 || cfg == 5 || cfg == 8 || cfg == 9) && a.N % 256 != 0) cfg = 1;
    switch (cfg) {
        case 1: return launch_cfg<4, 2, 4, 4, 1>(a, stream);
        case 2: return launch_cfg<4, 4, 4, 4, 2>(a, stream);
        case 3: return launch_cfg<2, 2, 8, 4, 1>(a, stream);
        case 4: return launch_cfg<2, 2, 4, 4, 2>(a, stream);
        case 5: return launch_cfg<2, 4, 8, 4, 2>(a, stream);
        case 6: return launch_cfg<4, 2, 4, 4, 2, 32>(a, stream);
        case 7: return launch_cfg<2, 2, 4, 4, 2, 32>(a, stream);
        case 8: return launch_persistent<4, 4, 4, 4>(a, stream);
        case 9: return launch_pipelined<2, 4>(a, stream);
        default: return launch_cfg<2, 2, 4, 4, 1>(a, stream);
    }
}
