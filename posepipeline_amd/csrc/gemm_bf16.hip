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
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

namespace {

typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;

// round-to-nearest-even float32 -> bf16 (v_cvt_pk_bf16_f32; NaN stays NaN)
__device__ __forceinline__ unsigned short f32_to_bf16_rne(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
__device__ __forceinline__ ushort4 f32x4_to_bf16(f32x4_t v) { return __builtin_bit_cast(ushort4, __builtin_convertvector(v, bf16x4_t)); }

// GELU(v) = 0.5 v (1 + erf(v / sqrt 2)); erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below the bf16 rounding
// of the result): 1 rcp + 1 exp instead of libm's branchy erff, evaluated on PAIRS (v_pk_fma_f32 / v_pk_mul_f32: two float32
// results per instruction) -- the epilogue of the fc1 GEMM applies it to 128 values per lane.
__device__ __forceinline__ f32x2_t gelu_erf2(f32x2_t v) {
    const f32x2_t x = v * 0.70710678118654752440f;
    const f32x2_t ax = __builtin_elementwise_abs(x);
    const f32x2_t den = __builtin_elementwise_fma(f32x2_t{0.3275911f, 0.3275911f}, ax, f32x2_t{1.f, 1.f});
    const f32x2_t t{__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
    f32x2_t p = __builtin_elementwise_fma(f32x2_t{1.061405429f, 1.061405429f}, t, f32x2_t{-1.453152027f, -1.453152027f});
    p = __builtin_elementwise_fma(p, t, f32x2_t{1.421413741f, 1.421413741f});
    p = __builtin_elementwise_fma(p, t, f32x2_t{-0.284496736f, -0.284496736f});
    p = __builtin_elementwise_fma(p, t, f32x2_t{0.254829592f, 0.254829592f});
    const f32x2_t arg = (ax * ax) * -1.4426950408889634f;                 // e^(-x^2) = 2^(-x^2 log2 e)
    const f32x2_t e{__builtin_amdgcn_exp2f(arg[0]), __builtin_amdgcn_exp2f(arg[1])};
    const f32x2_t erf_ax = __builtin_elementwise_fma(-(p * t), e, f32x2_t{1.f, 1.f});
    const f32x2_t sg{copysignf(erf_ax[0], x[0]), copysignf(erf_ax[1], x[1])};
    return (v * 0.5f) * (sg + 1.0f);
}
__device__ __forceinline__ f32x4_t gelu_erf4(f32x4_t v) {
    const f32x2_t a = gelu_erf2(f32x2_t{v[0], v[1]}), b = gelu_erf2(f32x2_t{v[2], v[3]});
    return f32x4_t{a[0], a[1], b[0], b[1]};
}

// bias -> GELU -> residual -> store of 4 consecutive n of one row m
__device__ __forceinline__ void gemm_store4(const GemmArgs& a, f32x4_t v, int m, int rrow, int n) {
    if (a.bias) {
        const float4 b = *reinterpret_cast<const float4*>(a.bias + n);
        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
    }
    if (a.act == 1) {
        v = gelu_erf4(v);
    }
    if (a.res) {
        const float4 rr = *reinterpret_cast<const float4*>(a.res + (size_t)rrow * a.N + n);
        v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
    }
    if (a.out_bf16) {
        const ushort4 o = f32x4_to_bf16(v);
        size_t off = (size_t)m * a.N + n;
        if (a.qkv_tokens > 0) {      // head-major q / k / v: [which][sample][head][token][d]; 4 | head_dim keeps the 4 values together
            const int dim = a.N / 3, which = n / dim, rem = n - which * dim, head = rem / a.qkv_hd, d = rem - head * a.qkv_hd;
            const int smp = m / a.qkv_tokens, tok = m - smp * a.qkv_tokens;
            off = ((((size_t)which * (a.M / a.qkv_tokens) + smp) * (dim / a.qkv_hd) + head) * a.qkv_tokens + tok) * a.qkv_hd + d;
        }
        *reinterpret_cast<ushort4*>(reinterpret_cast<unsigned short*>(a.C) + off) = o;
    } else {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.C) + (size_t)m * a.N + n) = make_float4(v[0], v[1], v[2], v[3]);
    }
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
        for (int ni = 0; ni < WN; ++ni) gemm_store4(a, acc[mi][ni], m, rrow, nbase + ni * 16 + kg * 4);
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


// Ping-pong form (configuration 10): 256 x 256 tile, K step 64, 8 waves of 128 (m) x 64 (n), v_mfma_f32_32x32x16_bf16, two
// 64 KiB LDS buffers, one workgroup per CU.  The waves form two groups (waves 0-3 / 4-7:
// the m halves of the tile, one wave of each group per SIMD) that run the SAME instruction stream ONE PHASE APART.  A K tile is
// 8 phases per wave, alternately a
//   load segment    ds_read_b128 of the fragments of the next product, LDS-DMA requests for a later K tile, s_waitcnt
//   matrix segment  8 MFMAs (256 matrix-pipe cycles): one quadrant (64 m x 32 n) of the wave tile over the whole K step
// and every segment ends in ONE s_barrier of all 8 waves, so while one group's wave multiplies, the other wave of the SIMD
// loads: the matrix pipe of a SIMD sees 8 x 256 cycles of MFMAs per K tile back to back when a load segment fits in 256 cycles.
// Quadrant order (A0,Be) (A0,Bo) (A1,Bo) (A1,Be) with e = parity of the K tile, so that one operand stays in registers between
// consecutive products: 8 + 4 + 8 + 4 fragment reads per K tile (the last 4 already belong to the NEXT tile).
//
// Hand-counted synchronisation (g0 = phases 8 kt + 0 .. 7 for L1 C1 L2 C2 L3 C3 L4 C4 of K tile kt, g1 one phase later):
//   * buffer e = kt & 1 holds K tile kt.  It is read in L4(kt - 1) [Be], L1 [A0], L2 [Bo], L3 [A1]; the last reader is g1's
//     L3(kt) in phase 8 kt + 5, and every load segment ends in s_waitcnt lgkmcnt(0) BEFORE its barrier.
//   * K tile kt + 2 is requested into buffer e in L4(kt) (phases 8 kt + 6 / 7: after that barrier), L1(kt + 1) and L2(kt + 1);
//     the requests are raw global_load_lds_dwordx4 (the compiler neither sees nor waits for them).
//   * every wave waits s_waitcnt vmcnt(0) at the end of L3(kt + 1) (phases 8 kt + 12 / 13) = its own requests for K tile kt + 2
//     have landed (nothing younger is in flight: the next request is issued in L4(kt + 1)); the first read of K tile kt + 2 is
//     g0's L4(kt + 1) in phase 8 kt + 14, two barriers after g0's wait and one after g1's.
//   * barrier balance per output tile: g1 starts with one extra barrier and omits the one after its last matrix segment.
//   * one output tile per workgroup.  (A persistent tile loop that requests the next tile's first K tiles before its epilogue was
//     measured SLOWER, 346 vs 275 us on the qkv shape: vmcnt counts a wave's stores with its loads, so the next tile's first
//     "landed" wait also waits for the epilogue's stores; a fresh workgroup starts with fresh counters while they drain.)
// Epilogue: the accumulators start at bias[n] (loaded while the first K tiles fly), so what is left is GELU, the rounding and
// the stores.  A lane holds C[m][n .. n + 3] with m = block row (lane & 31): stored from there, one instruction would touch 32
// rows with 32 (fp32) or 16 (bf16) bytes each.  The operand buffers are free once a wave has left the K loop (every fragment read
// of the tile happened before the last barrier it passed), so each wave transposes its 128 x 64 tile through a private 16 KiB
// of them ([128][128 B], 16-byte chunk ^ (row & 7); fp32: one 32-column half at a time) and writes whole 128-byte lines, 8 rows
// per instruction; the fp32 residual is read in the same lines.  No integer division per row (the head-major q / k / v offset
// advances incrementally).
// POSEPIPE_GEMM_CFG 10 / 11 / 12: the 8 DMA requests of a wave spread 4-2-2 / 4-4-0 / 2-3-3 over L4 / L1 / L2.
template <int D4, int D1, int D2>
__global__ __launch_bounds__(512, 1) void gemm_bf16_pingpong_kernel(GemmArgs a) {
    static_assert(D4 + D1 + D2 == 8, "8 LDS-DMA requests per wave and K tile");
    constexpr int BM = 256, BN = 256, BK = 64, ROWB = 128;
    constexpr int TILE = BM * ROWB;          // one operand tile: 32 KiB
    constexpr int BUF = 2 * TILE;            // A tile, then B tile
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];   // 2 x BUF
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, q = wave & 3;
    const int tiles_n = a.N / BN, tiles_m = (a.M + BM - 1) / BM, ntiles = tiles_m * tiles_n;

    // virtual tile id -> (m0, n0): XCD-contiguous ranges of the grouped order (group_m tile rows sweep the tile columns); a
    // workgroup's ids are congruent mod 8 (the grid is a multiple of 8), so it stays on its XCD's range
    auto locate = [&](int vp, int& m0, int& n0) {
        const int qq = ntiles >> 3, r = ntiles & 7, xcd = vp & 7, loc = vp >> 3;
        const int pid = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + loc;
        const int GM = a.group_m, per_group = GM * tiles_n;
        const int gi = pid / per_group, in_grp = pid - gi * per_group;
        const int rows_here = min(GM, tiles_m - gi * GM);
        const int tn = in_grp / rows_here, tm = gi * GM + (in_grp - tn * rows_here);
        m0 = tm * BM;
        n0 = tn * BN;
    };

    // LDS-DMA: request r (0 .. 7) of wave w fills 1 KiB slab 4 w + (r & 3) of the A tile (r < 4) or of the B tile: 8 tile rows
    // of 128 bytes; lane -> (row, 16-byte position), source chunk = position ^ ((row >> 1) & 7)  (the read side undoes it).
    // The slab's first row goes into the scalar base, so a lane keeps two offsets (slab parity decides ((row >> 1) & 7) bit 2);
    // the last tile row of a ragged M clamps its rows per lane instead (uniform branch).
    unsigned goff[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        const int row = par * 8 + (lane >> 3);
        goff[par] = (unsigned)(lane >> 3) * (unsigned)a.K * 2u + (unsigned)((lane & 7) ^ ((row >> 1) & 7)) * 16u;
    }
    const unsigned char* gbaseA;
    const unsigned char* gbaseB;
    int rows_left;                           // rows of A below m0 (>= 256: no clamp)
    auto point = [&](int m0, int n0) {
        gbaseA = reinterpret_cast<const unsigned char*>(a.A) + ((size_t)m0 * a.K + (size_t)wave * 32 * a.K) * 2;
        gbaseB = reinterpret_cast<const unsigned char*>(a.B) + ((size_t)n0 * a.K + (size_t)wave * 32 * a.K) * 2;
        rows_left = a.M - m0;
    };
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const unsigned slab0 = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)wave * 4096u);
    // requests [r0, r1) of K tile kt into buffer buf
    auto dma = [&](int kt, int buf, int r0, int r1) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (r < r0 || r >= r1) continue;
            const unsigned char* src = (r < 4 ? gbaseA : gbaseB) + (size_t)kt * (BK * 2) + (size_t)(r & 3) * 8 * a.K * 2;
            const unsigned dst = slab0 + (unsigned)(buf * BUF + (r < 4 ? 0 : TILE) + (r & 3) * 1024);
            unsigned off = goff[r & 1];
            if (r < 4 && rows_left < BM) {       // ragged last tile row: rows >= M read row M - 1 (their products are never stored)
                const int row = (wave * 4 + r) * 8 + (lane >> 3);
                const unsigned chunk16 = goff[r & 1] - (unsigned)(lane >> 3) * (unsigned)a.K * 2u;
                off = (unsigned)min(row, rows_left - 1) * (unsigned)a.K * 2u + chunk16;
                src = gbaseA - (size_t)wave * 32 * a.K * 2 + (size_t)kt * (BK * 2);
            }
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(src), "s"(dst) : "memory", "m0");
        }
    };

    // fragment reads: lane -> row (lane & 31) of a 32-row block, k half (lane >> 5) of a 16-wide k step
    const int rl = lane & 31, hk = lane >> 5, sw = (rl >> 1) & 7;
    const unsigned char* rdA = lds + (grp * 128 + rl) * ROWB;
    const unsigned char* rdB = lds + TILE + (q * 64 + rl) * ROWB;
    int kofs[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kofs[ks] = ((2 * ks + hk) ^ sw) << 4;

    bf16x8_t fa[2][4], fb[2][4];             // fa[j]: A block (2 half + j) of the current half; fb[s]: B block s
    auto readA = [&](int buf, int half) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                fa[j][ks] = *reinterpret_cast<const bf16x8_t*>(rdA + buf * BUF + (half * 2 + j) * 32 * ROWB + kofs[ks]);
    };
    auto readB = [&](int buf, int s) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fb[s][ks] = *reinterpret_cast<const bf16x8_t*>(rdB + buf * BUF + s * 32 * ROWB + kofs[ks]);
    };
    f32x16_t acc[2][4];                      // [B block][A block]
    auto mma = [&](int s, int half) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[s][half * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[s][ks], fa[j][ks], acc[s][half * 2 + j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto end_load = [&](bool landed) {       // end of a load segment
        if (landed) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto end_mma = [&]() {
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    const int nk = a.K / BK;                 // even (launcher)

    int m0, n0;
    locate(blockIdx.x, m0, n0);
    point(m0, n0);
    dma(0, 0, 0, 8);
    if (nk > 1) dma(1, 1, 0, D4);
    {
        // accumulators start at the bias (the loads overlap the flight of the first K tiles)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
                if (a.bias) b = *reinterpret_cast<const float4*>(a.bias + n0 + q * 64 + s * 32 + 8 * g4 + 4 * hk);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[s][j][4 * g4 + 0] = b.x; acc[s][j][4 * g4 + 1] = b.y; acc[s][j][4 * g4 + 2] = b.z; acc[s][j][4 * g4 + 3] = b.w;
                }
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // K tile 0 (and the first requests of K tile 1) landed
        __builtin_amdgcn_s_barrier();
        readB(0, 0);                             // "L4(-1)"
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (grp == 1) __builtin_amdgcn_s_barrier();

        // one K tile of parity E (compile time): buffer E, first B block E
#define PP_TILE(E, kt)                                                                                   \
    {                                                                                                    \
        readA(E, 0);                                                                                     \
        if ((kt) + 1 < nk) dma((kt) + 1, 1 - (E), D4, D4 + D1);                                           \
        end_load(false);                                                                                 \
        mma(E, 0);                                                                                       \
        end_mma();                                                                                       \
        readB(E, 1 - (E));                                                                               \
        if ((kt) + 1 < nk) dma((kt) + 1, 1 - (E), D4 + D1, 8);                                            \
        end_load(false);                                                                                 \
        mma(1 - (E), 0);                                                                                 \
        end_mma();                                                                                       \
        readA(E, 1);                                                                                     \
        end_load(true);                                                                                  \
        mma(1 - (E), 1);                                                                                 \
        end_mma();                                                                                       \
        if ((kt) + 1 < nk) readB(1 - (E), 1 - (E));                                                      \
        if ((kt) + 2 < nk) dma((kt) + 2, E, 0, D4);                                                      \
        end_load(false);                                                                                 \
        mma(E, 1);                                                                                       \
        if (!(grp == 1 && (kt) + 1 >= nk)) end_mma();                                                    \
    }
        for (int kt = 0; kt < nk; kt += 2) {
            PP_TILE(0, kt)
            PP_TILE(1, kt + 1)
        }
#undef PP_TILE

        const int rl_e = rl, hk_e = hk;
        const int rdrow = lane >> 3, rdpos = lane & 7, rdchunk = rdpos ^ rdrow;
        unsigned char* stg = lds + wave * 16384;
        const int mw = m0 + grp * 128, nw = n0 + q * 64;
        if (!a.out_bf16) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        f32x4_t v{acc[s][j][4 * g4], acc[s][j][4 * g4 + 1], acc[s][j][4 * g4 + 2], acc[s][j][4 * g4 + 3]};
                        if (a.act == 1) v = gelu_erf4(v);
                        *reinterpret_cast<f32x4_t*>(stg + (j * 32 + rl_e) * 128 + (((2 * g4 + hk_e) ^ (rl_e & 7)) << 4)) = v;
                    }
#pragma unroll
                for (int it = 0; it < 16; ++it) {
                    const int ml = it * 8 + rdrow, m = mw + ml, n = nw + s * 32 + rdchunk * 4;
                    f32x4_t v = *reinterpret_cast<const f32x4_t*>(stg + ml * 128 + rdpos * 16);
                    if (m >= a.M) continue;
                    if (a.res) {
                        const int rrow = a.res_mod > 0 ? m % a.res_mod : m;
                        const float4 rr = *reinterpret_cast<const float4*>(a.res + (size_t)rrow * a.N + n);
                        v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
                    }
                    *reinterpret_cast<f32x4_t*>(reinterpret_cast<float*>(a.C) + (size_t)m * a.N + n) = v;
                }
            }
        } else {
            // destination of the lane's 8 values of row m: m N + n, or head-major q / k / v
            // [which][sample][head][token][d] (8 | head_dim: launcher); rows advance by 8 per instruction: (sample, token) incrementally
            const int n = nw + rdchunk * 8;
            size_t col = (size_t)n, row_stride = (size_t)a.N;
            int smp = 0, tok = 0;
            size_t smp_stride = 0;
            if (a.qkv_tokens > 0) {
                const int dim = a.N / 3, which = n / dim, rem = n - which * dim, head = rem / a.qkv_hd, d = rem - head * a.qkv_hd;
                const int heads = dim / a.qkv_hd;
                smp = (mw + rdrow) / a.qkv_tokens;
                tok = (mw + rdrow) - smp * a.qkv_tokens;
                smp_stride = (size_t)heads * a.qkv_tokens * a.qkv_hd;
                col = ((size_t)which * (a.M / a.qkv_tokens) * heads + head) * a.qkv_tokens * a.qkv_hd + d;
                row_stride = (size_t)a.qkv_hd;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        f32x4_t v{acc[s][j][4 * g4], acc[s][j][4 * g4 + 1], acc[s][j][4 * g4 + 2], acc[s][j][4 * g4 + 3]};
                        if (a.act == 1) v = gelu_erf4(v);
                        if (a.res) {     // (not a ViT case) the residual is added before the rounding to bf16
                            const int m = min(mw + j * 32 + rl_e, a.M - 1), rrow = a.res_mod > 0 ? m % a.res_mod : m;
                            const float4 rr = *reinterpret_cast<const float4*>(a.res + (size_t)rrow * a.N + nw + s * 32 + 8 * g4 + 4 * hk_e);
                            v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
                        }
                        *reinterpret_cast<ushort4*>(stg + (j * 32 + rl_e) * 128 + (((s * 4 + g4) ^ (rl_e & 7)) << 4) + hk_e * 8) = f32x4_to_bf16(v);
                    }
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int ml = it * 8 + rdrow, m = mw + ml;
                const uint4 o = *reinterpret_cast<const uint4*>(stg + ml * 128 + rdpos * 16);
                size_t off;
                if (a.qkv_tokens > 0) {
                    off = col + (size_t)smp * smp_stride + (size_t)tok * row_stride;
                    tok += 8;
                    if (tok >= a.qkv_tokens) { tok -= a.qkv_tokens; ++smp; }
                } else {
                    off = (size_t)m * row_stride + col;
                }
                if (m < a.M) *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(a.C) + off) = o;
            }
        }
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
    static PpPerDeviceOnce configured;
    if (lds > 64 * 1024)
        configured.run([&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds); });
    if (a.M <= 0) return PP_OK;        // configure-only call (pp_gemm_bf16_prepare)
    const int tiles = ((a.M + BM - 1) / BM) * (a.N / BN);
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(64 * WAVES_M * WAVES_N), lds, stream, a);
    PP_HIP_CHECK(hipGetLastError());
    return PP_OK;
}

template <int D4, int D1, int D2>
int launch_pingpong(const GemmArgs& a, hipStream_t stream) {
    constexpr int lds = 128 * 1024;          // two 64 KiB operand buffers (the epilogue stages through them)
    auto* kern = &gemm_bf16_pingpong_kernel<D4, D1, D2>;
    static PpPerDeviceOnce configured;
    configured.run([&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds); });
    if (a.M <= 0) return PP_OK;
    const int tiles = ((a.M + 255) / 256) * (a.N / 256);
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(512), lds, stream, a);
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
//  10  ping-pong form (gemm_bf16_pingpong_kernel): THE form of every problem with N % 256 == 0 and K % 128 == 0 -- i.e. of every
//      GEMM of the ViT-B / L / H encoders and of the deconvolution head, at every batch size, so that results do not depend on
//      which batch size selected which kernel (0 .. 2 are bit-identical to each other; 10 sums K in steps of 16 with the bias
//      first).  0 .. 2 remain for the shapes 10 does not take.
// Round 4 removed the forms that had lost their A/Bs and were reachable only through the knob (measured in rounds 1 - 3, numbers
// in DESIGN.md 5b): 256 x 128 with 128 x 64 wave tiles, two-stage 128 x 128, 256 x 256 with 8 fat waves, K step 32, a persistent
// tile loop, a register-pipelined form, and the 4-4-0 / 2-3-3 DMA placements of the ping-pong form.
// set the dynamic-LDS attribute of the default large-tile kernel outside any hipGraph capture (called when an encoder /
// deconvolution object is created)
int pp_gemm_bf16_prepare() {
    GemmArgs none{};
    const int r = launch_pingpong<4, 2, 2>(none, nullptr);
    return r != PP_OK ? r : launch_cfg<4, 4, 4, 4, 2>(none, nullptr);
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
    // measured on the ViT-H shapes (qkv / proj / fc1 / fc2, TFLOP/s): ping-pong form at M = 24576 952 / 806 / 855 / 913 against 840 /
    // 750 / 774 / 877 for (2) on the same box; inside the encoder the GEMMs of a step take 32.8 ms against 33.7.  Shapes it does not
    // take: 256 x 256 two-stage tiles where they fill the chip, else the smaller tiles for the larger grid.
    const long rows256 = (a.M + 255) / 256;
    const bool pingpong_ok = a.N % 256 == 0 && a.K % 128 == 0 && !(a.qkv_tokens > 0 && (a.qkv_hd % 8 != 0 || a.qkv_tokens % 8 != 0));
    int cfg = env_cfg ? atoi(env_cfg)
                      : pingpong_ok ? 10
                      : (a.N % 256 == 0 && rows256 * (a.N / 256) >= 128) ? 2 : (rows256 * (a.N / 128) >= 256 ? 1 : 0);
    if ((cfg == 2 || cfg == 10) && a.N % 256 != 0) cfg = 1;
    if (cfg == 10 && !pingpong_ok) cfg = 2;
    switch (cfg) {
        case 1: return launch_cfg<4, 2, 4, 4, 1>(a, stream);
        case 2: return launch_cfg<4, 4, 4, 4, 2>(a, stream);
        case 10: return launch_pingpong<4, 2, 2>(a, stream);
        default: return launch_cfg<2, 2, 4, 4, 1>(a, stream);
    }
}
