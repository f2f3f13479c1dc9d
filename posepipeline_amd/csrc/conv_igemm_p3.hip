// Software-pipelined variant of the implicit-GEMM convolution (same arithmetic, same results as conv_igemm.hip).
//
// Why: tools/conv_probe.py on MI355X (profiles/r02_conv_probe.txt) shows that the two-barrier K step of conv_igemm_kernel
// leaves a bubble of ~1300 cycles per 32-k chunk on every SIMD: the four resident workgroups of a CU start together and the
// SIMD's round-robin MFMA issue keeps their waves in lock-step, so all of them write LDS, meet the barrier and read
// their operands at the SAME time while the matrix pipe idles (72 chunks x 1300 cycles = the 14 % the 256->256 3x3
// layers lose; short-K layers lose the same per chunk plus their prologue / epilogue).
//
// This kernel takes LDS out of the critical path with a three-stage pipeline over 16-k steps:
//   stage A  global -> registers      (issued three steps ahead)
//   stage B  registers -> LDS         (written two steps ahead, into one of THREE 12 KB buffers)
//   stage C  LDS -> operand registers (ds_read_b128 issued ONE step ahead, double-buffered in VGPRs)
// so that in step s a wave issues the MFMAs of step s from registers that are already loaded, with the LDS writes of
// step s+2, the operand reads of step s+1 and the global loads of step s+3 interleaved between them, and ONE barrier
// per step.  After the barrier the next MFMA can issue immediately.  LDS per block 36 KB (4 blocks per CU as before).
//
// Numerics: unchanged.  Every output is the same k-ordered fmaf chain (k = (kh, kw, cin) ascending) -- the MFMA steps of
// one accumulator are issued in the same order, only the staging differs -- so results are bit-identical to
// conv_igemm_kernel and to oracle/conv_ref.c (tests/test_gpu_conv.py runs both kernels).
//
// LDS layout of a stage: rows of 16 floats (64 B).  Row = output channel (weights) or pixel; within a row the 16-byte slot
// g holds the operands k = 4s + g, s = 0..3, of MFMA lane group g (lane >> 4), so one ds_read_b128 feeds the four MFMA steps
// of a 16-k step.  Slots are XOR-swizzled with (row >> 1) & 3: the transposing ds_write_b32 of the pixel loader then
// hits every bank exactly twice per 64 lanes (the minimum), the reads stay 1 KB-contiguous per wave.
#include <type_traits>

#include "pp_internal.h"
#include "pp_amax.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short i16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int BK = 16;

__device__ __forceinline__ unsigned udiv(unsigned n, unsigned m, unsigned s1, unsigned s2) {
    const unsigned t = __umulhi(m, n);
    return (t + ((n - t) >> s1)) >> s2;
}

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

// WALK: Cin >= 16, so a 16-k step spans at most two kernel taps and the tap bookkeeping lives on the scalar unit.
// FAKE: timing experiment only (POSEPIPE_CONV_VARIANT=2): the tap walk is replaced by one add per load, so the results are
// WRONG; the difference in speed is what the address arithmetic of the K loop costs.
// TAB: the per-step tap bookkeeping comes from a table the launcher built once per layer geometry (ConvArgs::tap_table:
// [step][k-quad] -> (packed (dh, dw) of the quad's tap or the never-in-range marker, byte offset of the quad from the pixel's
// (hi0, wi0, channel 0))): one 8-byte load per lane and step, fetched one step ahead, replaces ~40 scalar and ~12 vector
// instructions of the walk -- measured with the FAKE variant, that arithmetic was 6 % (K = 2304) to 14 % (K = 432 / 864) of
// the kernel's time (profiles/r02_conv_probe.txt).
// NOCHK (with TAB): convolutions without padding never leave the image, so the per-quad bounds test is dropped: the byte
// offset is pixel base + table entry, one add per 16-byte load.  The only out-of-range cases left are rows past M and k past
// K, and both are made out of range for the BUFFER descriptor instead (base / entry = 0x80000000, tensors < 2 GiB), where the
// hardware returns zeros.  Covers every 1x1 layer, the RoI head's 7x7 'valid' fc6 / fc7 and VideoPose3D.
template <int CT, int PT, bool WALK, bool FAKE, bool TAB, bool NOCHK>
__device__ __forceinline__ void conv_p3_body(const ConvArgs& a) {
    constexpr int BC = 16 * CT;          // output channels per block
    constexpr int BP = 64 * PT;          // pixels per block (4 waves x PT x 16)
    constexpr int NQ = PT;               // pixel quads per thread and step (BP rows x 4 quads / 256 threads)
    constexpr int WROWS = 64;            // weight rows staged per step (256 threads = 64 rows x 4 quads; rows >= BC unused)
    constexpr int STAGE = (WROWS + BP) * BK;
    __shared__ __attribute__((aligned(16))) float smem[3 * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    int tile_x = blockIdx.x, tile_y = blockIdx.y;
    if (a.xcd_remap) {      // XCD-aware tile order, as in conv_igemm_kernel
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

    // ---- loader role: k-quad kq of pixel rows prow0 + 64 i and of weight row prow0 ------------------------------
    const int kq = tid & 3;
    const int prow0 = tid >> 2;
    const int fsw = (prow0 >> 1) & 3;    // rows prow0 + 64 i share it
    unsigned pbase[NQ];
    int phw[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const int m = m0 + prow0 + 64 * i;
        const bool mok = m < a.M;
        const unsigned mm = mok ? (unsigned)m : 0u;
        const int n = (int)udiv(mm, a.div_hw_m, a.div_hw_s1, a.div_hw_s2);
        const int rem = (int)mm - n * a.HWout;
        const int ho = (int)udiv((unsigned)rem, a.div_w_m, a.div_w_s1, a.div_w_s2);
        const int wo = rem - ho * a.Wout;
        const int hi0 = ho * a.stride - a.pad_h, wi0 = wo * a.stride - a.pad_w;
        pbase[i] = (WALK || TAB) ? (unsigned)(((n * (a.Hin + a.x_pad) + hi0) * (a.Win + a.x_pad) + wi0) * a.Cin) * 4u
                        : (unsigned)n * (unsigned)(a.Hin * a.Win * a.Cin);
        phw[i] = mok ? ((hi0 << 16) | (wi0 & 0xffff)) : (int)0x80000000;
        if (NOCHK && !mok) pbase[i] = 0x80000000u;
    }
    // weights: blob rows are [32] floats per 32-k chunk in operand order (element 8 g + s <-> k = 4 s + g); the 16-k step
    // h of a chunk is the float4 at 8 g + 4 h of each row = exactly lane group g's four operands of that step
    const unsigned wofs = (unsigned)(((c0 + prow0 < a.CoutPad) ? prow0 : 0) * 32 + kq * 8);
    const float* wblob = a.w + (size_t)c0 * 32;

    float4 xr[NQ];
    float4 wr;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)a.x_bytes, 0x00020000);
    int c0s = 0, kh0 = 0, kw0 = 0;       // WALK state (wave-uniform): the step starts c0s channels into tap (kh0, kw0)
    const int kq16 = 16 * kq;
    const unsigned lim = ((unsigned)(a.Hin - 1) << 16) | (unsigned)(a.Win - 1);

    // TAB: this lane's table entry of the step the next load_step call serves
    const uint2* tap = TAB ? a.tap_table + kq : nullptr;
    uint2 te = TAB ? tap[0] : make_uint2(0u, 0u);

    auto load_step = [&](int k0) {
        const bool kok = k0 + 4 * kq < a.K;
        if constexpr (TAB && NOCHK) {
#pragma unroll
            for (int i = 0; i < NQ; ++i)
                xr[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)(pbase[i] + te.y), 0, 0));
            tap += 4;
            te = tap[0];
        } else if constexpr (TAB) {
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                const u16x2 hw = __builtin_bit_cast(u16x2, __builtin_bit_cast(i16x2, phw[i]) + __builtin_bit_cast(i16x2, te.x));
                const bool ok = __builtin_bit_cast(unsigned, __builtin_elementwise_max(hw, __builtin_bit_cast(u16x2, lim))) == lim;
                const unsigned off = ok ? pbase[i] + te.y : 0xffffffffu;
                xr[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)off, 0, 0));
            }
            tap += 4;
            te = tap[0];             // the table has one spare step past the end: no bounds test
        } else if constexpr (FAKE) {
#pragma unroll
            for (int i = 0; i < NQ; ++i)
                xr[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)(pbase[i] + (unsigned)k0 * 4u + kq16), 0, 0));
        } else if constexpr (WALK) {
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
            const unsigned dpk = kok ? (first ? p0 : p1) : 0x7fff7fffu;
            const unsigned delta = (first ? d0 : d1) + (unsigned)kq16;
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
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
                const unsigned off = ok ? (pbase[i] + (unsigned)((hi * a.Win + wi) * a.Cin + qc)) * 4u : 0xffffffffu;
                xr[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)off, 0, 0));
            }
        }
        const float* wsrc = wblob + (size_t)(k0 >> 5) * a.CoutPad * 32 + ((k0 >> 2) & 4);
        wr = *reinterpret_cast<const float4*>(wsrc + wofs);
    };

    auto store_step = [&](int buf) {
        float* Ws = smem + buf * STAGE;
        float* Xs = Ws + WROWS * BK;
        // transpose: element r of the quad is k = 4 kq + r -> lane group g = r, MFMA step s = kq -> slot g, element s
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            float* row = Xs + (prow0 + 64 * i) * BK + kq;
            row[(0 ^ fsw) * 4] = xr[i].x;
            row[(1 ^ fsw) * 4] = xr[i].y;
            row[(2 ^ fsw) * 4] = xr[i].z;
            row[(3 ^ fsw) * 4] = xr[i].w;
        }
        *reinterpret_cast<float4*>(Ws + prow0 * BK + ((kq ^ fsw) * 4)) = wr;
    };

    f32x4 acc[CT][PT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) acc[ct][pt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int lrow = lane >> 4;   // g: k offset within an MFMA step
    const int lcol = lane & 15;   // cout (A) / pixel (B) within the 16-tile
    const int rd_w = lcol * BK + ((lrow ^ ((lcol >> 1) & 3)) * 4);
    const int rd_x = WROWS * BK + (wave * (16 * PT) + lcol) * BK + ((lrow ^ ((lcol >> 1) & 3)) * 4);

    auto read_operands = [&](int buf, f32x4 (&av)[CT], f32x4 (&bv)[PT]) {
        const float* base = smem + buf * STAGE;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) av[ct] = *reinterpret_cast<const f32x4*>(base + rd_w + ct * 16 * BK);
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) bv[pt] = *reinterpret_cast<const f32x4*>(base + rd_x + pt * 16 * BK);
    };
    auto mma = [&](const f32x4 (&av)[CT], const f32x4 (&bv)[PT], int s) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int pt = 0; pt < PT; ++pt)
                acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ct][s], bv[pt][s], acc[ct][pt], 0, 0, 0);
    };

    // number of 16-k steps: the zero-padded second half of the last 32-k chunk is skipped when it holds no taps
    const int nsteps = (a.K + BK - 1) / BK;

    f32x4 avA[CT], bvA[PT], avB[CT], bvB[PT];
    // ---- prologue: steps 0 and 1 into LDS, operands of step 0 into registers, step 2 in flight ------------------
    load_step(0);
    store_step(0);
    if (nsteps > 1) load_step(BK);
    __syncthreads();
    read_operands(0, avA, bvA);
    if (nsteps > 1) store_step(1);
    if (nsteps > 2) load_step(2 * BK);
    __syncthreads();

    // One pipeline step: MFMAs of step s from (av, bv); meanwhile stage B of step s + 2, stage C of step s + 1 into
    // (avn, bvn), stage A of step s + 3.  The groups are fenced (sched_barrier) so that every MFMA sub-step carries one
    // kind of memory work in its shadow: LDS writes, then operand reads, then the address arithmetic + global loads.
    auto step = [&](auto full, int s, int b1, int b2, const f32x4 (&av)[CT], const f32x4 (&bv)[PT], f32x4 (&avn)[CT],
                    f32x4 (&bvn)[PT]) {
        constexpr bool FULL = decltype(full)::value;       // steady state: every stage has work, no branches in the body
        mma(av, bv, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (FULL || s + 2 < nsteps) store_step(b2);
        __builtin_amdgcn_sched_barrier(0);
        mma(av, bv, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (FULL || s + 1 < nsteps) read_operands(b1, avn, bvn);
        __builtin_amdgcn_sched_barrier(0);
        mma(av, bv, 2);
        __builtin_amdgcn_sched_barrier(0);
        if (FULL || s + 3 < nsteps) load_step((s + 3) * BK);
        __builtin_amdgcn_sched_barrier(0);
        mma(av, bv, 3);
        __syncthreads();
    };
    constexpr std::true_type T{};
    constexpr std::false_type F{};

    // buffers rotate with period 3, operand registers with period 2: unroll by 6 so that every index is a constant
    int s = 0;
    for (; s + 6 + 3 <= nsteps; s += 6) {
        step(T, s + 0, 1, 2, avA, bvA, avB, bvB);
        step(T, s + 1, 2, 0, avB, bvB, avA, bvA);
        step(T, s + 2, 0, 1, avA, bvA, avB, bvB);
        step(T, s + 3, 1, 2, avB, bvB, avA, bvA);
        step(T, s + 4, 2, 0, avA, bvA, avB, bvB);
        step(T, s + 5, 0, 1, avB, bvB, avA, bvA);
    }
    // tail (< 9 steps): same rotation, every stage guarded
    for (; s < nsteps; s += 6) {
        step(F, s + 0, 1, 2, avA, bvA, avB, bvB);
        if (s + 1 < nsteps) step(F, s + 1, 2, 0, avB, bvB, avA, bvA);
        if (s + 2 < nsteps) step(F, s + 2, 0, 1, avA, bvA, avB, bvB);
        if (s + 3 < nsteps) step(F, s + 3, 1, 2, avB, bvB, avA, bvA);
        if (s + 4 < nsteps) step(F, s + 4, 2, 0, avA, bvA, avB, bvB);
        if (s + 5 < nsteps) step(F, s + 5, 0, 1, avB, bvB, avA, bvA);
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
        const bool padded = (a.y_pad | a.r1_pad | a.r2_pad) != 0;      // halo buffers: pixel m is not at m * Cout any more
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
            size_t moff[PB], r1off[PB], r2off[PB];
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) {
                const int m = m0 + wave * (16 * PT) + (p0 + pb) * 16 + lcol;
                mok[pb] = m < a.M;
                const unsigned mm = mok[pb] ? (unsigned)m : 0u;
                moff[pb] = r1off[pb] = r2off[pb] = (size_t)mm * (size_t)a.Cout;
                if (padded) {
                    const int n = (int)udiv(mm, a.div_hw_m, a.div_hw_s1, a.div_hw_s2);
                    const int rem = (int)mm - n * a.HWout;
                    const int ho = (int)udiv((unsigned)rem, a.div_w_m, a.div_w_s1, a.div_w_s2);
                    const int wo = rem - ho * a.Wout;
                    auto at = [&](int pad) { return (((size_t)n * (a.Hout + pad) + ho) * (a.Wout + pad) + wo) * (size_t)a.Cout; };
                    moff[pb] = at(a.y_pad);
                    r1off[pb] = at(a.r1_pad);
                    r2off[pb] = at(a.r2_pad);
                }
            }
            float4 o[CT][PB];
            if (a.res1) {
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int pb = 0; pb < PB; ++pb)
                        o[ct][pb] = *reinterpret_cast<const float4*>(a.res1 + r1off[pb] + cos[ct]);
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
                        r2[ct][pb] = *reinterpret_cast<const float4*>(a.res2 + r2off[pb] + cos[ct]);
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
                    const size_t opix = ((size_t)n * (Ho2 + a.y_pad) + h2) * (Wo2 + a.y_pad) + w2;
                    float o[4] = {v[0], v[1], v[2], v[3]};
                    if (a.res1) {
                        size_t rpix = ((size_t)n * (Ho2 + a.r1_pad) + h2) * (Wo2 + a.r1_pad) + w2;
                        if (!res1_plain) {
                            rpix = ((size_t)n * (a.res1_H + a.r1_pad) + (h2 >> a.res1_shift)) * (a.res1_W + a.r1_pad) +
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
                        const float* rp = a.res2 + (((size_t)n * (Ho2 + a.r2_pad) + h2) * (Wo2 + a.r2_pad) + w2) * a.Cout + co;
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

// 128-pixel tiles and narrower: 4 workgroups per CU (128 registers per lane)
template <int CT, int PT, bool WALK, bool FAKE = false, bool TAB = false, bool NOCHK = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void conv_p3_kernel(ConvArgs a) {
    conv_p3_body<CT, PT, WALK, FAKE, TAB, NOCHK>(a);
}

template <int CT, int PT>
int launch_t(const ConvArgs& a, hipStream_t stream, bool fake = false) {
    dim3 grid((a.M + 64 * PT - 1) / (64 * PT), (a.CoutPad + 16 * CT - 1) / (16 * CT));
    if (a.tap_table && !fake) {
        if (a.no_bounds)
            hipLaunchKernelGGL((conv_p3_kernel<CT, PT, true, false, true, true>), grid, dim3(256), 0, stream, a);
        else
            hipLaunchKernelGGL((conv_p3_kernel<CT, PT, true, false, true>), grid, dim3(256), 0, stream, a);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) {
            pp_set_error("conv_p3 (table) launch failed: %s", hipGetErrorString(e));
            return PP_ERR_HIP;
        }
        return PP_OK;
    }
    if constexpr (PT == 2 && CT >= 3) {
        if (fake && a.Cin >= BK) {
            hipLaunchKernelGGL((conv_p3_kernel<CT, PT, true, true>), grid, dim3(256), 0, stream, a);
            return PP_OK;
        }
    }
    if (a.Cin >= BK)
        hipLaunchKernelGGL((conv_p3_kernel<CT, PT, true>), grid, dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL((conv_p3_kernel<CT, PT, false>), grid, dim3(256), 0, stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        pp_set_error("conv_p3 launch failed: %s", hipGetErrorString(e));
        return PP_ERR_HIP;
    }
    return PP_OK;
}

}  // namespace

// ct in 1..4, pt in {1, 2}: the pipelined kernel for one launch (arguments prepared by pp_launch_conv)
int pp_launch_conv_p3(const ConvArgs& a, int ct, int pt, hipStream_t stream, bool fake_addresses) {
    switch (ct * 2 + (pt >= 2 ? 1 : 0)) {
        case 9: return launch_t<4, 2>(a, stream, fake_addresses);
        case 8: return launch_t<4, 1>(a, stream);
        case 7: return launch_t<3, 2>(a, stream, fake_addresses);
        case 6: return launch_t<3, 1>(a, stream);
        case 5: return launch_t<2, 2>(a, stream);
        case 4: return launch_t<2, 1>(a, stream);
        case 3: return launch_t<1, 2>(a, stream);
        default: return launch_t<1, 1>(a, stream);
    }
}
