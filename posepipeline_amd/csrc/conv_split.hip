// fp32 convolution on the bf16 matrix cores: three-way operand split, six MFMA products per term, fp32 accumulation.
//
// Why: v_mfma_f32_16x16x4_f32 peaks at 256 FLOP/clk/CU (157 TFLOP/s at 2.4 GHz); conv_igemm(_p3).hip sustain 0.76-0.85 of it and
// the detector / HRNet chunks are bound by exactly that (profiles/r02_conv_ablation.txt).  v_mfma_f32_32x32x16_bf16 runs 16x
// that rate.  A float32 value is the EXACT sum of three bfloat16 values (24 significand bits = 8 + 8 + 8, same exponent range):
//     a = a0 + a1 + a2,   a0 = bf16(a),  a1 = bf16(a - a0),  a2 = a - a0 - a1   (round to nearest even; every residual is exact)
// so a*b = sum_{i,j} ai*bj, every ai*bj exact in the MFMA's fp32 product.  Keeping the six terms with i + j <= 2 drops
// a1*b2 + a2*b1 + a2*b2 ~ 2^-26 |a*b| with random signs (|a1| <= 2^-9 |a|, |a2| <= 2^-17 |a|) -- a sixth of ONE float32
// rounding of the product, which a float32 FMA chain commits at every step anyway.  The sums are accumulated in the MFMA's float32 accumulators.  Cost: 6 bf16 MFMAs of 16 k for what
// takes 4 fp32 MFMAs of 4 k = 2.67x the arithmetic peak (419 TFLOP/s fp32-equivalent).
// Results are NOT bit-identical to oracle/conv_ref.c (different summation order and the dropped 2^-23 terms): the measured
// deviation is that of a reordered float32 sum (tests/test_gpu_split.py: same error against a float64 convolution as the
// exact kernel, and a control with the bit-exact kernels on the transposed network).  POSEPIPE_CONV_EXACT=1 / pp_conv_variant(3) selects the bit-exact fp32-MFMA kernels instead.
//
// Kernels: conv_split_kernel (3x3, stride 1, pad 1 -- 52 % of the detector's and ~85 % of HRNet's time; 4 waves with per-wave
// weight loads, or 8 waves with the weights through an LDS ring for long-K layers; also a one-tap form) and
// conv_split_gemm_kernel (1x1 layers from 256 input channels with Cout % 128 == 0, the RoI head's fc6 / fc7).
// Structure of conv_split_kernel:
// An implicit GEMM that gathers per tap re-reads every input element nine times through L1/L2; at 2.67x the MFMA rate that
// feed (87 GB/s per CU for a 128 x 64 tile) is over what the vector memory path delivers.  Instead a workgroup stages the
// input PATCH of its output tile once per 16 input channels:
//   * output tile: 8 pixel blocks of 32 pixels (one MFMA N tile each) arranged as a rectangle, x 64 output channels
//     (2 MFMA M tiles): 4 waves x (2 pixel blocks x 2 channel blocks) = 4 accumulators of 16 registers per wave;
//   * patch: (TH + 2) x (TW + 2) pixels x 16 channels, loaded as float32 (coalesced 64 B per pixel), split into the three
//     bf16 planes in registers (5.5 VALU ops per element, once per element per block) and written to LDS as
//     [plane][k half][pixel] 16-byte slots: the B fragment of tap (dy, dx) for a pixel block is ONE ds_read_b128 per plane
//     at slot (pixel + dy * pitch + dx) -- the nine taps are nine address offsets into the same patch;
//   * weights: pre-split on the device at net creation into MFMA A-fragment order ([16-channel chunk][tap][32 couts][plane]
//     [lane] x 16 B), read straight from global memory into registers, 1 KB contiguous per wave instruction, one tap ahead;
//   * one barrier per 16-channel chunk (9 taps x 24 MFMAs = 6912 matrix-pipe cycles per wave); the patch is double-buffered,
//     the float32 loads of chunk c + 1 are in flight during chunk c and split / written half-way through it.
// Traffic per 16-channel chunk and block: patch 340 x 64 B + weights 9 x 6 KB (L1 / L2 resident) = 11 B / matrix-pipe cycle.
#include <algorithm>
#include <atomic>
#include <map>
#include <mutex>
#include <type_traits>
#include <vector>
#include <cstring>
#include <cstdio>

#include "pp_internal.h"
#include "pp_amax.h"

// timing experiments only (tools/build_variant.sh; WRONG results): 1 no weight loads, 2 no patch staging, 4 no barrier,
// 8 no fragment reads from LDS, 16 no MFMAs in the K loop of conv_split_kernel, 32 no epilogue (one store per lane)
#ifndef PP_SPLIT_ABLATE
#define PP_SPLIT_ABLATE 0
#endif

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

std::atomic<int> g_split_gemm_epi{-1};       // pp_debug_knob("split_gemm_epilogue"): -1 = POSEPIPE_SPLIT_GEMM_EPI

namespace {


struct SplitArgs {
    const float* x;
    const uint4* w;          // split weights, fragment order
    const float* bias;
    const float* res1;
    const float* res2;
    float* y;
    int N, H, W, Cin, Cout, ncb;          // ncb: 32-channel blocks of the split weights
    int xp_h, xp_w;                       // stored rows per image / pixels per row of the input (dims + halo)
    int y_pad, r1_pad, r2_pad, relu;
    int nchunks;                          // Cin / 16
    int tiles_x, tiles_y, TH, TW, bw_log2, gx_log2;   // MODE_TILE
    int PWp;                              // patch pitch (TILE) / stored row pitch of the input (STREAM), in pixels
    int NP, NPp;                          // patch pixels, padded so that the two k halves are 64 B apart mod 128
    int mode, stride;
    int r1_shift, r1_H, r1_W;             // res1 is read at (y >> r1_shift, x >> r1_shift) of an r1_H x r1_W map (FPN top-down add)
    int ktaps, KW, pad, Hin, Win;         // GEMM with ktaps > 1: a KH x KW (strided) convolution as one product per (channel chunk, tap)
    long long S;                          // STREAM: positions of the padded input stream, GEMM: output pixels
    unsigned x_bytes;
    int xcd_remap;
    int SP;                               // strided-patch form: slots per phase map ((TH + 1) x PWp)
    int gx, gy;                           // xcd_remap: logical grid (pixel tiles, channel columns) of the 1-D launch
    int col_major;                        // xcd_remap: an XCD walks its tiles column by column (small inputs) instead of tile by tile
    int epi_lds;                          // conv_split_gemm_kernel: epilogue transposed through LDS (whole 128-byte lines per store)
    // host-made reciprocals (make_magic / pp_udiv) of the divisors of the tap kernels' index arithmetic: round 4's timeline showed
    // the ~700 instructions of a workgroup's set-up -- ten 32 / 64-bit division sequences among them -- taking 3.5 - 5k cycles
    // before the first load was even requested (profiles/r04_w48_timeline_before.txt)
    unsigned dv_per[2], dv_row[2];        // positions per image / per row of the output-coordinate decode (STREAM: the padded input stream)
    unsigned dv_tx[2], dv_ty[2], dv_pw[2];   // MODE_TILE: tiles_x, tiles_y, patch pitch
    unsigned dv_ncol[2], dv_run0[2], dv_run1[2];   // xcd_tile_column: gy, ntiles / 8, ntiles / 8 + 1
    unsigned dv_ktaps[2], dv_kw[2];       // tap-gather product: ktaps, KW
    // fp16 form (H kernels): per-output-channel 1 / c of the weight normalisation (behind the fragments of the split copy) and the
    // running maximum of |x| PER SAMPLE (bit pattern of a non-negative float, pp_amax.h) that the activation scale follows
    const float* wscale;
    const unsigned* x_amax;
    // any form: where to fold max |y| per sample of what this launch stores (null: the output feeds no fp16-form convolution)
    unsigned* y_amax;
#ifdef PP_SPLIT_TIMELINE
    unsigned long long* dbg;              // diagnosis builds only (tools/build_variant.sh tl -DPP_SPLIT_TIMELINE): 16 x u64 per workgroup
#endif
};

// Diagnosis builds (-DPP_SPLIT_TIMELINE, never the shipped library): every workgroup of the tap kernels records s_memtime at entry,
// after its prologue, after its K loop and at its end, plus where it ran (HW_ID / XCC_ID); the launcher synchronises after each
// launch and appends the records to $POSEPIPE_SPLIT_TIMELINE (tools/split_timeline.py reads them).
#ifdef PP_SPLIT_TIMELINE
#define PP_TL_DECL unsigned long long tl_t[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define PP_TL_MARK(i) do { if (a.dbg) tl_t[i] = __builtin_amdgcn_s_memtime(); } while (0)
#define PP_TL_FLUSH()                                                                                                    \
    do {                                                                                                                 \
        if (a.dbg && threadIdx.x == 0) {                                                                                 \
            unsigned hw, xcc;                                                                                            \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));                                            \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));                                          \
            unsigned long long* r = a.dbg + (size_t)blockIdx.x * 16;                                                     \
            r[0] = blockIdx.x; r[1] = ((unsigned long long)xcc << 32) | hw;                                              \
            r[2] = tl_t[0]; r[3] = tl_t[1]; r[4] = tl_t[2]; r[5] = tl_t[3]; r[6] = L; r[7] = col;                       \
            for (int i_ = 4; i_ < 10; ++i_) r[4 + i_] = tl_t[i_];                                                        \
        }                                                                                                                \
    } while (0)
#else
#define PP_TL_DECL
#define PP_TL_MARK(i)
#define PP_TL_FLUSH()
#endif

// n / d for any 32-bit n through the host-made pair {m, sh} = make_magic(d) (Granlund - Montgomery: k = ceil(log2 d),
// m = floor(2^32 (2^k - d) / d) + 1, q = (t + ((n - t) >> 1)) >> (k - 1) with t = mulhi(m, n); d == 1 is flagged by sh >= 32):
// five integer instructions instead of the ~40 of a 32-bit and the ~150 of a 64-bit division sequence
__device__ __forceinline__ unsigned pp_udiv(unsigned n, const unsigned (&dv)[2]) {
    const unsigned t = __umulhi(n, dv[0]);
    const unsigned q = (t + ((n - t) >> 1)) >> (dv[1] & 31u);
    return dv[1] >= 32u ? n : q;
}

// Workgroup id -> (tile, channel column) of a 1-D launch of 8 * ceil(ntiles / 8) * ncol ids.  Consecutive ids go round-robin over
// the 8 XCDs (each with its own L2).  XCD x owns a CONTIGUOUS run of tiles (neighbouring tiles share their halo rows in its L2)
// and walks it tile by tile, all channel columns of a tile one after the other: the columns read the same input tile, which
// then comes from HBM once instead of once per column (256 -> 1024 at 40x68 has 8 columns, 512 -> 2048 16).  false: idle id.
// col_major != 0 (small maps: the whole input sits in the Infinity Cache anyway, and what an XCD's L2 should keep is ONE column's
// split weights): the XCD walks its run once per column instead -- measured on HRNet's 192 -> 192 at 24x18: 187 vs 166 TFLOP/s.
__device__ __forceinline__ bool xcd_tile_column(unsigned L, const SplitArgs& a, unsigned& tile, unsigned& col) {
    const unsigned ntiles = (unsigned)a.gx, ncol = (unsigned)a.gy;
    const unsigned xcd = L & 7u, j = L >> 3;
    const unsigned q = ntiles >> 3, r = ntiles & 7u;
    const unsigned run = xcd < r ? q + 1 : q;
    unsigned tj;
    if (a.col_major) {
        if (run == 0) return false;
        col = xcd < r ? pp_udiv(j, a.dv_run1) : pp_udiv(j, a.dv_run0);
        tj = j - col * run;
        if (col >= ncol) return false;
    } else {
        tj = pp_udiv(j, a.dv_ncol);
        col = j - tj * ncol;
        if (tj >= run) return false;
    }
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + tj;
    return true;
}

// Round-to-nearest-even split of four floats into the three bf16 planes (4 x 16 bit each): p0 = bf16(a), p1 = bf16(a - p0),
// p2 = a - p0 - p1.  The residuals are exact float32 values (a - p0 has at most 16 significant bits, a - p0 - p1 at most 8), so
// a = p0 + p1 + p2 EXACTLY, as with a truncating split -- but the planes below the first are SIGNED with half the magnitude
// (|p1| <= 2^-9 |a|, |p2| <= 2^-17 |a|): the three dropped products p1*q2 + p2*q1 + p2*q2 are ~2^-26 |a q| with random signs.
// Round 2's truncating split kept every plane the sign of a, so on same-sign data (ReLU activations x positive kernels) the
// dropped terms, 2^-22 |a q| on average, all pushed the same way: -2e-7 per layer, -1.3e-5 over HRNet-W48's ~70 layers
// (tests/test_gpu_parity_modes.py found it as a uniform score deficit).  v_cvt_pk_bf16_f32 converts and packs two values per
// instruction: 4.5 vector instructions per element (5.5 with masks and v_perm).  Values within 2^-8 of FLT_MAX would round to
// infinity; activations and weights are nowhere near.
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ void split4(const float4 v, uint2& p0, uint2& p1, uint2& p2) {
    const f32x2_t a = {v.x, v.y}, b = {v.z, v.w};
    const bf16x2_t a0 = __builtin_convertvector(a, bf16x2_t), b0 = __builtin_convertvector(b, bf16x2_t);
    const f32x2_t ra = a - __builtin_convertvector(a0, f32x2_t), rb = b - __builtin_convertvector(b0, f32x2_t);          // exact
    const bf16x2_t a1 = __builtin_convertvector(ra, bf16x2_t), b1 = __builtin_convertvector(rb, bf16x2_t);
    const f32x2_t sa = ra - __builtin_convertvector(a1, f32x2_t), sb = rb - __builtin_convertvector(b1, f32x2_t);        // exact, <= 8 bits
    const bf16x2_t a2 = __builtin_convertvector(sa, bf16x2_t), b2 = __builtin_convertvector(sb, bf16x2_t);              // exact
    // a bf16x2 register = (element 2i+1) << 16 | element 2i
    p0 = make_uint2(__builtin_bit_cast(unsigned, a0), __builtin_bit_cast(unsigned, b0));
    p1 = make_uint2(__builtin_bit_cast(unsigned, a1), __builtin_bit_cast(unsigned, b1));
    p2 = make_uint2(__builtin_bit_cast(unsigned, a2), __builtin_bit_cast(unsigned, b2));
}

// ---- fp16 form (round 5): two-term operand split, THREE products per term ------------------------------------------------------
// The six-product form's ceiling is 2500 / 6 = 417 TFLOP/s fp32-equivalent, and two rounds of tuning put the kernels at 0.46 of it
// with the long-K layers at the chip's power limit.  float16 carries 11 significand bits: two terms carry 22,
//     x s = h0 + h1,   h0 = f16(x s),  h1 = f16(x s - h0)     (s: power of two PER SAMPLE, max |x| s in [2^14, 2^15), exact)
//     w c = g0 + g1,   g0 = f16(w c),  g1 = f16(w c - g0)     (c: power of two PER OUTPUT CHANNEL, max |w| c in [2^14, 2^15))
//     (x s)(w c) ~ h0 g0 + h1 g0 + h0 g1                      (dropped: h1 g1 ~ 2^-22 |x w|, random sign)
// -- three v_mfma_f32_32x32x16_f16 (same rate as bf16) into ONE float32 accumulator set, TWO planes per operand (the weight planes
// are a third smaller than the bf16 form's: LDS reads per tap 10 -> 8, weight bytes through L2 / the DMA ring -33 %); the epilogue
// multiplies by 1 / (s c) (exact) inside the bias add's fma.  Representation error: |x s - h0 - h1| <= 2^-22 |x s| wherever the
// residual is a normal float16 (|x s| >= 2^-2, i.e. down to 2^-17 of the sample's maximum) and <= 2^-25 absolutely = 2^-39 of the
// sample's maximum below that -- far under what ONE float32 rounding of the running sum costs (2^-24 of the sum); the weights
// likewise per channel.  Per product that is ~2^-23.7 rms with
// random sign against the ~2^-24 EVERY step of a float32 FMA chain commits on the running sum: over K >= 144 terms the chain's own
// rounding dominates (tests/test_gpu_split.py holds the same yardstick as for the six-product form: error against float64 <= the
// float32 kernel's).  Range of activations: s = 2^k is chosen PER SAMPLE from the running maximum of the tensor (pp_amax.h: the kernel
// that stores a tensor folds max |v| into amax[sample]; max |x| s lies in [2^14, 2^15)), so nothing can leave the format upwards,
// values down to 2^-17 of the sample's maximum keep their 22 bits and smaller ones are off by <= 2^-39 of it -- whatever the
// magnitude of the data (1e-30 .. 1e30 alike).  Non-finite inputs stay non-finite (inf s = inf in float16, NaN stays NaN).
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
// (round 6) two elements -> their (h0, h1) register pair in FOUR instructions: v_fma_mixlo / mixhi_f16 evaluate fma(x, s, c) in float32
// with c read as float16 and round the result into one half of the destination -- h0 = f16(x s + 0), h1 = f16(x s - h0) -- where the
// plain sequence takes six (v_pk_mul, v_cvt_pk_f16_f32, two v_cvt_f32_f16, v_pk_add, v_cvt_pk_f16_f32).  x s is exact (s a power of two)
// and so is the difference: the same bits as before, except the sign of an exact zero (tools/dbg/mix_test.hip: 4M random bit patterns,
// four scales, 0 mismatches).  The split is the VALU work of every fp16-form kernel: the product kernel converts at each fragment read
// (ablation: 3.3 ms of a 105 ms step go into it), the strided-patch kernel stages four input pixels per output (3.6 VALU per MFMA).
__device__ __forceinline__ void split2h(const float x0, const float x1, const float s, unsigned& h0, unsigned& h1) {
    unsigned d, e;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(d) : "v"(x0), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(d) : "v"(x1), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(e) : "v"(x0), "v"(s), "v"(d));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(e) : "v"(x1), "v"(s), "v"(d));
    h0 = d;
    h1 = e;
}
__device__ __forceinline__ void split4h(const float4 v, const float s, uint2& p0, uint2& p1) {
    split2h(v.x, v.y, s, p0.x, p1.x);
    split2h(v.z, v.w, s, p0.y, p1.y);
}

// Activations of the DeepSortYOLOv4 / YOLOX programs, as conv_igemm_p3.hip: every transcendental is evaluated in double precision
// and rounded to float once (what oracle/yolo.py restates); applied like PP_RELU_FIRST: y = res + act(conv + bias).
__device__ __forceinline__ float split_activate(float x, int act) {
    if (act == PP_ACT_LEAKY) return x >= 0.f ? x : 0.1f * x;
    if (act == PP_ACT_MISH) {
        if (x > 20.f) return x;
        const double n = exp((double)x);
        const double t = n * (n + 2.0);
        return x * (float)(t / (t + 2.0));
    }
    if (act == PP_ACT_ELU) return x > 0.f ? x : (float)expm1((double)x);
    if (act == PP_ACT_SWISH) return x * (float)(1.0 / (1.0 + exp(-(double)x)));
    return x;
}

// ds_read_b128 is serviced in the lane groups {0-3, 12-15, 20-27} and {4-11, 16-19, 28-31} (+32): lane -> q such that each
// group is 16 consecutive q.  Pixel blocks narrower than 32 pixels give every group whole rows of the block.
__device__ __forceinline__ int lane_q(int r) {
    const int g1 = ((r >= 4 && r < 12) || (r >= 16 && r < 20) || r >= 28) ? 1 : 0;
    int idx;
    if (!g1) idx = r < 4 ? r : r < 16 ? r - 8 : r - 12;              // 0-3 -> 0-3, 12-15 -> 4-7, 20-27 -> 8-15
    else idx = r < 12 ? r - 4 : r < 20 ? r - 8 : r - 16;             // 4-11 -> 0-7, 16-19 -> 8-11, 28-31 -> 12-15
    return g1 * 16 + idx;
}
// position of block pixel q inside a block of width 32 >> bw_log2 ... (bw_log2: 0 = 1 x 32, 1 = 2 x 16, 2 = 4 x 8 pixels)
__device__ __forceinline__ void block_pixel(int r, int bw_log2, int& iy, int& ix) {
    if (bw_log2 == 0) {
        iy = 0;
        ix = r;
        return;
    }
    const int q = lane_q(r);
    if (bw_log2 == 1) {
        iy = q >> 4;
        ix = q & 15;
    } else {            // rows 0, 2 in the first lane group, 1, 3 in the second (patch pitch = 4 mod 8: conflict-free)
        iy = ((q >> 3) & 1) * 2 + (q >> 4);
        ix = q & 7;
    }
}

// MODE_TILE   3x3 / stride 1 / pad 1 on any buffer: rectangular output tile, patch = tile + 1-pixel frame (bounds-tested loads)
// MODE_STREAM 3x3 / stride 1 / pad 1 on a zero-halo input (pp_buf.pad >= 1): the padded tensor is ONE pixel stream in which tap
//             (dy, dx) of position s is position s + dy * pitch + dx (the halo supplies every out-of-image zero), so a tile is
//             256 CONSECUTIVE stream positions whatever the map's width -- no tile quantisation on HRNet's 36 / 18 / 9-pixel rows;
//             patch = the 256 + 2 pitch + 2 positions around it, loaded without any test; halo positions are not stored
// MODE_GEMM   1x1 (any stride), and 'valid' convs whose kernel covers the whole input (the RoI head's 7x7 fc6, rewritten by
//             the launcher as 1x1 over 49 x 256 channels): tile = 256 consecutive output pixels, one "tap".
//             ktaps > 1 (round 3: 3x3 stride 2): the same product form, one step per (16-channel chunk, tap) -- the operand of a
//             step is the tile's 256 input pixels AT THAT TAP (gathered with the stride, out-of-image taps read as zero through
//             a per-slot validity mask).  No patch reuse between taps (a strided patch is 4x the tile: 117 KB of LDS per stage),
//             so every input element is staged 2.25 times instead of once; at 16 KB per step that is ~17 % of the vector
//             memory path, and the layers leave the fp32 MFMA kernels (95 - 130 TFLOP/s) for this kernel's one-tap rate.
enum { MODE_TILE = 0, MODE_STREAM = 1, MODE_GEMM = 2 };

// COB: output-channel blocks (32 channels) per wave and per workgroup; PXB: pixel blocks (32 pixels) per wave.  PXB = 1
// (128-pixel tiles, ~150 registers, three workgroups per CU) was measured for the layers whose grid does not fill the chip
// twice: 24x18 x 192 channels +9 %, 12x9 x 384 -20 %, 40x68 x 256 -17 % alone; no gain in the 4-stream program -- not instantiated.
//
// NW = 8 (512 threads, one workgroup per CU, 512-pixel tiles) is the form for layers with enough tiles: there the split weights go
// through LDS -- a 4-slot ring of per-tap fragment sets filled by the DMA path (global_load_lds_dwordx4, one 1 KB fragment per
// wave and tap), one barrier per tap.  Why: with per-wave fragment loads straight from global memory (NW = 4) every wave of a
// workgroup pulls the same 6 KB per tap through the texture-address path; an ablation build without those loads runs the
// 256 -> 256 layer at 307 instead of 228 TFLOP/s (profiles/r02_conv_split_ablation.txt), i.e. they are the largest single loss.
//
// RING4 (round 3; NW = 4, 3x3 tile / stream forms): the layers with too few chunks or tiles for the 8-wave form (all of HRNet) get
// the weights through the LDS ring, too, while keeping TWO workgroups per CU -- what made the 8-wave form lose there was one
// workgroup per CU sitting at its per-tap barrier with nothing else to run.  LDS per workgroup: the patch becomes SINGLE-buffered
// (33 - 40 KB: chunk c + 1 waits in registers, as before, and is written during the last tap of chunk c, after the barrier that
// closed tap 7 has seen every fragment read of the old patch complete) + the 4-slot ring (12 / 24 KB) = <= 73 KB, two per CU.
// Every wave copies ceil(COB * 3 / 4) fragments per tap (duplicates where 4 does not divide: the counts must be wave-uniform).
// H (round 5): the fp16 form -- TWO activation planes (h0, h1) in the patch, three weight planes (g0, g1, g2) in the same fragment
// order and ring, three products per term instead of six; the epilogue multiplies by 1 / (s c) per output channel.
// S2 (round 6; fp16 form, RING4, PXB = 1): 3x3 / STRIDE 2 / pad 1 with a strided patch.  Tap (dy, dx) of output pixel (y, x) is input
// pixel (2y + dy - 1, 2x + dx - 1): in stored coordinates r = 2y + dy, c = 2x + dx, i.e. PHASE (dy & 1, dx & 1) of the input at position
// (y + (dy >> 1), x + (dx >> 1)).  The patch of a TH x TW output tile -- (2 TH + 1) x (2 TW + 1) input pixels -- is therefore written to LDS
// DE-INTERLEAVED into its four phases, each a (TH + 1) x (TW + 1) map of pitch PWp: within one phase neighbouring output pixels are
// neighbouring slots, and the nine taps are nine offsets (phase base + shift) exactly as in the stride-1 kernel -- same fragment reads,
// same weight fragments and ring, every input element staged ONCE per channel column and no padding products (the tap-gather product
// staged every element 2.25 times; 95 - 210 TFLOP/s).  The de-interleave costs nothing at the LDS side: slot q of the patch is phase-major
// and the GLOBAL address of a slot is computed from it (neighbouring slots load pixels two apart).  128-pixel tiles (the strided patch
// of 256 would need 18 float4 of staging registers per thread): two workgroups per CU with the single-buffered patch (<= 40 KB) + ring.
template <int T, int NSLOT, int COB, int PXB = 2, int NW = 4, bool RING4 = false, bool H = false, bool S2 = false>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void conv_split_kernel(SplitArgs a) {
    static_assert(!S2 || (T == 9 && RING4 && H && PXB == 1), "S2 is the 4-wave fp16 ring form with one pixel block per wave");
    constexpr int NT = 64 * NW;
    constexpr int XP = H ? 2 : 3;                     // activation planes
    constexpr int WPL = H ? 2 : 3;                    // weight planes (H: g0, g1 -- the third product reuses g0)
    constexpr bool WLDS = NW == 8 || RING4;
    static_assert(!RING4 || (NW == 4 && T == 9), "RING4 is the 4-wave 3x3 form");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    unsigned L = blockIdx.x, col = blockIdx.y;
    PP_TL_DECL;
    PP_TL_MARK(0);
    if (a.xcd_remap) {      // 1-D launch, see xcd_tile_column
        if (!xcd_tile_column(blockIdx.x, a, L, col)) return;
    }
    const int cb0 = (int)col * COB;
    const int plane_bytes = 2 * a.NPp * 16;           // [half][pixel] x 16 B
    const int buf_bytes = XP * plane_bytes;

    // tile origin (32-bit throughout: one launch addresses < 4 GiB of input, so positions and pixels stay below 2^30)
    int n = 0, x0 = 0, y0 = 0;
    unsigned s0 = 0;                                   // STREAM: first stream position, GEMM: first output pixel
    if (a.mode == MODE_TILE) {
        const unsigned L2 = pp_udiv(L, a.dv_tx);
        const int tx = (int)(L - L2 * (unsigned)a.tiles_x);
        n = (int)pp_udiv(L2, a.dv_ty);
        const int ty = (int)(L2 - (unsigned)n * (unsigned)a.tiles_y);
        x0 = tx * a.TW;
        y0 = ty * a.TH;
    } else {
        s0 = L * (unsigned)(32 * PXB * NW);
    }

    // ---- patch loader: slot j of this thread = (patch pixel p, channel quad) -----------------------------------------
    // Set-up order (round 4): what the FIRST LOADS need comes first and is lean -- the loads are requested before the rest of the
    // index arithmetic (fragment addresses, output coordinates), whose cost then hides behind their latency
    unsigned goff[NSLOT];
    unsigned vmask[T == 1 ? NSLOT : 1];               // GEMM with taps: bit t = tap t of this slot's pixel lies inside the image
    int simg[H ? NSLOT : 1];                          // fp16 form: the sample slot j belongs to (its activation scale)
    // LDS byte offset (plane 0) of slot j: woff0 + (NT / 4) * 16 j (NT / 4 pixels further, same quad)
    const int woff0 = ((((tid & 3) >> 1) * a.NPp + (tid >> 2)) * 16 + (tid & 1) * 8);
    if (a.mode == MODE_STREAM) {
        // position of slot j = pos0 + (NT / 4) j.  Positions before the stream (first tile only) are masked; positions past it
        // need no test: their byte offset is past the descriptor's range and reads as zero (x_bytes = the stream's size), and
        // a slot past the patch (only the last one can be) is loaded but never stored
        const int pos0 = (int)s0 - a.PWp - 1 + (tid >> 2);
        const unsigned cin4 = (unsigned)a.Cin * 4u;
        const unsigned base = (unsigned)pos0 * cin4 + (unsigned)(tid & 3) * 16u;
#pragma unroll
        for (int j = 0; j < NSLOT; ++j) goff[j] = pos0 + (NT / 4) * j < 0 ? 0xffffffffu : base + (unsigned)((NT / 4) * j) * cin4;
        if constexpr (T == 1) {
#pragma unroll
            for (int j = 0; j < NSLOT; ++j) vmask[j] = 0;
        }
        if constexpr (H) {       // halo positions carry zeros: whichever neighbouring sample they are counted to
#pragma unroll
            for (int j = 0; j < NSLOT; ++j) {
                const int pj = pos0 + (NT / 4) * j;
                simg[j] = (int)pp_udiv((unsigned)min(max(pj, 0), (int)a.S - 1), a.dv_per);
            }
        }
    } else {
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) {
        const int u = tid + NT * j;
        const int p = u >> 2, quad = u & 3;
        const bool in_patch = p < a.NP;
        unsigned off = 0xffffffffu;
        if constexpr (H) simg[j] = n;
        if constexpr (S2) {
            // slot p = (phase, row, column) of the de-interleaved patch (dv_row holds the reciprocal of SP in this form)
            const int ph = (int)pp_udiv((unsigned)p, a.dv_row), r = p - ph * a.SP;
            const int qr = (int)pp_udiv((unsigned)r, a.dv_pw), qc = r - qr * a.PWp;
            const int pr = 2 * qr + (ph >> 1), pc = 2 * qc + (ph & 1);
            const int iy = 2 * y0 - 1 + pr, ix = 2 * x0 - 1 + pc;
            if (in_patch && pc <= 2 * a.TW && (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win)
                off = (unsigned)(((n * a.xp_h + iy) * a.xp_w + ix) * a.Cin) * 4u;
        } else if (a.mode == MODE_TILE) {
            const int pr = (int)pp_udiv((unsigned)p, a.dv_pw), pc = p - pr * a.PWp;
            const int iy = y0 - 1 + pr, ix = x0 - 1 + pc;
            if (in_patch && pc < a.TW + 2 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
                off = (unsigned)(((n * a.xp_h + iy) * a.xp_w + ix) * a.Cin) * 4u;
        } else {
            const unsigned m = s0 + (unsigned)p;
            if constexpr (T == 1) vmask[j] = 0;
            if (in_patch && m < (unsigned)a.S) {
                const int img = (int)pp_udiv(m, a.dv_per), rem = (int)(m - (unsigned)img * (unsigned)(a.H * a.W));
                const int ho = (int)pp_udiv((unsigned)rem, a.dv_row), wo = rem - ho * a.W;
                if constexpr (H) simg[j] = img;
                if (T == 1 && a.ktaps > 1) {
                    // origin of the pixel's window (may lie before the image: the offset wraps, a valid tap's sum is exact again)
                    const int iy0 = ho * a.stride - a.pad, ix0 = wo * a.stride - a.pad;
                    off = (unsigned)(((img * a.xp_h + iy0) * a.xp_w + ix0) * a.Cin) * 4u;
                    unsigned vm = 0;
                    for (int t = 0, dy = 0, dx = 0; t < a.ktaps; ++t) {
                        if ((unsigned)(iy0 + dy) < (unsigned)a.Hin && (unsigned)(ix0 + dx) < (unsigned)a.Win) vm |= 1u << t;
                        if (++dx == a.KW) { dx = 0; ++dy; }
                    }
                    if constexpr (T == 1) vmask[j] = vm;
                    if (off == 0xffffffffu) off = 0xfffffffeu;      // (never a multiple of 16; keeps the sentinel unambiguous)
                } else {
                    off = (unsigned)(((img * a.xp_h + ho * a.stride) * a.xp_w + wo * a.stride) * a.Cin) * 4u;
                }
            }
        }
        goff[j] = off == 0xffffffffu ? off : off + (unsigned)quad * 16u;
    }
    }
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)a.x_bytes, 0x00020000);
    float4 xr[NSLOT];
    auto load_patch = [&](int c) {
        unsigned add = (unsigned)c * 64u;
        int tbit = -1;
        if (T == 1 && a.ktaps > 1) {                  // step c = (channel chunk c / ktaps, tap c % ktaps)
            const int cc = (int)pp_udiv((unsigned)c, a.dv_ktaps), t = c - cc * a.ktaps;
            const int dy = (int)pp_udiv((unsigned)t, a.dv_kw), dx = t - dy * a.KW;
            add = (unsigned)(((dy * a.xp_w + dx) * a.Cin + 16 * cc) * 4);
            tbit = t;
        }
#pragma unroll
        for (int j = 0; j < NSLOT; ++j) {
            bool ok = goff[j] != 0xffffffffu;
            if constexpr (T == 1) {
                if (tbit >= 0) ok = ((vmask[j] >> tbit) & 1u) != 0;
            }
            const unsigned off = ok ? goff[j] + add : 0xffffffffu;
            xr[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)off, 0, 0));
        }
    };
    // split + write slots [j0, j1) of the staged chunk; branch-free (it sits between MFMAs) except for the last slot, the only
    // one that can be partly outside the patch (NSLOT = ceil(NP / (NT / 4)))
    const bool last_ok = (tid >> 2) + (NT / 4) * (NSLOT - 1) < a.NP;
    // fp16 form: the activation scale of slot j's sample is a power of two: its EXPONENT byte (pp_amax_exp of the sample's maximum), four
    // slots to a register (round 6: NSLOT floats were live through the K loop), set behind the first requests
    unsigned sxe[H ? (NSLOT + 3) / 4 : 1];
    auto store_patch = [&](int buf, int j0, int j1) {
#pragma unroll
        for (int j = 0; j < NSLOT; ++j) {
            if (j < j0 || j >= j1) continue;
            uint2 p0, p1, p2;
            if constexpr (H) split4h(xr[j], __uint_as_float((268u - ((sxe[j >> 2] >> (8 * (j & 3))) & 0xffu)) << 23), p0, p1);
            else split4(xr[j], p0, p1, p2);
            if (j == NSLOT - 1 && !last_ok) continue;
            unsigned char* d = smem + buf * buf_bytes + woff0 + (NT / 4) * 16 * j;
            *reinterpret_cast<uint2*>(d) = p0;
            *reinterpret_cast<uint2*>(d + plane_bytes) = p1;
            if constexpr (!H) *reinterpret_cast<uint2*>(d + 2 * plane_bytes) = p2;
        }
    };
    // ---- the first requests: weights of the first steps (DMA ring) / first step (registers), patch of chunk 0 -----------------
    PP_TL_MARK(4);
    constexpr int WSLOT = COB * WPL * 1024;
    constexpr int NDMA = (COB * WPL + NW - 1) / NW;  // 1 KB fragments each wave copies per step (8 waves: 1; 4 waves: 2 / 1 for COB 2 / 1)
    const unsigned wring = (unsigned)((RING4 ? 1 : 2) * buf_bytes);
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int nsteps = a.nchunks * T;
    const size_t wstep = (size_t)a.ncb * WPL * 64;       // uint4 per step
    // step s = chunk * T + tap.  Ring slot s & 3 holds the COB x 3 fragments of step s.
    auto issue_w = [&](int step) {
        if (step >= nsteps) return;
#pragma unroll
        for (int i = 0; i < NDMA; ++i) {
            const int myslab = (wave + NW * i) % (COB * WPL);      // (waves past COB * 3 copy a duplicate: same bytes, same place)
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + wring + (unsigned)((step & 3) * WSLOT + myslab * 1024));
            const uint4* g = a.w + (size_t)cb0 * (WPL * 64) + myslab * 64 + lane + (size_t)step * wstep;
            // raw instruction, see conv_split_gemm_kernel: the builtin makes the compiler order every later ds_read behind vmcnt(0)
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(dst) : "memory", "m0");
        }
    };
    // weights: fragment (step, channel block cb, plane) at ((step * ncb + cb) * 3 + plane) * 64 + lane, step = chunk * T + tap
    const uint4* wlane = a.w + (size_t)cb0 * WPL * 64 + lane;
    // two register sets of weight fragments (step s + 1 is read while step s multiplies); FOUR channel blocks per wave have room for one
    // only: its fragments of step s + 1 are read behind the last MFMA of step s, in front of the closing barrier
    constexpr int WSETS = COB >= 4 ? 1 : 2;
    uint4 wf[WSETS][COB][WPL];
    uint4 xf[PXB][XP];
    const uint4* wp = wlane;                           // weights of the next step to fetch (one spare step at the end of the buffer)
    auto load_w = [&](uint4 (&dst)[COB][WPL]) {
#pragma unroll
        for (int cb = 0; cb < COB; ++cb)
#pragma unroll
            for (int pl = 0; pl < WPL; ++pl) dst[cb][pl] = wp[(cb * WPL + pl) * 64];
        wp += wstep;
    };
    if constexpr (WLDS) {
        issue_w(0);
        issue_w(1);
        issue_w(2);
        load_patch(0);
    } else {
        load_patch(0);
        load_w(wf[0]);
    }
    // fp16 form: the maxima of the slots' samples are REQUESTED here, with the first loads, and turned into scales behind the index
    // arithmetic below (consuming them here would put a full wait in front of it)
    unsigned xam[H ? NSLOT : 1];
    if constexpr (H) {
#pragma unroll
        for (int j = 0; j < NSLOT; ++j) xam[j] = a.x_amax[simg[j]];
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- operand addresses (behind the first loads) and output pixels ------------------------------------------------------------------
    // (round 6) a lane's output pixels are needed by the EPILOGUE only: computing them here kept 8 - 10 registers alive through the K loop of
    // kernels that sit at the 256-register limit (the fp16-form 3 / 4-block kernels spilled 12 - 124 bytes per lane).  The same arithmetic
    // runs once before the loop for the LDS offsets and once after it -- from a laundered thread id, so that the compiler does not keep
    // the first evaluation -- for the coordinates.
    auto pixel_coords = [&](int lane_, int wave_, int pb, int& aofs_, int& on_, int& oy_, int& ox_, bool& ok_) {
        if (a.mode == MODE_TILE) {
            const int BW = 32 >> a.bw_log2, BH = 1 << a.bw_log2;
            int by, bx;
            block_pixel(lane_ & 31, a.bw_log2, by, bx);
            const int b = wave_ * PXB + pb;
            const int gy = b >> a.gx_log2, gx = b & ((1 << a.gx_log2) - 1);
            const int ry = gy * BH + by, rx = gx * BW + bx;
            aofs_ = (((lane_ >> 5) * a.NPp) + ry * a.PWp + rx) * 16;
            on_ = n;
            oy_ = y0 + ry;
            ox_ = x0 + rx;
            ok_ = oy_ < a.H && ox_ < a.W;
        } else {
            // STREAM: (image, row, column) of a position of the padded input stream; GEMM: of an output pixel (dv_per / dv_row hold
            // the reciprocals of the positions per image / per row of that mode)
            const unsigned per = a.mode == MODE_STREAM ? (unsigned)(a.xp_h * a.PWp) : (unsigned)(a.H * a.W);
            const unsigned row = a.mode == MODE_STREAM ? (unsigned)a.PWp : (unsigned)a.W;
            const int pl = (wave_ * PXB + pb) * 32 + (lane_ & 31);
            aofs_ = (((lane_ >> 5) * a.NPp) + pl) * 16;
            const unsigned sp = s0 + (unsigned)pl;
            const bool in = sp < (unsigned)a.S;
            const unsigned sc = in ? sp : 0u;
            const unsigned img = pp_udiv(sc, a.dv_per);
            const unsigned rem = sc - img * per;
            const unsigned yy = pp_udiv(rem, a.dv_row);
            on_ = (int)img;
            oy_ = (int)yy;
            ox_ = (int)(rem - yy * row);
            ok_ = in && oy_ < a.H && ox_ < a.W;
        }
    };
    int aofs[PXB];           // LDS byte offset of this lane's pixel (tap (0,0), plane 0) per pixel block
#pragma unroll
    for (int pb = 0; pb < PXB; ++pb) {
        int on_, oy_, ox_;
        bool ok_;
        pixel_coords(lane, wave, pb, aofs[pb], on_, oy_, ox_, ok_);
    }

    f32x16 acc[COB][PXB];
#pragma unroll
    for (int cb = 0; cb < COB; ++cb)
#pragma unroll
        for (int pb = 0; pb < PXB; ++pb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[cb][pb][i] = 0.f;
    if constexpr (H) {
#pragma unroll
        for (int q = 0; q < (NSLOT + 3) / 4; ++q) sxe[q] = 0;
#pragma unroll
        for (int j = 0; j < NSLOT; ++j) sxe[j >> 2] |= pp_amax_exp(xam[j]) << (8 * (j & 3));
    }

    auto load_x = [&](const unsigned char* pbuf, int t, int pb) {
        const int dy = t / 3, dx = t % 3;
        const int toff = S2 ? ((((dy & 1) * 2 + (dx & 1)) * a.SP) + (dy >> 1) * a.PWp + (dx >> 1)) * 16 : T == 9 ? (dy * a.PWp + dx) * 16 : 0;
#pragma unroll
        for (int pl = 0; pl < XP; ++pl) xf[pb][pl] = *reinterpret_cast<const uint4*>(pbuf + pl * plane_bytes + aofs[pb] + toff);
    };

    // ---- prologue (4 waves; the 8-wave form has its own below) -------------------------------------------------------------
    if constexpr (!WLDS) {
        store_patch(0, 0, NSLOT);                      // (patch 0 and the first weights were requested above)
        PP_TL_MARK(5);
        if (a.nchunks > 1) load_patch(1);
        __syncthreads();
#pragma unroll
        for (int pb = 0; pb < PXB; ++pb) load_x(smem, 0, pb);
    }

    // the six products with i + j <= 2 (smallest terms first) of one pixel block against every channel block; consecutive MFMAs
    // go to different accumulators (no back-to-back dependency on the matrix pipe)
    auto mma = [&](const uint4 (&wc)[COB][WPL], int pb) {
        if constexpr (H) {       // g1 h0 + g0 h1 + g0 h0
            constexpr int WI[3] = {1, 0, 0}, XI[3] = {0, 1, 0};
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int cb = 0; cb < COB; ++cb)
                    acc[cb][pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wc[cb][WI[p]]),
                                                                         __builtin_bit_cast(f16x8, xf[pb][XI[p]]), acc[cb][pb], 0, 0, 0);
        } else {
        constexpr int WI[6] = {2, 1, 0, 1, 0, 0}, XI[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
        for (int p = 0; p < 6; ++p)
#pragma unroll
            for (int cb = 0; cb < COB; ++cb)
                acc[cb][pb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wc[cb][WI[p]]),
                                                                      __builtin_bit_cast(bf16x8, xf[pb][XI[p]]), acc[cb][pb], 0, 0, 0);
        }
    };

    if constexpr (WLDS) {
        // ---- 8 waves: split weights through a 4-slot LDS ring, one barrier per tap ---------------------------------------------
        // step s = chunk * T + tap.  Ring slot s & 3 holds the COB x 3 fragments of step s.  During step s: the fragments of step
        // s + 1 are read into the other register set (they landed before the barrier that ended step s - 1), the DMA of step
        // s + 3 is issued into the slot step s - 1 used, and before the closing barrier the DMA of step s + 2 is awaited -- by
        // counting: vmcnt is in issue order, so "all but the DMA of s + 3 and the patch loads issued after it" have landed.
        auto read_w = [&](int step, uint4 (&dst)[COB][WPL]) {
            const unsigned char* base = smem + wring + (step & 3) * WSLOT + lane * 16;
#pragma unroll
            for (int cb = 0; cb < COB; ++cb)
#pragma unroll
                for (int pl = 0; pl < WPL; ++pl) dst[cb][pl] = *reinterpret_cast<const uint4*>(base + (cb * WPL + pl) * 1024);
        };
        // allow `n` (0, NDMA, NSLOT, NSLOT + NDMA) of the youngest vector-memory operations to stay in flight
        auto wait_all_but = [&](int n) {
            // (8-wave form, no lgkmcnt: a wave's patch writes and fragment reads are complete long before the barriers that matter
            // for them -- the patch of chunk c + 1 is written four taps before its first read, a ring slot is reused two barriers
            // after its last read was consumed.  RING4 reuses its single patch buffer at once and waits for lgkmcnt where it must.)
            if (n == NSLOT + NDMA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSLOT + NDMA) : "memory");
            else if (n == NSLOT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSLOT) : "memory");
            else if (n == NDMA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
        // (the DMAs of steps 0 - 2 and patch 0 were requested above, in this order)
        store_patch(0, 0, NSLOT);                      // waits for patch 0, hence (in order) for the three DMAs before it
        PP_TL_MARK(5);
        if (a.nchunks > 1) load_patch(1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        read_w(0, wf[0]);
#pragma unroll
        for (int pb = 0; pb < PXB; ++pb) load_x(smem, 0, pb);

        auto chunk8 = [&](auto par, int c) {
            constexpr int PAR = decltype(par)::value;
            const unsigned char* pbuf = smem + (c & 1) * buf_bytes;
            const int st0 = c * T;
            const bool more_patch = c + 1 < a.nchunks;            // patch c + 1 is staged in this chunk
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int cur = (PAR + t) & 1;
                const int st = st0 + t;
#pragma unroll
                for (int pb = 0; pb < PXB; ++pb) {
                    if (pb == PXB - 1 && t == T / 2) {
                        // patch c + 1: split and written half-way through the chunk; the DMA of this step is issued AFTER it, so
                        // that the compiler's wait for the patch loads (it does not see the DMAs) covers no DMA younger than a tap
                        if (more_patch) store_patch((c + 1) & 1, 0, NSLOT);
                        issue_w(st + 3);
                    }
                    mma(wf[cur % WSETS], pb);
                    __builtin_amdgcn_sched_barrier(0);
                    if (pb == 0) {
                        // (round 4) the next step's fragment reads and the DMA request sit BEHIND the first MFMAs of the step: straight
                        // after the barrier both waves of a SIMD are here at once, and everything issued before the first MFMA is
                        // matrix-pipe idle time.  Same-box A/B (profile_net det 64 / w48 128): 256 -> 256 at 160x272 259.9 -> 264.5
                        // TFLOP/s, 80x136 227 -> 233, 40x68 221 -> 228, HRNet 192 -> 192 179 -> 189.  Moving the BARRIER itself behind the
                        // first MFMA group as well (the group needs nothing the barrier guarantees) measured no further gain (260 vs 260):
                        // the loop runs at the chip's power limit there, a saved cycle comes back as a lower clock.
                        if (st + 1 < nsteps) read_w(st + 1, wf[(cur ^ 1) % WSETS]);
                        if (t != T / 2) issue_w(st + 3);
                    }
                    if (t + 1 < T) load_x(pbuf, t + 1, pb);
                    __builtin_amdgcn_sched_barrier(0);
                }
                // closing barrier of the step: the DMA of step st + 2 has landed (younger: the DMA of st + 3 if there is one, and at
                // the first tap of a chunk the patch loads issued at its start); LDS writes of store_patch are complete
                {
                    const int younger = (st + 3 < nsteps ? NDMA : 0) + ((t == 0 && c + 1 < a.nchunks) ? NSLOT : 0);
                    wait_all_but(younger);
                    __builtin_amdgcn_s_barrier();
                }
            }
            if (c + 1 < a.nchunks) {
#pragma unroll
                for (int pb = 0; pb < PXB; ++pb) load_x(smem + ((c + 1) & 1) * buf_bytes, 0, pb);
            }
            if (c + 2 < a.nchunks) load_patch(c + 2);
        };
        // RING4: ONE patch buffer.  Chunk c + 1 sits in registers from the end of chunk c - 1 and is written during tap T - 1 of chunk c.
        auto chunk4 = [&](auto par, int c) {
            constexpr int PAR = decltype(par)::value;
            const int st0 = c * T;
            const bool more_patch = c + 1 < a.nchunks;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int cur = (PAR + t) & 1;
                const int st = st0 + t;
#pragma unroll
                for (int pb = 0; pb < PXB; ++pb) {
                    if (pb == 0 && t == T - 1) {
                        // every wave's fragment reads of this chunk's patch completed before the barrier that closed tap T - 2:
                        // overwrite it with chunk c + 1 (the compiler's wait for those loads must not cover a younger DMA, hence
                        // the DMA of this step after it), then request chunk c + 2
                        if (more_patch) store_patch(0, 0, NSLOT);
                        issue_w(st + 3);
                        if (c + 2 < a.nchunks) load_patch(c + 2);
                    }
                    mma(wf[cur % WSETS], pb);
                    __builtin_amdgcn_sched_barrier(0);
                    if (pb == 0) {       // behind the first MFMAs of the step, see chunk8
                        if (WSETS == 2 && st + 1 < nsteps) read_w(st + 1, wf[(cur ^ 1) % WSETS]);
                        if (t != T - 1) issue_w(st + 3);
                    }
                    if (t + 1 < T) load_x(smem, t + 1, pb);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (WSETS == 1 && st + 1 < nsteps) read_w(st + 1, wf[0]);
                {
                    // closing barrier: the DMA of step st + 2 has landed.  Younger: the DMA(s) of st + 3, and the patch loads of the
                    // next-but-one chunk where they were issued after it -- in tap T - 1 (this step) and, seen from tap 0, in the
                    // step before
                    const int younger = (st + 3 < nsteps ? NDMA : 0) +
                                        (((t == T - 1 && c + 2 < a.nchunks) || (t == 0 && c + 1 < a.nchunks)) ? NSLOT : 0);
                    wait_all_but(younger);
                    // t == T - 2: the reads of the last tap's fragments (issued above) must be complete before anyone overwrites the
                    // patch; t == T - 1: the patch writes must be complete before anyone reads the new patch
                    if (t >= T - 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                }
            }
            if (more_patch) {
#pragma unroll
                for (int pb = 0; pb < PXB; ++pb) load_x(smem, 0, pb);
            }
        };
        PP_TL_MARK(1);
        if constexpr (RING4) {
            int c4 = 0;
            for (; c4 + 1 < a.nchunks; c4 += 2) {
                chunk4(std::integral_constant<int, 0>{}, c4);
                chunk4(std::integral_constant<int, 1>{}, c4 + 1);
            }
            if (c4 < a.nchunks) chunk4(std::integral_constant<int, 0>{}, c4);
        } else {
        int c8 = 0;
        for (; c8 + 1 < a.nchunks; c8 += 2) {
            chunk8(std::integral_constant<int, 0>{}, c8);
            chunk8(std::integral_constant<int, 1>{}, c8 + 1);
        }
        if (c8 < a.nchunks) chunk8(std::integral_constant<int, 0>{}, c8);
        }   // !RING4
    } else {
    // One 16-channel chunk: T steps.  Weights of step s + 1 are requested (global -> the other register set) before the MFMAs of
    // step s; a pixel block's fragments of step s + 1 are read from LDS into the SAME registers as soon as its MFMAs of step s
    // are issued, i.e. while the other pixel block's MFMAs run.  The fences keep the compiler from sinking the loads.
    auto chunk = [&](auto par, int c) {
        constexpr int PAR = decltype(par)::value;
        const unsigned char* pbuf = smem + (c & 1) * buf_bytes;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int cur = (PAR + t) & 1;
#if !(PP_SPLIT_ABLATE & 1)
            load_w(wf[(cur ^ 1) % WSETS]);
#endif
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int pb = 0; pb < PXB; ++pb) {
                // the next chunk's patch (loaded one chunk ago) is split and written half-way through the chunk.  As its own
                // (branched) region: interleaving it with the MFMAs costs ~30 registers (spills with 7-8 patch slots) and
                // measured slower -- the CU's other workgroup keeps the matrix pipe busy meanwhile
#if !(PP_SPLIT_ABLATE & 2)
                if (pb == PXB - 1 && t == T / 2 && c + 1 < a.nchunks) store_patch((c + 1) & 1, 0, NSLOT);
#endif
#if !(PP_SPLIT_ABLATE & 16)
                mma(wf[(PP_SPLIT_ABLATE & 1) ? 0 : cur % WSETS], pb);
#endif
                __builtin_amdgcn_sched_barrier(0);
#if !(PP_SPLIT_ABLATE & 8)
                if (t + 1 < T) load_x(pbuf, t + 1, pb);
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#if !(PP_SPLIT_ABLATE & 4)
        __syncthreads();
#endif
#if !(PP_SPLIT_ABLATE & 8)
        if (c + 1 < a.nchunks) {
#pragma unroll
            for (int pb = 0; pb < PXB; ++pb) load_x(smem + ((c + 1) & 1) * buf_bytes, 0, pb);
        }
#endif
#if !(PP_SPLIT_ABLATE & 2)
        if (c + 2 < a.nchunks) load_patch(c + 2);
#endif
    };
    PP_TL_MARK(1);
    int c = 0;
    for (; c + 1 < a.nchunks; c += 2) {
        chunk(std::integral_constant<int, 0>{}, c);
        chunk(std::integral_constant<int, 1>{}, c + 1);
    }
    if (c < a.nchunks) chunk(std::integral_constant<int, 0>{}, c);
    }   // !WLDS
    PP_TL_MARK(2);

    // ---- epilogue: bias, residuals, ReLU; accumulator register i of a lane = channel 8 (i / 4) + 4 (lane / 32) + i % 4 ---
    int on[PXB], oy[PXB], ox[PXB];
    bool ook[PXB];
    unsigned oam[H ? PXB : 1];                         // fp16 form: maximum of the output pixel's sample (the epilogue's 1 / s)
    {
        int tid_e = tid;
        asm volatile("" : "+v"(tid_e));                // (see pixel_coords: evaluate again instead of carrying the results through the loop)
#pragma unroll
        for (int pb = 0; pb < PXB; ++pb) {
            int aofs_;
            pixel_coords(tid_e & 63, tid_e >> 6, pb, aofs_, on[pb], oy[pb], ox[pb], ook[pb]);
        }
        if constexpr (H) {
#pragma unroll
            for (int pb = 0; pb < PXB; ++pb) oam[pb] = a.x_amax[on[pb]];
        }
    }
#if (PP_SPLIT_ABLATE & 32)
    if (a.relu != 77) {       // keep the accumulators alive with one store per lane
        if (ook[0]) a.y[(((size_t)on[0] * (a.H + a.y_pad) + oy[0]) * (a.W + a.y_pad) + ox[0]) * a.Cout + (lane >> 5)] = acc[0][0][0] + acc[COB - 1][PXB - 1][15];
        return;
    }
#endif
    // all bias / residual loads are issued back to back from clamped (always valid) addresses before anything consumes them: one
    // memory round trip per operand instead of one per (pixel block, channel group) -- the epilogue was 17 % of W48's 48-channel
    // layers (profiles/r02_conv_split_ablation.txt).  Wave tiles of three / four channel blocks (round 5) go block by block: all of
    // them at once would hold 4 x 16 x COB registers beside the accumulators (and the allocator then spills accumulator tuples).
    constexpr int CBB = COB >= 3 ? 1 : COB;            // channel blocks per epilogue batch
    size_t ypix[PXB], r1pix[PXB], r2pix[PXB];
    float xinv[H ? PXB : 1];                           // H: 1 / s of the pixel's sample
#pragma unroll
    for (int pb = 0; pb < PXB; ++pb) {
        if constexpr (H) xinv[pb] = pp_act_unscale(oam[pb]);
        const int n_ = ook[pb] ? on[pb] : 0, y_ = ook[pb] ? oy[pb] : 0, x_ = ook[pb] ? ox[pb] : 0;
        ypix[pb] = ((size_t)n_ * (a.H + a.y_pad) + y_) * (a.W + a.y_pad) + x_;
        r1pix[pb] = ((size_t)n_ * (a.r1_H + a.r1_pad) + (y_ >> a.r1_shift)) * (a.r1_W + a.r1_pad) + (x_ >> a.r1_shift);
        r2pix[pb] = ((size_t)n_ * (a.H + a.r2_pad) + y_) * (a.W + a.r2_pad) + x_;
    }
    float ymax[PXB];
#pragma unroll
    for (int pb = 0; pb < PXB; ++pb) ymax[pb] = 0.f;
#pragma unroll
    for (int cq = 0; cq < COB; cq += CBB) {
    bool cok[CBB][4];
    int cos[CBB][4];
    float4 b4[CBB][4];
    float4 sc4[H ? CBB : 1][H ? 4 : 1];                // H: 1 / (s c) of the lane's channels
#pragma unroll
    for (int cb = 0; cb < CBB; ++cb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int co = (cb0 + cq + cb) * 32 + 8 * g + 4 * (lane >> 5);
            cok[cb][g] = co < a.Cout;
            cos[cb][g] = cok[cb][g] ? co : 0;
            b4[cb][g] = *reinterpret_cast<const float4*>(a.bias + cos[cb][g]);
            if constexpr (H) sc4[cb][g] = *reinterpret_cast<const float4*>(a.wscale + cos[cb][g]);
        }
    float4 rv[PXB][CBB][4];
    if (a.res1) {
#pragma unroll
        for (int pb = 0; pb < PXB; ++pb)
#pragma unroll
            for (int cb = 0; cb < CBB; ++cb)
#pragma unroll
                for (int g = 0; g < 4; ++g) rv[pb][cb][g] = *reinterpret_cast<const float4*>(a.res1 + r1pix[pb] * a.Cout + cos[cb][g]);
    }
#pragma unroll
    for (int pb = 0; pb < PXB; ++pb)
#pragma unroll
        for (int cb = 0; cb < CBB; ++cb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x16 cc = acc[cq + cb][pb];
                const float4 b = b4[cb][g];
                float4 v;
                if constexpr (H) {       // (the products with powers of two are exact: the fma rounds once, like the add)
                    const float xi = xinv[pb];
                    const float4 sc = make_float4(sc4[cb][g].x * xi, sc4[cb][g].y * xi, sc4[cb][g].z * xi, sc4[cb][g].w * xi);
                    v = make_float4(__builtin_fmaf(cc[4 * g + 0], sc.x, b.x), __builtin_fmaf(cc[4 * g + 1], sc.y, b.y),
                                    __builtin_fmaf(cc[4 * g + 2], sc.z, b.z), __builtin_fmaf(cc[4 * g + 3], sc.w, b.w));
                } else {
                    v = make_float4(cc[4 * g + 0] + b.x, cc[4 * g + 1] + b.y, cc[4 * g + 2] + b.z, cc[4 * g + 3] + b.w);
                }
                if (a.relu == PP_RELU_FIRST) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                else if (a.relu >= PP_ACT_LEAKY) { v.x = split_activate(v.x, a.relu); v.y = split_activate(v.y, a.relu); v.z = split_activate(v.z, a.relu); v.w = split_activate(v.w, a.relu); }
                if (a.res1) { v.x += rv[pb][cb][g].x; v.y += rv[pb][cb][g].y; v.z += rv[pb][cb][g].z; v.w += rv[pb][cb][g].w; }
                rv[pb][cb][g] = v;
            }
    PP_TL_MARK(6);
    if (a.res2) {
#pragma unroll
        for (int pb = 0; pb < PXB; ++pb)
#pragma unroll
            for (int cb = 0; cb < CBB; ++cb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 r = *reinterpret_cast<const float4*>(a.res2 + r2pix[pb] * a.Cout + cos[cb][g]);
                    rv[pb][cb][g].x += r.x; rv[pb][cb][g].y += r.y; rv[pb][cb][g].z += r.z; rv[pb][cb][g].w += r.w;
                }
    }
#pragma unroll
    for (int pb = 0; pb < PXB; ++pb) {
#pragma unroll
        for (int cb = 0; cb < CBB; ++cb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 v = rv[pb][cb][g];
                if (a.relu == PP_RELU_LAST) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                if (ook[pb] && cok[cb][g]) {
                    *reinterpret_cast<float4*>(a.y + ypix[pb] * a.Cout + cos[cb][g]) = v;
                    ymax[pb] = fmaxf(ymax[pb], pp_abs4max(v));
                }
            }
    }
    }   // channel-block batches
    if (a.y_amax) {          // the consumer is a fp16-form convolution: max |y| per sample of what was stored (pp_amax.h)
        int wg_first = n, wg_last = n;                 // samples of the workgroup's pixels
        if (a.mode != MODE_TILE) {
            const unsigned last = (unsigned)min((long long)s0 + 32 * PXB * NW - 1, a.S - 1);
            wg_first = (int)pp_udiv(min(s0, last), a.dv_per);
            wg_last = (int)pp_udiv(last, a.dv_per);
        }
        pp_amax_commit_wg<NW, PXB>(a.y_amax, on, ymax, wg_first, wg_last, reinterpret_cast<float*>(smem));
    }
    PP_TL_MARK(3);
#ifdef PP_SPLIT_TIMELINE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PP_TL_MARK(7);
#endif
    PP_TL_FLUSH();
}

// ---- 3x3 layers with 33 .. 48 output channels (HRNet-W48's first branch: 27 % of the program) -----------------------------------
// With 32-row MFMA tiles 48 output channels occupy 3/4 of two channel blocks: a quarter of the matrix work is padding.  This form
// uses v_mfma_f32_16x16x32_bf16 (same rate, 16 x 16 outputs, K = 32): THREE channel blocks of 16 -- no padding -- and K = 32 made
// of TWO TAPS x the chunk's 16 input channels, so the patch layout in LDS and its loader are unchanged: a lane's B fragment is
// pixel (lane & 15) at k half ((lane >> 4) & 1) of tap A (lanes 0 - 31) or tap B (lanes 32 - 63) -- one ds_read_b128 per plane
// at slot pixel + tap offset.  Nine taps make five pairs (the tenth half-step multiplies zero weights): per chunk and 64 pixels
// 3 x 5 x 6 x 4 = 360 MFMAs of 8 passes instead of 2 x 9 x 6 x 2 = 216 of 16: -17 % matrix-pipe time.  A wave covers 64 pixels =
// four 16-pixel sub-blocks (the halves of its two 32-pixel blocks, same lane -> pixel map), 12 accumulators of 4 registers; the
// accumulator rows of a lane are 4 CONSECUTIVE channels of its pixel: the epilogue keeps its float4 stores.
// Weights: [chunk][pair][16-channel block][plane][lane] x 16 B (split_weights48_kernel), read per wave one step ahead.
typedef __attribute__((ext_vector_type(4))) float f32x4;

// WRING (round 5, fp16 form): the split weights through a 4-slot LDS ring filled by the DMA path three pair-steps ahead (as the
// 4-wave ring form of conv_split_kernel), one barrier per pair-step.  Why: with per-wave fragment loads one step ahead a pair-step
// (36 MFMAs of 16 cycles) could not be shorter than an L2 round trip -- the 15 steps of a 48 -> 48 tile took ~15 us for ~4 us of MFMAs.
template <int NSLOT, bool H = false, bool WRING = false>
__global__ __launch_bounds__(256, 2) void conv_split48_kernel(SplitArgs a) {
    constexpr int NT = 256, NPAIR = 5, CB = 3, SB = 4;
    constexpr int XP = H ? 2 : 3;                     // activation planes (H: the fp16 form, see conv_split_kernel)
    constexpr int WPL = H ? 2 : 3;                    // weight planes
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    unsigned L = blockIdx.x, col = blockIdx.y;
    PP_TL_DECL;
    PP_TL_MARK(0);
    if (a.xcd_remap) {
        if (!xcd_tile_column(blockIdx.x, a, L, col)) return;
    }
    const int cbase = (int)col * CB;                  // first 16-channel block of this workgroup (a.ncb blocks in all)
    const int plane_bytes = 2 * a.NPp * 16;
    const int buf_bytes = XP * plane_bytes;
    int n = 0, x0 = 0, y0 = 0;
    unsigned s0 = 0;
    if (a.mode == MODE_TILE) {
        const unsigned L2 = pp_udiv(L, a.dv_tx);
        const int tx = (int)(L - L2 * (unsigned)a.tiles_x);
        n = (int)pp_udiv(L2, a.dv_ty);
        const int ty = (int)(L2 - (unsigned)n * (unsigned)a.tiles_y);
        x0 = tx * a.TW;
        y0 = ty * a.TH;
    } else {
        s0 = L * 256u;
    }
    // ---- patch loader (as conv_split_kernel: the first loads are requested before the rest of the index arithmetic) -------------
    unsigned goff[NSLOT];
    int simg[H ? NSLOT : 1];                          // fp16 form: the sample slot j belongs to (its activation scale)
    const int woff0 = ((((tid & 3) >> 1) * a.NPp + (tid >> 2)) * 16 + (tid & 1) * 8);
    if (a.mode == MODE_STREAM) {
        const int pos0 = (int)s0 - a.PWp - 1 + (tid >> 2);
        const unsigned cin4 = (unsigned)a.Cin * 4u;
        const unsigned base = (unsigned)pos0 * cin4 + (unsigned)(tid & 3) * 16u;
#pragma unroll
        for (int j = 0; j < NSLOT; ++j) goff[j] = pos0 + (NT / 4) * j < 0 ? 0xffffffffu : base + (unsigned)((NT / 4) * j) * cin4;
        if constexpr (H) {
#pragma unroll
            for (int j = 0; j < NSLOT; ++j) {
                const int pj = pos0 + (NT / 4) * j;
                simg[j] = (int)pp_udiv((unsigned)min(max(pj, 0), (int)a.S - 1), a.dv_per);
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < NSLOT; ++j) {
            const int u = tid + NT * j;
            const int p = u >> 2, quad = u & 3;
            const bool in_patch = p < a.NP;
            unsigned off = 0xffffffffu;
            if constexpr (H) simg[j] = n;
            const int pr = (int)pp_udiv((unsigned)p, a.dv_pw), pc = p - pr * a.PWp;
            const int iy = y0 - 1 + pr, ix = x0 - 1 + pc;
            if (in_patch && pc < a.TW + 2 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
                off = (unsigned)(((n * a.xp_h + iy) * a.xp_w + ix) * a.Cin) * 4u;
            goff[j] = off == 0xffffffffu ? off : off + (unsigned)quad * 16u;
        }
    }
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)a.x_bytes, 0x00020000);
    float4 xr[NSLOT];
    auto load_patch = [&](int c) {
#pragma unroll
        for (int j = 0; j < NSLOT; ++j) {
            const unsigned off = goff[j] == 0xffffffffu ? 0xffffffffu : goff[j] + (unsigned)c * 64u;
            xr[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)off, 0, 0));
        }
    };
    const bool last_ok = (tid >> 2) + (NT / 4) * (NSLOT - 1) < a.NP;
    float sx[H ? NSLOT : 1];                          // fp16 form: activation scale of slot j's sample
    auto store_patch = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NSLOT; ++j) {
            uint2 p0, p1, p2;
            if constexpr (H) split4h(xr[j], sx[j], p0, p1);
            else split4(xr[j], p0, p1, p2);
            if (j == NSLOT - 1 && !last_ok) continue;
            unsigned char* d = smem + buf * buf_bytes + woff0 + (NT / 4) * 16 * j;
            *reinterpret_cast<uint2*>(d) = p0;
            *reinterpret_cast<uint2*>(d + plane_bytes) = p1;
            if constexpr (!H) *reinterpret_cast<uint2*>(d + 2 * plane_bytes) = p2;
        }
    };
    // weights: fragment (step, block cb, plane) at ((step * ncb + cb) * 3 + plane) * 64 + lane, step = chunk * 5 + pair
    const uint4* wp = a.w + (size_t)cbase * WPL * 64 + lane;
    const size_t wstep = (size_t)a.ncb * WPL * 64;
    uint4 wf[2][CB][WPL];
    auto load_w = [&](uint4 (&dst)[CB][WPL]) {
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int pl = 0; pl < WPL; ++pl) dst[cb][pl] = wp[(cb * WPL + pl) * 64];
        wp += wstep;
    };
    // ring form: slot (step & 3) of the ring behind the two patch buffers holds the CB x WPL fragments of step = chunk * 5 + pair
    constexpr int WSLOT = CB * WPL * 1024, NDMA = (CB * WPL + 3) / 4;
    const int nsteps = a.nchunks * NPAIR;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const unsigned wring = (unsigned)(2 * buf_bytes);
    auto issue_w = [&](int step) {
        if (step >= nsteps) return;
#pragma unroll
        for (int i = 0; i < NDMA; ++i) {
            const int myslab = (wave + 4 * i) % (CB * WPL);      // (waves past CB * WPL copy a duplicate: same bytes, same place)
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + wring + (unsigned)((step & 3) * WSLOT + myslab * 1024));
            const uint4* g = a.w + (size_t)cbase * WPL * 64 + myslab * 64 + lane + (size_t)step * wstep;
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(dst) : "memory", "m0");
        }
    };
    auto read_w = [&](int step, uint4 (&dst)[CB][WPL]) {
        const unsigned char* base = smem + wring + (step & 3) * WSLOT + lane * 16;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int pl = 0; pl < WPL; ++pl) dst[cb][pl] = *reinterpret_cast<const uint4*>(base + (cb * WPL + pl) * 1024);
    };
    auto wait_all_but = [&](int n) {      // allow `n` (0, NDMA, NSLOT, NSLOT + NDMA) of the youngest vector-memory operations to stay in flight
        if (n == NSLOT + NDMA) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NSLOT + NDMA) : "memory");
        else if (n == NSLOT) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NSLOT) : "memory");
        else if (n == NDMA) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NDMA) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    };
    // ---- the first requests ---------------------------------------------------------------------------------------------------------
    PP_TL_MARK(4);
    if constexpr (WRING) {
        issue_w(0);
        issue_w(1);
        issue_w(2);
        load_patch(0);
    } else {
        load_patch(0);
        load_w(wf[0]);
    }
    unsigned xam[H ? NSLOT : 1];                      // (requested with the first loads, consumed behind the index arithmetic)
    if constexpr (H) {
#pragma unroll
        for (int j = 0; j < NSLOT; ++j) xam[j] = a.x_amax[simg[j]];
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- operand addresses and output pixels: sub-block sb = 2 * (32-pixel block of the wave) + half ------------------------------
    // (the output coordinates are recomputed in the epilogue: 16 registers less across the K loop)
    const int khalf = (lane >> 4) & 1;
    auto sub_block = [&](int sb, int& aof, int& on_, int& oy_, int& ox_, bool& ok) {
        const int b = wave * 2 + (sb >> 1), r = (sb & 1) * 16 + (lane & 15);
        if (a.mode == MODE_TILE) {
            const int BW = 32 >> a.bw_log2, BH = 1 << a.bw_log2;
            int by, bx;
            block_pixel(r, a.bw_log2, by, bx);
            const int gy = b >> a.gx_log2, gx = b & ((1 << a.gx_log2) - 1);
            const int ry = gy * BH + by, rx = gx * BW + bx;
            aof = ((khalf * a.NPp) + ry * a.PWp + rx) * 16;
            on_ = n;
            oy_ = y0 + ry;
            ox_ = x0 + rx;
            ok = oy_ < a.H && ox_ < a.W;
        } else {
            const int pl = b * 32 + r;
            aof = ((khalf * a.NPp) + pl) * 16;
            const unsigned sp = s0 + (unsigned)pl;
            const bool in = sp < (unsigned)a.S;
            const unsigned sc = in ? sp : 0u;
            const unsigned img = pp_udiv(sc, a.dv_per);
            const unsigned rem = sc - img * (unsigned)(a.xp_h * a.PWp);
            const unsigned yy = pp_udiv(rem, a.dv_row);
            on_ = (int)img;
            oy_ = (int)yy;
            ox_ = (int)(rem - yy * (unsigned)a.PWp);
            ok = in && oy_ < a.H && ox_ < a.W;
        }
    };
    int aofs[SB];            // LDS byte offset of this lane's pixel (tap (0,0), plane 0, its k half)
    unsigned oam[H ? SB : 1];                          // fp16 form: maximum of the output pixel's sample (the epilogue's 1 / s)
#pragma unroll
    for (int sb = 0; sb < SB; ++sb) {
        int t0, t1, t2;
        bool t3;
        sub_block(sb, aofs[sb], t0, t1, t2, t3);
        if constexpr (H) oam[sb] = a.x_amax[t0];
    }
    // tap offset of this lane per pair: lanes 0 - 31 read tap 2 * pair, lanes 32 - 63 tap 2 * pair + 1 (pair 4: tap 8 twice, the
    // second against zero weights)
    int tofs[NPAIR];
#pragma unroll
    for (int q = 0; q < NPAIR; ++q) {
        const int t = min(2 * q + (lane >> 5), 8);
        tofs[q] = ((t / 3) * a.PWp + (t % 3)) * 16;
    }
    f32x4 acc[CB][SB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int sb = 0; sb < SB; ++sb) acc[cb][sb] = f32x4{0.f, 0.f, 0.f, 0.f};
    uint4 xf[2][XP];         // the B fragments of sub-block sb live in set sb & 1; the next sub-block's are read during this one's MFMAs
    auto load_x = [&](const unsigned char* pbuf, int q, int sb) {
#pragma unroll
        for (int pl = 0; pl < XP; ++pl) xf[sb & 1][pl] = *reinterpret_cast<const uint4*>(pbuf + pl * plane_bytes + aofs[sb] + tofs[q]);
    };
    auto mma = [&](const uint4 (&wc)[CB][WPL], int sb) {
        if constexpr (H) {       // g1 h0 + g0 h1 + g0 h0
            constexpr int WI[3] = {1, 0, 0}, XI[3] = {0, 1, 0};
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
                    acc[cb][sb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wc[cb][WI[p]]),
                                                                         __builtin_bit_cast(f16x8, xf[sb & 1][XI[p]]), acc[cb][sb], 0, 0, 0);
        } else {
        constexpr int WI[6] = {2, 1, 0, 1, 0, 0}, XI[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
        for (int p = 0; p < 6; ++p)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
                acc[cb][sb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wc[cb][WI[p]]),
                                                                      __builtin_bit_cast(bf16x8, xf[sb & 1][XI[p]]), acc[cb][sb], 0, 0, 0);
        }
    };
    // ---- prologue ---------------------------------------------------------------------------------------------------------------
    if constexpr (H) {
#pragma unroll
        for (int j = 0; j < NSLOT; ++j) sx[j] = pp_act_scale(xam[j]);
    }
    store_patch(0);              // (patch 0 and the first weights were requested above; ring form: waits, in order, for the DMAs as well)
    PP_TL_MARK(5);
    if (a.nchunks > 1) load_patch(1);
    if constexpr (WRING) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        read_w(0, wf[0]);
    } else {
        __syncthreads();
    }
    load_x(smem, 0, 0);
    // ---- K loop: 5 pair-steps per 16-channel chunk (odd: the weight register sets swap roles from chunk to chunk) -------------------
    auto chunk = [&](auto par, int c) {
        constexpr int PAR = decltype(par)::value;
        const unsigned char* pbuf = smem + (c & 1) * buf_bytes;
#pragma unroll
        for (int q = 0; q < NPAIR; ++q) {
            const int cur = (PAR + q) & 1;
            load_w(wf[cur ^ 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int sb = 0; sb < SB; ++sb) {
                if (sb == SB - 1 && q == NPAIR / 2 && c + 1 < a.nchunks) store_patch((c + 1) & 1);
                // the next sub-block's fragments (of this pair, or sub-block 0 of the next pair) are requested before this one's MFMAs
                if (sb + 1 < SB) load_x(pbuf, q, sb + 1);
                else if (q + 1 < NPAIR) load_x(pbuf, q + 1, 0);
                __builtin_amdgcn_sched_barrier(0);
                mma(wf[cur], sb);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
        if (c + 1 < a.nchunks) load_x(smem + ((c + 1) & 1) * buf_bytes, 0, 0);
        if (c + 2 < a.nchunks) load_patch(c + 2);
    };
    // ring form.  Order of the vector-memory operations (vmcnt is in issue order): step st issues the DMA of step st + 3 behind its first
    // sub-block -- in the middle step of a chunk (q = 2) behind the patch store instead (the compiler's wait for the staged patch must
    // not cover a younger DMA), followed by the loads of the patch after next.  The closing barrier of step st needs the DMA of st + 2
    // (issued in step st - 1) landed: younger are the DMA of st + 3 and, in steps q = 2 / 3, those patch loads.
    auto chunk_ring = [&](auto par, int c) {
        constexpr int PAR = decltype(par)::value;
        const unsigned char* pbuf = smem + (c & 1) * buf_bytes;
#pragma unroll
        for (int q = 0; q < NPAIR; ++q) {
            const int cur = (PAR + q) & 1;
            const int st = c * NPAIR + q;
#pragma unroll
            for (int sb = 0; sb < SB; ++sb) {
                if (sb == SB - 1 && q == NPAIR / 2) {
                    if (c + 1 < a.nchunks) store_patch((c + 1) & 1);
                    issue_w(st + 3);
                    if (c + 2 < a.nchunks) load_patch(c + 2);
                }
                if (sb + 1 < SB) load_x(pbuf, q, sb + 1);
                else if (q + 1 < NPAIR) load_x(pbuf, q + 1, 0);
                __builtin_amdgcn_sched_barrier(0);
                mma(wf[cur], sb);
                __builtin_amdgcn_sched_barrier(0);
                if (sb == 0) {
                    if (st + 1 < nsteps) read_w(st + 1, wf[cur ^ 1]);
                    if (q != NPAIR / 2) issue_w(st + 3);
                }
            }
            const int younger = (st + 3 < nsteps ? NDMA : 0) + (((q == NPAIR / 2 || q == NPAIR / 2 + 1) && c + 2 < a.nchunks) ? NSLOT : 0);
            wait_all_but(younger);
            __builtin_amdgcn_s_barrier();
        }
        if (c + 1 < a.nchunks) load_x(smem + ((c + 1) & 1) * buf_bytes, 0, 0);
    };
    PP_TL_MARK(1);
    int c = 0;
    for (; c + 1 < a.nchunks; c += 2) {           // 5 steps per chunk: the register-set parity alternates per chunk
        if constexpr (WRING) {
            chunk_ring(std::integral_constant<int, 0>{}, c);
            chunk_ring(std::integral_constant<int, 1>{}, c + 1);
        } else {
            chunk(std::integral_constant<int, 0>{}, c);
            chunk(std::integral_constant<int, 1>{}, c + 1);
        }
    }
    if (c < a.nchunks) {
        if constexpr (WRING) chunk_ring(std::integral_constant<int, 0>{}, c);
        else chunk(std::integral_constant<int, 0>{}, c);
    }
    PP_TL_MARK(2);
    // ---- epilogue: accumulator register i of a lane = channel 16 cb + 4 (lane >> 4) + i of pixel (lane & 15) -------------------------
    bool cok[CB];
    int cos[CB];
    float4 b4[CB];
    float4 sc4[H ? CB : 1];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
        const int co = (cbase + cb) * 16 + 4 * (lane >> 4);
        cok[cb] = co < a.Cout;
        cos[cb] = cok[cb] ? co : 0;
        b4[cb] = *reinterpret_cast<const float4*>(a.bias + cos[cb]);
        if constexpr (H) sc4[cb] = *reinterpret_cast<const float4*>(a.wscale + cos[cb]);
    }
    size_t ypix[SB], r1pix[SB], r2pix[SB];
    bool ook[SB];
    int onn[SB];
    float xinv[H ? SB : 1];                            // H: 1 / s of the pixel's sample
#pragma unroll
    for (int sb = 0; sb < SB; ++sb) {
        int aof_, on_, oy_, ox_;
        sub_block(sb, aof_, on_, oy_, ox_, ook[sb]);
        onn[sb] = on_;
        if constexpr (H) xinv[sb] = pp_act_unscale(oam[sb]);
        const int n_ = ook[sb] ? on_ : 0, y_ = ook[sb] ? oy_ : 0, x_ = ook[sb] ? ox_ : 0;
        ypix[sb] = ((size_t)n_ * (a.H + a.y_pad) + y_) * (a.W + a.y_pad) + x_;
        r1pix[sb] = ((size_t)n_ * (a.r1_H + a.r1_pad) + (y_ >> a.r1_shift)) * (a.r1_W + a.r1_pad) + (x_ >> a.r1_shift);
        r2pix[sb] = ((size_t)n_ * (a.H + a.r2_pad) + y_) * (a.W + a.r2_pad) + x_;
    }
    float4 rv[SB][CB];
    if (a.res1) {
#pragma unroll
        for (int sb = 0; sb < SB; ++sb)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) rv[sb][cb] = *reinterpret_cast<const float4*>(a.res1 + r1pix[sb] * a.Cout + cos[cb]);
    }
#pragma unroll
    for (int sb = 0; sb < SB; ++sb)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            const f32x4 cc = acc[cb][sb];
            const float4 b = b4[cb];
            float4 v;
            if constexpr (H) v = make_float4(__builtin_fmaf(cc[0], sc4[cb].x * xinv[sb], b.x), __builtin_fmaf(cc[1], sc4[cb].y * xinv[sb], b.y),
                                             __builtin_fmaf(cc[2], sc4[cb].z * xinv[sb], b.z), __builtin_fmaf(cc[3], sc4[cb].w * xinv[sb], b.w));
            else v = make_float4(cc[0] + b.x, cc[1] + b.y, cc[2] + b.z, cc[3] + b.w);
            if (a.relu == PP_RELU_FIRST) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            else if (a.relu >= PP_ACT_LEAKY) { v.x = split_activate(v.x, a.relu); v.y = split_activate(v.y, a.relu); v.z = split_activate(v.z, a.relu); v.w = split_activate(v.w, a.relu); }
            if (a.res1) { v.x += rv[sb][cb].x; v.y += rv[sb][cb].y; v.z += rv[sb][cb].z; v.w += rv[sb][cb].w; }
            rv[sb][cb] = v;
        }
    PP_TL_MARK(6);
    if (a.res2) {
#pragma unroll
        for (int sb = 0; sb < SB; ++sb)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                const float4 r = *reinterpret_cast<const float4*>(a.res2 + r2pix[sb] * a.Cout + cos[cb]);
                rv[sb][cb].x += r.x; rv[sb][cb].y += r.y; rv[sb][cb].z += r.z; rv[sb][cb].w += r.w;
            }
    }
    float ymax[SB];
#pragma unroll
    for (int sb = 0; sb < SB; ++sb) {
        ymax[sb] = 0.f;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            float4 v = rv[sb][cb];
            if (a.relu == PP_RELU_LAST) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            if (ook[sb] && cok[cb]) {
                *reinterpret_cast<float4*>(a.y + ypix[sb] * a.Cout + cos[cb]) = v;
                ymax[sb] = fmaxf(ymax[sb], pp_abs4max(v));
            }
        }
    }
    if (a.y_amax) {          // max |y| per sample of what was stored (pp_amax.h)
        int wg_first = n, wg_last = n;
        if (a.mode != MODE_TILE) {
            const unsigned last = (unsigned)min((long long)s0 + 255, a.S - 1);
            wg_first = (int)pp_udiv(min(s0, last), a.dv_per);
            wg_last = (int)pp_udiv(last, a.dv_per);
        }
        pp_amax_commit_wg<4, SB>(a.y_amax, onn, ymax, wg_first, wg_last, reinterpret_cast<float*>(smem));
    }
    PP_TL_MARK(3);
#ifdef PP_SPLIT_TIMELINE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PP_TL_MARK(7);
#endif
    PP_TL_FLUSH();
}

// fp16 form: the two weight planes of one value under its channel's normalisation c = 2^e (g0 = f16(w c), g1 = f16(w c - g0));
// cmax = max |w| of the output channel (0: an all-zero channel, c = 1)
__device__ __forceinline__ float channel_scale(float cmax) {
    if (!(cmax > 0.f) || !(cmax < 3.0e38f)) return 1.f;
    int e;
    (void)frexpf(cmax, &e);                    // cmax = m 2^e, m in [0.5, 1)  ->  cmax 2^(15 - e) in [2^14, 2^15)
    return ldexpf(1.f, 15 - e);
}
__device__ __forceinline__ void split_weight_h(float v, float c, unsigned short& q0, unsigned short& q1, unsigned short& q2) {
    const float vs = v * c;                    // exact (power of two; |vs| < 2^15)
    const _Float16 g0 = (_Float16)vs;
    const float r = vs - (float)g0;            // exact
    const _Float16 g1 = (_Float16)r;
    q0 = __builtin_bit_cast(unsigned short, g0);
    q1 = __builtin_bit_cast(unsigned short, g1);
    q2 = 0;                                    // (no third plane in this form)
}
// max |w| per output channel of a packed conv weight ([K / 32][CoutPad][32]); one wave per channel; also writes 1 / c
__global__ __launch_bounds__(64) void weight_channel_max_kernel(const float* w, int kchunks, int CoutPad, int nout, float* cmax, float* inv_scale) {
    const int cout = blockIdx.x, lane = threadIdx.x;
    float m = 0.f;
    if (cout < CoutPad)
        for (int c = lane >> 5; c < kchunks; c += 2) m = fmaxf(m, fabsf(w[((size_t)c * CoutPad + cout) * 32 + (lane & 31)]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0 && cout < nout) {
        cmax[cout] = m;
        inv_scale[cout] = 1.f / channel_scale(m);
    }
}

// split weights of the 48-channel form: [chunk][pair][16-channel block][plane][lane] x 16 B; lane = (k group g = lane >> 4: tap
// 2 * pair + (g >> 1), channels 8 (g & 1) .. + 7 of the chunk; channel block row lane & 15); tap 9 (the odd half of pair 4) is zero
template <bool H>
__global__ __launch_bounds__(256) void split_weights48_kernel(const float* w, uint4* out, int Cin, int CoutPad, int ncb16, size_t total, const float* cmax) {
    const size_t i = blockIdx.x * (size_t)256 + threadIdx.x;       // (chunk, pair, cb, lane)
    if (i >= total) return;
    const int lane = (int)(i & 63);
    size_t r = i >> 6;
    const int cb = (int)(r % ncb16);
    r /= ncb16;
    const int pair = (int)(r % 5);
    const int c = (int)(r / 5);
    const int g = lane >> 4;
    const int t = 2 * pair + (g >> 1);
    const int cout = cb * 16 + (lane & 15);
    unsigned short h[3][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int cin = c * 16 + (g & 1) * 8 + j;
        const int k = t * Cin + cin;
        float v = 0.f;
        if (t < 9 && cout < CoutPad) v = w[((size_t)(k >> 5) * CoutPad + cout) * 32 + 8 * (k & 3) + ((k & 31) >> 2)];
        if constexpr (H) {
            split_weight_h(v, channel_scale(cmax[cout]), h[0][j], h[1][j], h[2][j]);
            continue;
        }
        const __bf16 q0 = (__bf16)v;
        const float r1 = v - (float)q0;
        const __bf16 q1 = (__bf16)r1;
        const float r2 = r1 - (float)q1;
        const __bf16 q2 = (__bf16)r2;
        h[0][j] = __builtin_bit_cast(unsigned short, q0);
        h[1][j] = __builtin_bit_cast(unsigned short, q1);
        h[2][j] = __builtin_bit_cast(unsigned short, q2);
    }
    constexpr int WPL = H ? 2 : 3;
    const size_t frag = (i >> 6) * WPL;
#pragma unroll
    for (int pl = 0; pl < WPL; ++pl) {
        uint4 o;
        o.x = h[pl][0] | ((unsigned)h[pl][1] << 16);
        o.y = h[pl][2] | ((unsigned)h[pl][3] << 16);
        o.z = h[pl][4] | ((unsigned)h[pl][5] << 16);
        o.w = h[pl][6] | ((unsigned)h[pl][7] << 16);
        out[(frag + pl) * 64 + lane] = o;
    }
}

// ---- 1x1 / full-cover layers with >= 128 output channels: 8 waves, 256 x 256 (or 512 x 128) output tile --------------------------
// A product Y[M][N] = X[M][K] W[K][N] moves 4 / (2 BN) bytes of X and 6 / (2 BM) bytes of split W per float32 FLOP through the
// vector memory path.  With the tap kernel's 256 x 64 tile that is 0.043 B/FLOP -- 10.7 TB/s at the 250 TFLOP/s the matrix pipes
// could do, and the measured ~115 TFLOP/s of that form on ResNet's 1x1 layers is exactly ~5 TB/s of it.  A 256 x 256 tile needs
// 0.0195 B/FLOP, which takes 128 accumulator registers per lane: 8 waves (wave tile 128 pixels x 64 channels), one workgroup per
// CU, BOTH operands staged through LDS (X split once per workgroup and shared by the waves along N, W copied verbatim in fragment
// order and shared by the waves along M), two LDS stages of 16 input channels, one barrier per stage (3072 matrix-pipe cycles).
// WM x WN = 8 waves: one workgroup per CU; 4 waves (256 x 128 tile): two workgroups per CU, for layers with few 16-channel stages,
// where one workgroup's start-up and epilogue hide under the other's K loop.
template <int WM, int WN, bool H = false>
__global__ __launch_bounds__(64 * WM * WN, 512 / (64 * WM * WN)) void conv_split_gemm_kernel(SplitArgs a) {
    constexpr int XP = H ? 2 : 3;                    // activation planes (H: the fp16 form, see conv_split_kernel)
    constexpr int WPL = H ? 2 : 3;                   // weight planes
    constexpr int NT = 64 * WM * WN, NWAVE = WM * WN;
    constexpr int BM = 128 * WM, BN = 64 * WN;
    constexpr int XS = BM * 4 / NT;                  // float4 patch slots per thread and stage
    constexpr int WU = BN * 2 * WPL;                 // 16-byte units of split weights per stage: BN / 32 blocks x WPL planes x 64 lanes
    constexpr int NPp = BM + 4;
    constexpr int XPLANE = 2 * NPp * 16;
    constexpr int XBYTES = XP * XPLANE, STAGE = XBYTES + WU * 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    unsigned bx = blockIdx.x, by = blockIdx.y;
    PP_TL_DECL;
    PP_TL_MARK(0);
    if (a.xcd_remap) {      // 1-D launch, see xcd_tile_column
        if (!xcd_tile_column(blockIdx.x, a, bx, by)) return;
    }
    const unsigned L = bx, col = by;      // (names the timeline record uses)
    (void)L; (void)col;
    const long long m0 = (long long)bx * BM;
    const int cbB = by * (BN / 32);                  // first channel block of the workgroup
    const int cb0 = cbB + wn * 2;                    // ... of this wave

    // sample of the wave's first / last pixel (clamped to the tensor): equal = the wave's 128 pixels lie in one sample
    const unsigned mfirst = (unsigned)min(m0 + wm * 128, a.S - 1), mlast = (unsigned)min(m0 + wm * 128 + 127, a.S - 1);
    const int img_first = __builtin_amdgcn_readfirstlane((int)pp_udiv(mfirst, a.dv_per)), img_last = __builtin_amdgcn_readfirstlane((int)pp_udiv(mlast, a.dv_per));
    unsigned oam_first = 0;                          // fp16 form: maximum of that sample (the epilogue's 1 / s)
    if constexpr (H) oam_first = a.x_amax[img_first];
    const unsigned lds_base0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const size_t wstep0 = (size_t)a.ncb * (WPL * 64);
    f32x16 acc[2][4];                                // (zeroed right in front of the K loops: 128 live registers less during the set-up)
    if constexpr (H) {
    // ---- fp16 form (round 5): the pixels travel global -> LDS by DMA as well, as raw float32; the two float16 terms are made when
    // a wave reads its B fragment.  Why: with the pixels staged through registers (rounds 2 - 4, still the six-product form below) a
    // stage could not be shorter than the latency of the loads issued ONE stage earlier -- ~1.3 us from HBM under load against
    // 0.38 us of MFMAs in this form (fc6: a lone workgroup advanced one stage per ~2700 cycles; the product layers ran at 0.13 - 0.3
    // of the matrix peak) -- and a second register set for a deeper prefetch does not fit beside 128 accumulator registers (it
    // spilled 256 B per lane and stage).  A DMA needs no registers: a ring of THREE 16 KB pixel stages ([256 pixels][16 channels],
    // rows of 64 B) + three weight stages, everything requested two stages ahead, every wait an exact count (all vector-memory
    // operations of the loop are DMAs: 4 pixel blocks + 2 weight fragments per wave and stage).
    // LDS layout of a pixel stage: pixel-major rows of 64 B, the row's four 16-byte quads XOR-swizzled by (pixel >> 1) & 3 -- applied
    // on the GLOBAL side (DMA lane i of a block writes slot i, i.e. (pixel i / 4, slot i % 4), and fetches quad slot ^ swizzle of that
    // pixel: the four lanes of a pixel still cover its 64 contiguous bytes).  A lane's fragment (pixel lane & 31, k half lane >> 5)
    // is quads 2 half and 2 half + 1 = two ds_read_b128 at the row + ((2 half) ^ swizzle) * 16 and that address ^ 16: every service
    // group of ds_read_b128 covers all 8 bank quads exactly twice (conflict-free).
    // Conversion per fragment (8 floats of one pixel): scale (power of two per sample), h0 = f16(x s), h1 = f16(x s - h0):
    // 24 vector instructions, 96 per wave and stage beside its 24 MFMAs of 32 cycles.
    constexpr int XR = 3, XSTG = BM * 64, WR = 3, WBYTES = WU * 16;
    constexpr int NXD = BM / 16 / NWAVE;             // 1 KB pixel blocks (16 pixels) per wave and stage
    constexpr int NWD = WU / 64 / NWAVE;             // 1 KB weight fragments per wave and stage
    static_assert(BM / 16 % NWAVE == 0 && WU / 64 % NWAVE == 0, "every wave issues the same number of DMAs (vmcnt arithmetic)");
    unsigned gx[NXD];                                // byte offset of this lane's 16 bytes of stage 0 (the tensor is < 4 GiB)
#pragma unroll
    for (int i = 0; i < NXD; ++i) {
        const int pl = (wave + NWAVE * i) * 16 + (lane >> 2);
        const unsigned m = (unsigned)min(m0 + pl, a.S - 1);          // rows past the tensor: any valid address (never stored)
        const int img = (int)pp_udiv(m, a.dv_per), rem = (int)(m - (unsigned)img * (unsigned)(a.H * a.W));
        const int ho = (int)pp_udiv((unsigned)rem, a.dv_row), wo = rem - ho * a.W;
        gx[i] = (unsigned)(((img * a.xp_h + ho * a.stride) * a.xp_w + wo * a.stride) * a.Cin) * 4u + (unsigned)(((lane & 3) ^ ((pl >> 1) & 3)) * 16);
    }
    float xsc[4];                                    // activation scale of the sample of pixel pb * 32 + (lane & 31)
#pragma unroll
    for (int pb = 0; pb < 4; ++pb) xsc[pb] = pp_act_scale(oam_first);
    if (img_first != img_last) {
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) {
            const unsigned mp = (unsigned)min(m0 + wm * 128 + pb * 32 + (lane & 31), a.S - 1);
            xsc[pb] = pp_act_scale(a.x_amax[pp_udiv(mp, a.dv_per)]);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // (the maxima are in: from here on every vector-memory op is a DMA)
    // (scalar base + 32-bit lane offset: one address register per DMA)
    const unsigned wlane16 = (unsigned)lane * 16u;
    auto issue = [&](int c) {                         // weights and pixels of stage c
        const uint4* wsrc = a.w + (size_t)c * wstep0 + (size_t)cbB * (WPL * 64);          // workgroup-uniform
#pragma unroll
        for (int i = 0; i < NWD; ++i) {
            const int slab = i * NWAVE + wave;
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base0 + (unsigned)(XR * XSTG + (c % WR) * WBYTES + slab * 1024));
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(wlane16 + (unsigned)(slab * 1024)), "s"(wsrc), "s"(dst) : "memory", "m0");
        }
        const float* xsrc = a.x + (size_t)c * 16;                                          // 64 bytes per stage
#pragma unroll
        for (int i = 0; i < NXD; ++i) {
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base0 + (unsigned)((c % XR) * XSTG + (wave + NWAVE * i) * 1024));
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(gx[i]), "s"(xsrc), "s"(dst) : "memory", "m0");
        }
    };
    const int fofs = (wm * 128 + (lane & 31)) * 64 + ((((lane >> 5) * 2) ^ ((lane >> 1) & 3)) * 16);     // + pb * 2048; second quad: ^ 16
    const int wofs = XR * XSTG + ((wn * 2) * (WPL * 64) + lane) * 16;                                        // + slot * WBYTES + (cb * WPL + plane) * 1024
    auto frag = [&](const unsigned char* sx_, int pb, uint4 (&h)[2]) {
        const float4 q0 = *reinterpret_cast<const float4*>(sx_ + (fofs + pb * 2048));
        const float4 q1 = *reinterpret_cast<const float4*>(sx_ + ((fofs + pb * 2048) ^ 16));
        const float sc = xsc[pb];
        unsigned a_[4], b_[4];       // (split2h: 16 vector instructions per fragment instead of 24)
        split2h(q0.x, q0.y, sc, a_[0], b_[0]);
        split2h(q0.z, q0.w, sc, a_[1], b_[1]);
        split2h(q1.x, q1.y, sc, a_[2], b_[2]);
        split2h(q1.z, q1.w, sc, a_[3], b_[3]);
        h[0] = make_uint4(a_[0], a_[1], a_[2], a_[3]);
        h[1] = make_uint4(b_[0], b_[1], b_[2], b_[3]);
    };
    PP_TL_MARK(4);
    issue(0);
    if (a.nchunks > 1) issue(1);
    PP_TL_MARK(5);
    if (a.nchunks > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NXD + NWD) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    PP_TL_MARK(1);
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int pb = 0; pb < 4; ++pb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[cb][pb][i] = 0.f;
    for (int c = 0; c < a.nchunks; ++c) {
        const unsigned char* sxp = smem + (c % XR) * XSTG;
        const unsigned char* sw = smem + (c % WR) * WBYTES;
        uint4 wf[2][WPL], xh[2][2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int pl = 0; pl < WPL; ++pl) wf[cb][pl] = *reinterpret_cast<const uint4*>(sw + wofs + (cb * WPL + pl) * 1024);
        frag(sxp, 0, xh[0]);
        // stage c + 2 into the slots stage c - 1 used (every wave is past the barrier that closed it)
        if (c + 2 < a.nchunks) issue(c + 2);
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) {
            __builtin_amdgcn_sched_barrier(0);
            constexpr int WI[3] = {1, 0, 0}, XI[3] = {0, 1, 0};      // g1 h0 + g0 h1 + g0 h0
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
                    acc[cb][pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wf[cb][WI[p]]),
                                                                         __builtin_bit_cast(f16x8, xh[pb & 1][XI[p]]), acc[cb][pb], 0, 0, 0);
            // the next pixel block's fragment is read and converted in the shadow of these MFMAs
            if (pb < 3) frag(sxp, pb + 1, xh[(pb + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // stage c + 1 has landed (younger in the queue: only what this stage requested for c + 2); every LDS read of this stage is done
        if (c + 2 < a.nchunks) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NXD + NWD) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    } else {
    // ---- loaders -------------------------------------------------------------------------------------------------------------
    unsigned goff[XS];
    int simg[H ? XS : 1];                            // fp16 form: the sample slot j belongs to (its activation scale)
#pragma unroll
    for (int j = 0; j < XS; ++j) {
        const int u = tid + NT * j;
        const long long m = m0 + (u >> 2);
        unsigned off = 0xffffffffu;
        if constexpr (H) simg[j] = 0;
        if (m < a.S) {
            const int img = (int)pp_udiv((unsigned)m, a.dv_per), rem = (int)((unsigned)m - (unsigned)img * (unsigned)(a.H * a.W));
            if constexpr (H) simg[j] = img;
            const int ho = (int)pp_udiv((unsigned)rem, a.dv_row), wo = rem - ho * a.W;
            off = (unsigned)(((img * a.xp_h + ho * a.stride) * a.xp_w + wo * a.stride) * a.Cin) * 4u + (unsigned)(u & 3) * 16u;
        }
        goff[j] = off;
    }
    const int woff0 = ((((tid & 3) >> 1) * NPp + (tid >> 2)) * 16 + (tid & 1) * 8);      // + (NT / 4) * 16 j: NT / 4 pixels further
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)a.x_bytes, 0x00020000);
    const size_t wstep = (size_t)a.ncb * (WPL * 64);
    float4 xr[XS];
    auto load_x = [&](int c) {
#pragma unroll
        for (int j = 0; j < XS; ++j) {
            const unsigned off = goff[j] == 0xffffffffu ? 0xffffffffu : goff[j] + (unsigned)c * 64u;
            xr[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)off, 0, 0));
        }
    };
    float sx[H ? XS : 1];                            // fp16 form: activation scale of slot j's sample
    auto store_x = [&](int buf) {
        unsigned char* base = smem + buf * STAGE;
#pragma unroll
        for (int j = 0; j < XS; ++j) {
            uint2 p0, p1, p2;
            if constexpr (H) split4h(xr[j], sx[j], p0, p1);
            else split4(xr[j], p0, p1, p2);
            unsigned char* d = base + woff0 + (NT / 4) * 16 * j;
            *reinterpret_cast<uint2*>(d) = p0;
            *reinterpret_cast<uint2*>(d + XPLANE) = p1;
            if constexpr (!H) *reinterpret_cast<uint2*>(d + 2 * XPLANE) = p2;
        }
    };
    // split weights of a stage: copied verbatim global -> LDS by the DMA path (global_load_lds_dwordx4: no staging registers, no
    // ds_write); wave w moves the 1 KB fragments w, w + 8, ... of the stage's BN / 32 x 3 fragments
    constexpr int NSLAB = WU / 64;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    auto issue_w = [&](int c, int buf) {
        const uint4* src = a.w + (size_t)c * wstep + (size_t)cbB * (WPL * 64) + lane;
#pragma unroll
        for (int i = 0; i < (NSLAB + NWAVE - 1) / NWAVE; ++i) {
            const int slab = i * NWAVE + wave;
            if (NSLAB % NWAVE == 0 || slab < NSLAB)
            {
                // raw instruction: through the builtin the compiler treats the DMA as a possible alias of EVERY later LDS read and
                // inserts s_waitcnt vmcnt(0) in front of the stage's first ds_read (the DMA writes the OTHER stage; completion is
                // awaited explicitly before the barrier below)
                const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(buf * STAGE + XBYTES + slab * 1024));
                const uint4* g = src + slab * 64;
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(dst) : "memory", "m0");
            }
        }
    };

    const int xofs = ((lane >> 5) * NPp + wm * 128 + (lane & 31)) * 16;                 // + plane * XPLANE + pb * 512
    const int wofs = XBYTES + ((wn * 2) * (WPL * 64) + lane) * 16;                      // + (cb * WPL + plane) * 1024


    // vmcnt counts in issue order: a stage issues its weight DMA first and its pixel loads (for the stage after next) after it, so
    // "all but the XS youngest" = the DMA has landed while the pixel loads keep flying across the barrier
    PP_TL_MARK(4);
    issue_w(0, 0);
    load_x(0);
    store_x(0);
    PP_TL_MARK(5);
    if (a.nchunks > 1) load_x(1);
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(XS) : "memory");
    if (a.nchunks <= 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    PP_TL_MARK(1);

#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int pb = 0; pb < 4; ++pb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[cb][pb][i] = 0.f;
    for (int c = 0; c < a.nchunks; ++c) {
        const unsigned char* sb = smem + (c & 1) * STAGE;
        uint4 wf[2][WPL], xf[2][XP];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int pl = 0; pl < WPL; ++pl) wf[cb][pl] = *reinterpret_cast<const uint4*>(sb + wofs + (cb * WPL + pl) * 1024);
#pragma unroll
        for (int pl = 0; pl < XP; ++pl) xf[0][pl] = *reinterpret_cast<const uint4*>(sb + xofs + pl * XPLANE);
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) {
            if (pb < 3) {
#pragma unroll
                for (int pl = 0; pl < XP; ++pl) xf[(pb + 1) & 1][pl] = *reinterpret_cast<const uint4*>(sb + xofs + pl * XPLANE + (pb + 1) * 512);
            }
            __builtin_amdgcn_sched_barrier(0);
            // the next stage's pixels (loaded during the previous stage) are split and written to the other buffer in the shadow
            // of the MFMAs; then its weights are requested (DMA) and the pixels of the stage after it -- in this order: the
            // compiler's vmcnt bookkeeping does not see the raw DMA, so nothing it waits for may be older than an in-flight DMA
            if (pb == 1 && c + 1 < a.nchunks) {
                store_x((c + 1) & 1);
                issue_w(c + 1, (c + 1) & 1);
                if (c + 2 < a.nchunks) load_x(c + 2);
            }
            if constexpr (H) {       // g1 h0 + g0 h1 + g0 h0
                constexpr int WI[3] = {1, 0, 0}, XI[3] = {0, 1, 0};
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb)
                        acc[cb][pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wf[cb][WI[p]]),
                                                                             __builtin_bit_cast(f16x8, xf[pb & 1][XI[p]]), acc[cb][pb], 0, 0, 0);
            } else {
            constexpr int WI[6] = {2, 1, 0, 1, 0, 0}, XI[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
            for (int p = 0; p < 6; ++p)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
                    acc[cb][pb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[cb][WI[p]]),
                                                                          __builtin_bit_cast(bf16x8, xf[pb & 1][XI[p]]), acc[cb][pb], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (c + 2 < a.nchunks)
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(XS) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    }   // !H

    PP_TL_MARK(2);
    PP_TL_MARK(6);
    // ---- epilogue, transposed through LDS (a.epi_lds) ----------------------------------------------------------------------------
    // A lane holds 4 consecutive channels of ONE pixel (lane & 31): stored from the registers, an instruction touches 32 pixels with
    // 32 bytes each.  The stages are free here (every wave passed the loop's last barrier after its last fragment read), so each
    // wave transposes its 128 pixels x 32 channels (one channel block at a time) through 16 KiB of its own ([128][128 B], 16-byte
    // chunk ^ (pixel & 7)) and stores whole 128-byte lines, 8 pixels per instruction; residuals are read in the same lines.  The
    // pixel -> (output, residual) offsets are computed once per pixel (two lanes' worth of divisions per wave, not per store) into
    // a 2 KiB table beside it.  Same arithmetic per element, in the same order, as the register epilogue below: identical bits.
    if (a.epi_lds) {
        unsigned char* stg = smem + wave * (16384 + 2048);
        uint4* tab = reinterpret_cast<uint4*>(stg + 16384);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int p = lane + 64 * i;
            const long long m = m0 + wm * 128 + p;
            uint4 t = make_uint4(0u, 0u, 0u, 0u);
            if (m < a.S) {
                const int img = (int)pp_udiv((unsigned)m, a.dv_per), rem = (int)((unsigned)m - (unsigned)img * (unsigned)(a.H * a.W));
                const int oy = (int)pp_udiv((unsigned)rem, a.dv_row), ox = rem - oy * a.W;
                t.x = (unsigned)(((size_t)img * (a.H + a.y_pad) + oy) * (a.W + a.y_pad) + ox);
                t.y = (unsigned)(((size_t)img * (a.r1_H + a.r1_pad) + (oy >> a.r1_shift)) * (a.r1_W + a.r1_pad) + (ox >> a.r1_shift));
                t.z = (unsigned)(((size_t)img * (a.H + a.r2_pad) + oy) * (a.W + a.r2_pad) + ox);
                t.w = (unsigned)img + 1u;            // 0: no such pixel
            }
            tab[p] = t;
        }
        const int rdrow = lane >> 3, rdpos = lane & 7, rdchunk = rdpos ^ rdrow;
        float xinv[H ? 4 : 1];                       // H: 1 / s of the sample of pixel pb * 32 + (lane & 31)
        if constexpr (H) {
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) xinv[pb] = pp_act_unscale(oam_first);
            if (img_first != img_last) {
#pragma unroll
                for (int pb = 0; pb < 4; ++pb) {
                    const unsigned mp = (unsigned)min(m0 + wm * 128 + pb * 32 + (lane & 31), a.S - 1);
                    xinv[pb] = pp_act_unscale(a.x_amax[pp_udiv(mp, a.dv_per)]);
                }
            }
        }
        float ymax = 0.f, ymax1 = 0.f;               // max |y| of the lane's pixels in img_first / (two samples in the wave) in img_last
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
            for (int pb = 0; pb < 4; ++pb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = (cb0 + cb) * 32 + 8 * g + 4 * (lane >> 5);
                    const float4 b4 = *reinterpret_cast<const float4*>(a.bias + co);
                    const f32x16 cc = acc[cb][pb];
                    float4 v;
                    if constexpr (H) {
                        const float4 sc = *reinterpret_cast<const float4*>(a.wscale + co);
                        const float xi = xinv[pb];
                        v = make_float4(__builtin_fmaf(cc[4 * g + 0], sc.x * xi, b4.x), __builtin_fmaf(cc[4 * g + 1], sc.y * xi, b4.y),
                                        __builtin_fmaf(cc[4 * g + 2], sc.z * xi, b4.z), __builtin_fmaf(cc[4 * g + 3], sc.w * xi, b4.w));
                    } else {
                        v = make_float4(cc[4 * g + 0] + b4.x, cc[4 * g + 1] + b4.y, cc[4 * g + 2] + b4.z, cc[4 * g + 3] + b4.w);
                    }
                    if (a.relu == PP_RELU_FIRST) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    else if (a.relu >= PP_ACT_LEAKY) { v.x = split_activate(v.x, a.relu); v.y = split_activate(v.y, a.relu); v.z = split_activate(v.z, a.relu); v.w = split_activate(v.w, a.relu); }
                    *reinterpret_cast<float4*>(stg + (pb * 32 + (lane & 31)) * 128 + (((2 * g + (lane >> 5)) ^ (lane & 7)) << 4)) = v;
                }
            // the 16 residual lines of this channel block are requested back to back (from clamped, always valid addresses) before
            // any of them is consumed: ONE memory round trip per residual and channel block instead of sixteen in a row -- round
            // 4's timeline put the epilogue of the residual layers (256 -> 1024 at 40x68) at the length of their K loop
            const int co = (cb0 + cb) * 32 + rdchunk * 4;
            const unsigned* tabw = reinterpret_cast<const unsigned*>(tab);
            // two batches of eight lines (round 5): with all sixteen in flight (64 + 64 registers beside the 64 accumulator registers of
            // the other channel block) the allocator spilled accumulator tuples -- and kept them spilled INSIDE the K loop
#pragma unroll
            for (int h8 = 0; h8 < 2; ++h8) {
            float4 vv[8];
#pragma unroll
            for (int i8 = 0; i8 < 8; ++i8) vv[i8] = *reinterpret_cast<const float4*>(stg + ((h8 * 8 + i8) * 8 + rdrow) * 128 + rdpos * 16);
            if (a.res1) {
                float4 rv[8];
#pragma unroll
                for (int i8 = 0; i8 < 8; ++i8) rv[i8] = *reinterpret_cast<const float4*>(a.res1 + (size_t)tabw[((h8 * 8 + i8) * 8 + rdrow) * 4 + 1] * a.Cout + co);
#pragma unroll
                for (int i8 = 0; i8 < 8; ++i8) { vv[i8].x += rv[i8].x; vv[i8].y += rv[i8].y; vv[i8].z += rv[i8].z; vv[i8].w += rv[i8].w; }
            }
            if (a.res2) {
                float4 rv[8];
#pragma unroll
                for (int i8 = 0; i8 < 8; ++i8) rv[i8] = *reinterpret_cast<const float4*>(a.res2 + (size_t)tabw[((h8 * 8 + i8) * 8 + rdrow) * 4 + 2] * a.Cout + co);
#pragma unroll
                for (int i8 = 0; i8 < 8; ++i8) { vv[i8].x += rv[i8].x; vv[i8].y += rv[i8].y; vv[i8].z += rv[i8].z; vv[i8].w += rv[i8].w; }
            }
#pragma unroll
            for (int i8 = 0; i8 < 8; ++i8) {
                const int it = h8 * 8 + i8;
                float4 v = vv[i8];
                if (a.relu == PP_RELU_LAST) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                const uint2 t = *reinterpret_cast<const uint2*>(tabw + (it * 8 + rdrow) * 4);     // .x: output pixel; then (.w via the next read)
                const unsigned tw = tabw[(it * 8 + rdrow) * 4 + 3];
                if (tw) *reinterpret_cast<float4*>(a.y + (size_t)t.x * a.Cout + co) = v;
                if (a.y_amax) {
                    const float m = tw ? pp_abs4max(v) : 0.f;
                    if (img_last - img_first <= 1) {
                        if ((int)tw - 1 == img_first) ymax = fmaxf(ymax, m);
                        else ymax1 = fmaxf(ymax1, m);
                    } else {             // several samples in the wave's pixels (the RoI head: one per pixel): per pixel, its 8 lanes reduced
                        float m8 = fmaxf(m, __shfl_xor(m, 1));
                        m8 = fmaxf(m8, __shfl_xor(m8, 2));
                        m8 = fmaxf(m8, __shfl_xor(m8, 4));
                        if (rdpos == 0 && m8 > 0.f) atomicMax(a.y_amax + (tw - 1u), __float_as_uint(m8));
                    }
                }
            }
            }
        }
        if (a.y_amax) {      // (a wave with more than two samples has issued its atomics above and passes zeros)
            const unsigned wlast = (unsigned)min(m0 + BM - 1, a.S - 1);
            const int im[2] = {img_first, img_last};
            const float ym[2] = {ymax, ymax1};
            pp_amax_commit_wg<NWAVE, 2>(a.y_amax, im, ym, (int)pp_udiv((unsigned)min(m0, (long long)wlast), a.dv_per), (int)pp_udiv(wlast, a.dv_per),
                                        reinterpret_cast<float*>(smem));
        }
        PP_TL_MARK(3);
#ifdef PP_SPLIT_TIMELINE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PP_TL_MARK(7);
#endif
        PP_TL_FLUSH();
        return;
    }

    // ---- epilogue (as conv_split_kernel) -----------------------------------------------------------------------------------------
    int rimg[4];
    float rmax[4];
#pragma unroll
    for (int pb = 0; pb < 4; ++pb) {
        const long long mt = m0 + wm * 128 + pb * 32 + (lane & 31);
        const bool pok = mt < a.S;
        const long long m = pok ? mt : a.S - 1;
        const int img = (int)pp_udiv((unsigned)m, a.dv_per), rem = (int)((unsigned)m - (unsigned)img * (unsigned)(a.H * a.W));
        float xi = 1.f, ymax = 0.f;
        if constexpr (H) xi = pp_act_unscale(a.x_amax[img]);
        rimg[pb] = img;
        const int oy = (int)pp_udiv((unsigned)rem, a.dv_row), ox = rem - oy * a.W;
        const size_t ypix = ((size_t)img * (a.H + a.y_pad) + oy) * (a.W + a.y_pad) + ox;
        const size_t r1pix = ((size_t)img * (a.r1_H + a.r1_pad) + (oy >> a.r1_shift)) * (a.r1_W + a.r1_pad) + (ox >> a.r1_shift);
        const size_t r2pix = ((size_t)img * (a.H + a.r2_pad) + oy) * (a.W + a.r2_pad) + ox;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co = (cb0 + cb) * 32 + 8 * g + 4 * (lane >> 5);
                if (co >= a.Cout || !pok) continue;
                const float4 b4 = *reinterpret_cast<const float4*>(a.bias + co);
                const f32x16 cc = acc[cb][pb];
                float4 v;
                if constexpr (H) {
                    const float4 sc = *reinterpret_cast<const float4*>(a.wscale + co);
                    v = make_float4(__builtin_fmaf(cc[4 * g + 0], sc.x * xi, b4.x), __builtin_fmaf(cc[4 * g + 1], sc.y * xi, b4.y),
                                    __builtin_fmaf(cc[4 * g + 2], sc.z * xi, b4.z), __builtin_fmaf(cc[4 * g + 3], sc.w * xi, b4.w));
                } else {
                    v = make_float4(cc[4 * g + 0] + b4.x, cc[4 * g + 1] + b4.y, cc[4 * g + 2] + b4.z, cc[4 * g + 3] + b4.w);
                }
                if (a.relu == PP_RELU_FIRST) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                else if (a.relu >= PP_ACT_LEAKY) { v.x = split_activate(v.x, a.relu); v.y = split_activate(v.y, a.relu); v.z = split_activate(v.z, a.relu); v.w = split_activate(v.w, a.relu); }
                if (a.res1) {
                    const float4 r = *reinterpret_cast<const float4*>(a.res1 + r1pix * a.Cout + co);
                    v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
                }
                if (a.res2) {
                    const float4 r = *reinterpret_cast<const float4*>(a.res2 + r2pix * a.Cout + co);
                    v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
                }
                if (a.relu == PP_RELU_LAST) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                *reinterpret_cast<float4*>(a.y + ypix * a.Cout + co) = v;
                ymax = fmaxf(ymax, pp_abs4max(v));
            }
        }
        rmax[pb] = ymax;
    }
    if (a.y_amax) {
        const unsigned wlast = (unsigned)min(m0 + BM - 1, a.S - 1);
        pp_amax_commit_wg<NWAVE, 4>(a.y_amax, rimg, rmax, (int)pp_udiv((unsigned)min(m0, (long long)wlast), a.dv_per), (int)pp_udiv(wlast, a.dv_per),
                                    reinterpret_cast<float*>(smem));
    }
}

// ---- weight split: float32 blob rows ([K / 32][CoutPad][32], pack_conv order) -> fragment order, three bf16 planes ---------
template <bool H>
__global__ __launch_bounds__(256) void split_weights_kernel(const float* w, uint4* out, int Cin, int taps, int CoutPad, int ncb,
                                                            size_t total, const float* cmax) {
    const size_t i = blockIdx.x * (size_t)256 + threadIdx.x;       // (chunk, tap, cb, lane)
    if (i >= total) return;
    const int lane = (int)(i & 63);
    size_t r = i >> 6;
    const int cb = (int)(r % ncb);
    r /= ncb;
    const int t = (int)(r % taps);
    const int c = (int)(r / taps);
    const int cout = cb * 32 + (lane & 31);
    unsigned short h[3][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int cin = c * 16 + (lane >> 5) * 8 + j;
        const int k = t * Cin + cin;
        float v = 0.f;
        if (cout < CoutPad) v = w[((size_t)(k >> 5) * CoutPad + cout) * 32 + 8 * (k & 3) + ((k & 31) >> 2)];
        if constexpr (H) {
            split_weight_h(v, channel_scale(cmax[cout]), h[0][j], h[1][j], h[2][j]);
            continue;
        }
        // round-to-nearest-even split (see split4): planes 1 and 2 are signed residuals
        const __bf16 q0 = (__bf16)v;
        const float r1 = v - (float)q0;                                      // exact
        const __bf16 q1 = (__bf16)r1;
        const float r2 = r1 - (float)q1;                                     // exact, <= 8 significant bits
        const __bf16 q2 = (__bf16)r2;                                        // exact
        h[0][j] = __builtin_bit_cast(unsigned short, q0);
        h[1][j] = __builtin_bit_cast(unsigned short, q1);
        h[2][j] = __builtin_bit_cast(unsigned short, q2);
    }
    constexpr int WPL = H ? 2 : 3;
    const size_t frag = (i >> 6) * WPL;
#pragma unroll
    for (int pl = 0; pl < WPL; ++pl) {
        uint4 o;
        o.x = h[pl][0] | ((unsigned)h[pl][1] << 16);
        o.y = h[pl][2] | ((unsigned)h[pl][3] << 16);
        o.z = h[pl][4] | ((unsigned)h[pl][5] << 16);
        o.w = h[pl][6] | ((unsigned)h[pl][7] << 16);
        out[(frag + pl) * 64 + lane] = o;
    }
}

// {m, sh} for pp_udiv (device): n / d for every 32-bit n; d == 1 is flagged by sh = 32
void make_magic(unsigned d, unsigned (&dv)[2]) {
    if (d <= 1) { dv[0] = 0; dv[1] = 32; return; }
    unsigned k = 0;
    while ((1ull << k) < d) ++k;                                   // k = ceil(log2 d), 1 <= k <= 32
    dv[0] = (unsigned)((((1ull << k) - d) << 32) / d + 1);
    dv[1] = k - 1;
}

int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

struct TileGeom {
    int TH, TW, bw_log2, gx_log2, PWp, NP, NPp, tiles_x, tiles_y;
    double eff;
};

// the 256-pixel tile shape (8 blocks of 32 pixels) that wastes the fewest pixels on this map
TileGeom pick_tile(int H, int W, int max_np, int pxb) {
    static const int force = env_int("POSEPIPE_SPLIT_TILE", -1);
    TileGeom best{};
    best.eff = -1.0;
    // (bw_log2, gx_log2): 8x32, 4x64, 16x16, 32x8 pixel tiles (pxb = 2, 4 waves); 16x32, 8x64, 32x16, 64x8 (pxb = 4: 8 waves)
    const int cand[4][2] = {{0, 0}, {0, 1}, {1, 0}, {2, 0}};
    const int nblk = 4 * pxb;
    for (int i = 0; i < 4; ++i) {
        if (force >= 0 && i != force) continue;
        TileGeom g{};
        g.bw_log2 = cand[i][0];
        g.gx_log2 = cand[i][1];
        const int BW = 32 >> g.bw_log2, BH = 1 << g.bw_log2, GX = 1 << g.gx_log2, GY = nblk / GX;
        g.TW = GX * BW;
        g.TH = GY * BH;
        g.PWp = g.TW + 2;
        if (g.bw_log2 == 2)
            while (g.PWp % 8 != 4) ++g.PWp;
        g.NP = (g.TH + 2) * g.PWp;
        g.NPp = g.NP;
        while (g.NPp % 8 != 4) ++g.NPp;       // half planes 64 B apart mod 128: conflict-free ds_write_b64
        g.tiles_x = (W + g.TW - 1) / g.TW;
        g.tiles_y = (H + g.TH - 1) / g.TH;
        if (g.NP > max_np) continue;
        g.eff = (double)H * W / ((double)g.tiles_x * g.tiles_y * 128.0 * pxb) - 1e-4 * g.NP / 256.0;
        if (g.eff > best.eff) best = g;
    }
    return best;
}

// strided-patch form: the 128-pixel output tile (4 blocks of 32 pixels, one per wave) whose de-interleaved patch -- four phase maps of
// (TH + 1) x PWp slots -- serves the map best: pixels wasted at the edges against slots staged per tile
TileGeom pick_tile_s2(int H, int W, int max_np) {
    static const int force = env_int("POSEPIPE_SPLIT_TILE_S2", -1);
    TileGeom best{};
    best.eff = -1.0;
    // (bw_log2, gx_log2): 4x32 (1x32 blocks stacked), 8x16 (2x16 blocks stacked), 16x8 (4x8 blocks stacked), 4x32 as 2x2 blocks of 2x16
    const int cand[4][2] = {{0, 0}, {1, 0}, {2, 0}, {1, 1}};
    for (int i = 0; i < 4; ++i) {
        if (force >= 0 && i != force) continue;
        TileGeom g{};
        g.bw_log2 = cand[i][0];
        g.gx_log2 = cand[i][1];
        const int BW = 32 >> g.bw_log2, BH = 1 << g.bw_log2, GX = 1 << g.gx_log2, GY = 4 / GX;
        g.TW = GX * BW;
        g.TH = GY * BH;
        g.PWp = g.TW + 1;
        if (g.bw_log2 == 2)
            while (g.PWp % 8 != 4) ++g.PWp;       // two rows of a 4x8 block per lane group: 128 B apart mod 256
        g.NP = 4 * (g.TH + 1) * g.PWp;
        g.NPp = g.NP;
        while (g.NPp % 8 != 4) ++g.NPp;
        g.tiles_x = (W + g.TW - 1) / g.TW;
        g.tiles_y = (H + g.TH - 1) / g.TH;
        if (g.NP > max_np) continue;
        g.eff = (double)H * W / ((double)g.tiles_x * g.tiles_y * 128.0) - 0.05 * g.NP / 612.0;
        if (g.eff > best.eff) best = g;
    }
    return best;
}

// ---- the 7x7 / stride 2 / 4 -> 64 stem (ResNet-50 conv1), fp16 form ---------------------------------------------------------------------
// The float32 matrix kernels run this layer at 0.46 of THEIR peak (73 TFLOP/s) while it writes 2.85 GB per 64 frames: as a split product
// it is bound by that write.  K is walked as (dy, dx8, c) with the 7 taps of a row padded to 8 (the eighth has zero weights):
// K = 7 x 8 x 4 = 224 = 14 steps of 16, and the 8 k-values of a lane's fragment are TWO ADJACENT INPUT PIXELS x 4 channels -- with the patch
// held as two float16 planes of 8 B per pixel (h0, h1: split once at staging, as in the tap kernels) that is ONE aligned ds_read_b128
// per plane, 16 B apart between neighbouring output pixels (stride 2): conflict-free without a swizzle.
// Persistent workgroups (one per CU, 8 waves): all split weights (14 x 2 x 2 KB = 56 KB) stay in LDS, the patch is double-buffered
// (tile t + 1 is requested at the start of tile t and written half-way), tile = 16 x 32 outputs x 64 channels, a wave = 2 rows.
// Epilogue through the patch buffer the tile just left: [32 px][32 ch] per (row, channel block), whole 128-byte lines to memory.
// Measured (64 frames of 640 x 1088, same box): 2.84 ms on the float32 matrix kernel -> 0.98 ms; 0.80 ms with the stores compiled out
// (the loop, not the 2.85 GB it writes, is what bounds it: ~47 % matrix-pipe utilisation with two waves per SIMD and three barriers per tile).
struct StemArgs {
    const float* x;
    const uint4* w;
    const float* wscale;
    const float* bias;
    float* y;
    const unsigned* x_amax;
    unsigned* y_amax;
    int N, Hin, Win, xp_h, xp_w, Hout, Wout, y_pad, relu;
    int tiles_x, tiles_y, ntiles;
    unsigned dv_tx[2], dv_ty[2];
    unsigned x_bytes;
};
constexpr int STEM_TH = 16, STEM_TW = 32, STEM_PH = 2 * STEM_TH + 5, STEM_PW = 72, STEM_STEPS = 14;
constexpr int STEM_PLANE = STEM_PH * STEM_PW * 8, STEM_BUF = 2 * STEM_PLANE, STEM_WBYTES = STEM_STEPS * 2 * 2 * 1024;
constexpr int STEM_SLOTS = (STEM_PH * STEM_PW + 511) / 512;
constexpr int STEM_LDS = STEM_WBYTES + 2 * STEM_BUF + 256;

__global__ __launch_bounds__(512, 1) void conv_split_stem7_kernel(StemArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const wl = smem;
    unsigned char* const pbase = smem + STEM_WBYTES;
    float* const red = reinterpret_cast<float*>(smem + STEM_WBYTES + 2 * STEM_BUF);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int px = lane & 31, hk = lane >> 5;

    // weights -> LDS, once
    for (int i = tid; i < STEM_WBYTES / 16; i += 512) reinterpret_cast<uint4*>(wl)[i] = a.w[i];

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)a.x_bytes, 0x00020000);
    float4 xr[STEM_SLOTS];
    int tn = 0, ty0 = 0, tx0 = 0;                     // tile being requested / staged
    auto locate = [&](int t, int& n, int& y0, int& x0) {
        const unsigned q = pp_udiv((unsigned)t, a.dv_tx);
        x0 = (int)((unsigned)t - q * (unsigned)a.tiles_x) * STEM_TW;
        n = (int)pp_udiv(q, a.dv_ty);
        y0 = (int)(q - (unsigned)n * (unsigned)a.tiles_y) * STEM_TH;
    };
    auto load_patch = [&](int t) {
        locate(t, tn, ty0, tx0);
#pragma unroll
        for (int j = 0; j < STEM_SLOTS; ++j) {
            const int p = tid + 512 * j;
            const int pr = p / STEM_PW, pc = p - pr * STEM_PW;
            const int iy = 2 * ty0 - 3 + pr, ix = 2 * tx0 - 3 + pc;
            const bool ok = p < STEM_PH * STEM_PW && (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win;
            const unsigned off = ok ? (unsigned)((tn * a.xp_h + iy) * a.xp_w + ix) * 16u : 0xffffffffu;      // (< 4 GiB per launch: pp_launch_conv cuts larger batches)
            xr[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)off, 0, 0));
        }
    };
    auto store_patch = [&](int buf, float sx) {
#pragma unroll
        for (int j = 0; j < STEM_SLOTS; ++j) {
            const int p = tid + 512 * j;
            uint2 p0, p1;
            split4h(xr[j], sx, p0, p1);
            if (p >= STEM_PH * STEM_PW) continue;
            unsigned char* d = pbase + buf * STEM_BUF + p * 8;
            *reinterpret_cast<uint2*>(d) = p0;
            *reinterpret_cast<uint2*>(d + STEM_PLANE) = p1;
        }
    };

    // lane constants: fragment address of output row 2 wave (pixel block 0), step 0; weight fragments; epilogue roles
    const int xofs = ((4 * wave) * STEM_PW + 2 * px + 2 * hk) * 8;
    const unsigned char* const wlane = wl + lane * 16;
    const int rdrow = lane >> 3, rdc = lane & 7;
    float bias_l[2][4], wsc_l[2][4];       // epilogue read-back role: channels cb * 32 + 4 rdc .. + 3
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const float4 b = *reinterpret_cast<const float4*>(a.bias + cb * 32 + 4 * rdc);
        const float4 c = *reinterpret_cast<const float4*>(a.wscale + cb * 32 + 4 * rdc);
        bias_l[cb][0] = b.x; bias_l[cb][1] = b.y; bias_l[cb][2] = b.z; bias_l[cb][3] = b.w;
        wsc_l[cb][0] = c.x; wsc_l[cb][1] = c.y; wsc_l[cb][2] = c.z; wsc_l[cb][3] = c.w;
    }

    int t = blockIdx.x;
    if (t >= a.ntiles) return;
    load_patch(t);
    {
        const float sx = pp_act_scale(a.x_amax[tn]);
        store_patch(0, sx);
    }
    __syncthreads();
    int it = 0;
    for (; t < a.ntiles; t += gridDim.x, ++it) {
        const int buf = it & 1;
        int n, y0, x0;
        locate(t, n, y0, x0);
        const unsigned am = a.x_amax[n];
        const int tnext = t + (int)gridDim.x;
        const bool more = tnext < a.ntiles;
        unsigned am_next = 0;
        if (more) {
            load_patch(tnext);
            am_next = a.x_amax[tn];
        }
        f32x16 acc[2][2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[cb][pb][i] = 0.f;
        const unsigned char* const pl = pbase + buf * STEM_BUF + xofs;
        // two register sets of fragments: step s + 1 is read from LDS before the MFMAs of step s are issued (the fences keep the
        // compiler from sinking the reads next to their consumers: with two waves per SIMD nothing else covers the LDS latency)
        uint4 wf[2][2][2], xf[2][2][2];
        auto load_frags = [&](int s, int set) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int pn = 0; pn < 2; ++pn) wf[set][cb][pn] = *reinterpret_cast<const uint4*>(wlane + ((s * 2 + cb) * 2 + pn) * 1024);
#pragma unroll
            for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                for (int pn = 0; pn < 2; ++pn)
                    xf[set][pb][pn] = *reinterpret_cast<const uint4*>(pl + pn * STEM_PLANE + ((2 * pb + (s >> 1)) * STEM_PW + (s & 1) * 4) * 8);
        };
        load_frags(0, 0);
#pragma unroll
        for (int s = 0; s < STEM_STEPS; ++s) {
            if (s == STEM_STEPS / 2 && more) store_patch(buf ^ 1, pp_act_scale(am_next));
            if (s + 1 < STEM_STEPS) load_frags(s + 1, (s + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            constexpr int WI[3] = {1, 0, 0}, XI[3] = {0, 1, 0};      // g1 h0 + g0 h1 + g0 h0
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb)
                        acc[cb][pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wf[s & 1][cb][WI[p]]),
                                                                             __builtin_bit_cast(f16x8, xf[s & 1][pb][XI[p]]), acc[cb][pb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();                 // every fragment read of `buf` is done (and the next patch is in place)

        // ---- epilogue: (row, channel block) at a time through this wave's 4 KiB of the buffer the tile just left -------------------
        const float xinv = pp_act_unscale(am);
        unsigned char* const stg = pbase + buf * STEM_BUF + wave * 4096;
        float ymax = 0.f;
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
            const int oy = y0 + 2 * wave + pb;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 v = make_float4(acc[cb][pb][4 * g], acc[cb][pb][4 * g + 1], acc[cb][pb][4 * g + 2], acc[cb][pb][4 * g + 3]);
                    *reinterpret_cast<float4*>(stg + px * 128 + (((2 * g + hk) ^ (px & 7)) << 4)) = v;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int r = r4 * 8 + rdrow, ox = x0 + r;
                    const float4 c = *reinterpret_cast<const float4*>(stg + r * 128 + ((rdc ^ (r & 7)) << 4));
                    float4 v = make_float4(__builtin_fmaf(c.x, wsc_l[cb][0] * xinv, bias_l[cb][0]), __builtin_fmaf(c.y, wsc_l[cb][1] * xinv, bias_l[cb][1]),
                                           __builtin_fmaf(c.z, wsc_l[cb][2] * xinv, bias_l[cb][2]), __builtin_fmaf(c.w, wsc_l[cb][3] * xinv, bias_l[cb][3]));
                    if (a.relu != PP_RELU_NONE) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    if (oy < a.Hout && ox < a.Wout) {
                        *reinterpret_cast<float4*>(a.y + (((size_t)n * (a.Hout + a.y_pad) + oy) * (a.Wout + a.y_pad) + ox) * 64 + cb * 32 + 4 * rdc) = v;
                        ymax = fmaxf(ymax, pp_abs4max(v));
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        }
        if (a.y_amax) {
            const int img[1] = {n};
            const float m[1] = {ymax};
            pp_amax_commit_wg<8, 1>(a.y_amax, img, m, n, n, red);
        }
        __syncthreads();                 // the staging area becomes the patch after next
    }
}

// split weights of the stem: fragment (step s, channel block cb, plane) at ((s * 2 + cb) * 2 + plane) * 64 + lane; lane = (channel
// cb * 32 + (lane & 31), k half lane >> 5); its 8 values: k = 16 s + 8 half + j -> dy = s >> 1, dx = 4 (s & 1) + 2 half + (j >> 2), c = j & 3
__global__ __launch_bounds__(256) void split_weights_stem7_kernel(const float* w, uint4* out, int CoutPad, const float* cmax) {
    const int i = blockIdx.x * 256 + threadIdx.x;      // (s, cb, lane)
    if (i >= STEM_STEPS * 2 * 64) return;
    const int lane = i & 63, cb = (i >> 6) & 1, s = i >> 7;
    const int cout = cb * 32 + (lane & 31);
    unsigned short h[2][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int dy = s >> 1, dx = 4 * (s & 1) + 2 * (lane >> 5) + (j >> 2), c = j & 3;
        float v = 0.f;
        if (dx < 7) {
            const int k = (dy * 7 + dx) * 4 + c;
            v = w[((size_t)(k >> 5) * CoutPad + cout) * 32 + 8 * (k & 3) + ((k & 31) >> 2)];
        }
        unsigned short q2;
        split_weight_h(v, channel_scale(cmax[cout]), h[0][j], h[1][j], q2);
    }
#pragma unroll
    for (int pn = 0; pn < 2; ++pn) {
        uint4 o;
        o.x = h[pn][0] | ((unsigned)h[pn][1] << 16);
        o.y = h[pn][2] | ((unsigned)h[pn][3] << 16);
        o.z = h[pn][4] | ((unsigned)h[pn][5] << 16);
        o.w = h[pn][6] | ((unsigned)h[pn][7] << 16);
        out[(size_t)(((s * 2 + cb) * 2 + pn) * 64 + lane)] = o;
    }
}

}  // namespace

// ResNet-50's stem on conv_split_stem7_kernel (fp16 form only: the six-product form keeps the float32 kernels there)
static bool split_stem7(const ConvArgs& a) {
    static const int on = env_int("POSEPIPE_SPLIT_STEM7", 1);
    const bool f16 = a.split_f16 != 0 || (a.numerics == 0 && pp_conv_split_f16_default());
    return on && f16 && a.KH == 7 && a.KW == 7 && a.stride == 2 && a.pad_h == 3 && a.pad_w == 3 && a.dil_h == 1 && a.dil_w == 1 &&
           a.Cin == 4 && a.Cout == 64 && a.CoutPad >= 64 && a.up_log2 == 0 && !a.out_nchw && !a.res1 && !a.res2 &&
           (a.relu == PP_RELU_NONE || a.relu == PP_RELU_FIRST || a.relu == PP_RELU_LAST) && (a.y_stride == 0 || a.y_stride == a.Cout) &&
           a.y_coff == 0 && a.Hout == (a.Hin - 1) / 2 + 1 && a.Wout == (a.Win - 1) / 2 + 1;
}

// how the split kernel sees the layer: taps (9 / 1) and channels per tap (a full-cover 'valid' conv is a 1x1 over KH*KW*Cin)
// committed: the layer was selected when its split weights were built (pp_net_create_ex / the caller of pp_conv_split_eligible), so
// the launch path derives the SAME form from the shape alone -- no knob is read at launch time (ABI 7: nothing is decided at launch)
// POSEPIPE_SPLIT_S2P=0: the strided-patch form off (A/B runs; read once per process)
static bool split_s2_patch_on() {
    static const int on = env_int("POSEPIPE_SPLIT_S2P", 1);
    return on != 0;
}

static bool split_shape(const ConvArgs& a, int* taps, int* cin, int* mode, bool committed = false) {
    const bool k3 = a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad_h == 1 && a.pad_w == 1 && a.dil_h == 1 && a.dil_w == 1 &&
                    a.Hout == a.Hin && a.Wout == a.Win;
    const bool k1 = a.KH == 1 && a.KW == 1 && a.pad_h == 0 && a.pad_w == 0;
    const bool full = a.KH == a.Hin && a.KW == a.Win && a.pad_h == 0 && a.pad_w == 0 && a.dil_h == 1 && a.dil_w == 1 &&
                      a.x_pad == 0 && a.Hout == 1 && a.Wout == 1 && !k1;
    // 3x3 stride 2 (HRNet's transition / fuse layers, ResNet's strided blocks): the product form with one step per (chunk, tap)
    // Measured (gpurun r3c, per-op profiles): it is bound by the same L2 feed as the one-tap form on 1x1 layers (~140 TFLOP/s): ResNet's
    // 128 / 256 / 512-channel strided 3x3 122 -> 132, 132 -> 139, 135 -> 140 TFLOP/s, HRNet 192 -> 384 98 -> 104; with 48 / 96 input
    // channels or ONE channel block per workgroup (Cout 96) it LOSES to the fp32 kernels (48 -> 96: 95 -> 58), so: from 128 input
    // channels, two channel blocks.  (What these layers want is the patch form with a strided patch: 4x the LDS per tile.)
    static const int s2_on = env_int("POSEPIPE_SPLIT_S2", 1);
    // POSEPIPE_SPLIT_S2_MIN_CIN (read per SELECTION, i.e. at net creation: tests set it to run the form on every shape it supports,
    // whatever the selection rule would pick); a committed layer keeps the form it was created with
    const char* s2_env = committed ? nullptr : getenv("POSEPIPE_SPLIT_S2_MIN_CIN");
    // Round 5 (fp16 form: half the products of the form the rule above was measured on; same-box per-op tables, W48, 128 samples): 96 -> 192
    // 1.26 -> 1.09 ms, 96 -> 96 0.19 -> 0.16, 96 -> 384 0.18 -> 0.15, 256 -> 96 (three blocks) 0.85 -> 0.67, 48 -> 192 0.55 -> 0.44, 48 -> 384
    // 0.10 -> 0.08; it still LOSES on 48 -> 96 (1.34 -> 2.10) and 48 -> 48 (0.80 -> 0.89): from 96 input channels, or from 48 with >= 6 blocks.
    const int s2_ncb = (a.Cout + 31) / 32;
    const bool s2_f16 = a.split_f16 != 0 || (a.numerics == 0 && pp_conv_split_f16_default());
    const bool s2_pick = committed || (s2_env ? a.Cin >= atoi(s2_env)
                                              : s2_f16 ? (a.Cin >= 96 || (a.Cin >= 48 && s2_ncb >= 6)) : (a.Cin >= 128 && (s2_ncb & 1) == 0));
    // Round 6 (fp16 form): the STRIDED-PATCH form (conv_split_kernel<.., S2>) takes every 3x3 / stride-2 / pad-1 layer -- it stages each input
    // element once and has no padding products, so the break-even rules above do not apply to it.  Same weight fragments ([chunk][tap]
    // order) as the tap-gather product, which stays the bf16 form's (and POSEPIPE_SPLIT_S2P=0's) kernel: mode MODE_TILE + stride 2.
    const bool s2_shape = a.KH == 3 && a.KW == 3 && a.stride == 2 && a.pad_h == 1 && a.pad_w == 1 && a.dil_h == 1 && a.dil_w == 1 && !full;
    // Same-box per-op tables (profile_net w48 128 / det 64, S2P on / off): 48 -> 96 at 96x72 1.37 -> 0.91 ms (float32 kernel before), 256 -> 96
    // 0.66 -> 0.39, 64 -> 64 at 192x144 0.55 -> 0.37, 96 -> 192 1.07 -> 0.97, 48 -> 48 0.81 -> 0.76; ResNet-50 128 -> 128 at 160x272 1.07 -> 0.88,
    // 256 -> 256 1.01 -> 0.72, 512 -> 512 1.00 -> 0.93.  It LOSES where a 128-pixel tile does not fit the output map: 12x9 outputs (192 -> 384
    // 0.39 -> 0.43, 96 -> 384 0.15 -> 0.16, 48 -> 384 0.09 -> 0.11) and 48 -> 48 onto 24x18 (0.07 -> 0.08) -- those keep their former kernels.
    // The rule reads the layer's shape only (never the batch, never a knob at launch).
    const bool s2p_map = a.Hout * a.Wout >= 1500 || (a.Hout * a.Wout >= 400 && a.Cout >= 96);
    const bool s2p = split_s2_patch_on() && s2_f16 && s2_shape && s2p_map;
    const bool k3s2 = s2_on && s2_shape && (s2_pick || s2p);
    if (!(k3 || k1 || full || k3s2)) return false;
    *taps = (k3 || k3s2) ? 9 : 1;
    *cin = full ? a.K : a.Cin;
    *mode = (k3 || (k3s2 && s2p)) ? MODE_TILE : MODE_GEMM;
    return true;
}

// product kernel (conv_split_gemm_kernel) for 1x1 / full-cover layers: 3 = 256 x 128 tile, 4 waves, two workgroups per CU; 0 = no.
// Round 4: ONE form.  The per-workgroup timeline (tools/split_timeline.py) showed these layers bound by the CU's memory path, with the
// epilogue as long as the K loop (256 -> 1024 at 40x68: 64k cycles of loop, 57k of epilogue) -- the 8-wave forms (256 x 256, 512 x 128: one
// workgroup per CU) had nothing to run beside it.  With the residual lines of the epilogue requested back to back, the two-per-CU
// form wins on EVERY layer of the detector (per 64 frames, 8-wave form -> this one): 512 -> 2048 1.76 -> 1.48 ms, 512 -> 128 1.74 -> 1.52,
// 512 -> 256 2.00 -> 1.88, 1024 -> 256 2.88 -> 2.76, 2048 -> 512 1.13 -> 0.90, fc6 (K = 12544) 6.41 -> 6.24 (264 TFLOP/s), and
// 128 -> 512 leaves the fp32 kernels for it (4.06 -> 3.74); 64 -> 256 stayed there in round 4 (5.59 vs 5.91: profiles/r04_product_kernel_forms.txt).
// Round 5, after the rewrite of this kernel (pixels by DMA, three stages): 64 -> 256 at 160x272 5.60 -> 4.74 ms per 64 frames (4.08 -> 4.81 TB/s
// algorithmic), HRNet's 64 -> 256 at 96x72 2.34 -> 2.01 ms per 128 samples, same box -- the floor is 64 input channels now.
static int gemm8_cfg(const ConvArgs& a, int mode, int cin) {
    static const int on = env_int("POSEPIPE_SPLIT_GEMM8", 1), min_c4 = env_int("POSEPIPE_SPLIT_GEMM4_MIN_C", 64);
    const bool tap_gather = a.KH * a.KW > 1 && cin == a.Cin;        // 3x3 stride 2: the tap kernel's product form, not this one
    if (!on || mode != MODE_GEMM || tap_gather) return 0;
    return (cin >= min_c4 && a.Cout % 128 == 0) ? 3 : 0;
}

bool pp_conv_split_eligible(const ConvArgs& a) {
    // res1: the output's own shape, or a coarser map read with >> shift (FPN's lat[i-1] += up(lat[i]), fused as a shifted read)
    const bool res1_plain = !a.res1 || (a.res1_off_w == 0 && a.res1_shift >= 0 && a.res1_shift <= 4 &&
                                        ((a.Hout - 1) >> a.res1_shift) < a.res1_H && ((a.Wout - 1) >> a.res1_shift) < a.res1_W &&
                                        (a.res1_shift > 0 || (a.res1_H == a.Hout && a.res1_W == a.Wout)));
    int taps, cin, mode;
    if (split_stem7(a)) return true;
    // 1x1 layers with Cout % 128 == 0: the 8-wave product kernel.  Others: the tap kernel's one-tap form wins from ~1024 input
    // channels only (its 256 x 64 tile re-reads X Cout / 64 times; below, the fp32 kernel's smaller tiles do as well or better).
    static const int gemm_min_cin = env_int("POSEPIPE_SPLIT_GEMM_MIN_CIN", 1024);
    if (!split_shape(a, &taps, &cin, &mode)) return false;
    // Round 5 (fp16 form): ONE 64-channel column reads X once, and then the one-tap form beats the float32 kernel from 256 input
    // channels on large maps -- 256 -> 64 at 160x272 1.88 -> 1.62 ms per 64 frames, at 96x72 0.92 -> 0.83 per 128 samples; 12x9 maps
    // (384 -> 48: 0.05 -> 0.11) and the 16-channel RPN maps lose and stay where they were.
    const bool f16_form = a.split_f16 != 0 || (a.numerics == 0 && pp_conv_split_f16_default());
    const bool one_column = f16_form && cin >= 256 && a.Cout > 32 && a.Cout <= 64 && a.HWout >= 4096;   // (per SAMPLE: the choice must not depend on the batch)
    if (mode == MODE_GEMM && taps == 1 && cin < gemm_min_cin && !one_column && !gemm8_cfg(a, mode, cin)) return false;
    return cin % 16 == 0 && a.Cout % 4 == 0 && a.up_log2 == 0 && !a.out_nchw && res1_plain && a.relu <= PP_ACT_SWISH &&
           (a.y_stride == 0 || a.y_stride == a.Cout) && a.y_coff == 0;
}

static int split_ncb(const ConvArgs& a) { return (a.Cout + 31) / 32; }     // odd: one block per workgroup (COB = 1), else two

// 3x3 stride 1 with 33 .. 48 output channels: conv_split48_kernel (three 16-channel blocks, K = two taps x 16 channels)
static bool split_c48(const ConvArgs& a) {
    static const int on = env_int("POSEPIPE_SPLIT_C48", 1), mult = env_int("POSEPIPE_SPLIT_C48_MULT", 0);
    int taps, cin, mode;
    const bool shape = (a.Cout > 32 && a.Cout <= 48) || (mult == 1 && a.Cout % 48 == 0) || (mult == 2 && a.Cout == 192);
    return on && shape && a.stride == 1 && split_shape(a, &taps, &cin, &mode) && mode == MODE_TILE;
}
static int split_ncb16(const ConvArgs& a) { return (a.Cout + 47) / 48 * 3; }      // 16-channel blocks, whole columns of 3

// bytes of the fragments ...
static size_t split_frag_bytes(const ConvArgs& a, bool committed = false) {
    int taps = 1, cin = a.Cin, mode = 0;
    split_shape(a, &taps, &cin, &mode, committed);      // (committed: at launch the form is the one the split copy was built for)
    const size_t wpl = a.split_f16 ? 2 : 3;              // weight planes: (g0, g1) of the fp16 form, three bf16 planes
    if (split_stem7(a)) return STEM_WBYTES;
    if (split_c48(a)) return ((size_t)(cin / 16) * 5 + 1) * split_ncb16(a) * wpl * 64 * sizeof(uint4);
    return ((size_t)(cin / 16) * taps + 1) * split_ncb(a) * wpl * 64 * sizeof(uint4);      // + one spare step: the kernel fetches one step ahead
}
// ... and, fp16 form, of what follows them: 1 / c per output channel, then max |w| per output channel
static int split_nout(const ConvArgs& a) { return split_c48(a) ? split_ncb16(a) * 16 : split_ncb(a) * 32; }
size_t pp_conv_split_bytes(const ConvArgs& a) {
    return split_frag_bytes(a) + (a.split_f16 ? (size_t)2 * split_nout(a) * sizeof(float) : 0);
}

int pp_conv_split_weights(const ConvArgs& a, void* out, hipStream_t stream) {
    int taps = 1, cin = a.Cin, mode = 0;
    split_shape(a, &taps, &cin, &mode);
    const int ncb = split_ncb(a);
    float* inv_scale = nullptr;
    float* cmax = nullptr;
    if (a.split_f16) {
        const int nout = split_nout(a);
        inv_scale = reinterpret_cast<float*>(static_cast<unsigned char*>(out) + split_frag_bytes(a));
        cmax = inv_scale + nout;
        hipLaunchKernelGGL(weight_channel_max_kernel, dim3((unsigned)nout), dim3(64), 0, stream, a.w, a.Kpad / 32, a.CoutPad, nout, cmax, inv_scale);
    }
    if (split_stem7(a)) {
        hipLaunchKernelGGL(split_weights_stem7_kernel, dim3((STEM_STEPS * 2 * 64 + 255) / 256), dim3(256), 0, stream, a.w, (uint4*)out, a.CoutPad, cmax);
        hipError_t es = hipGetLastError();
        if (es != hipSuccess) {
            pp_set_error("split_weights_stem7 launch failed: %s", hipGetErrorString(es));
            return PP_ERR_HIP;
        }
        return PP_OK;
    }
    if (split_c48(a)) {
        const size_t total48 = (size_t)(cin / 16) * 5 * split_ncb16(a) * 64;
        if (a.split_f16)
            hipLaunchKernelGGL(split_weights48_kernel<true>, dim3((unsigned)((total48 + 255) / 256)), dim3(256), 0, stream, a.w, (uint4*)out, cin,
                               a.CoutPad, split_ncb16(a), total48, cmax);
        else
        hipLaunchKernelGGL(split_weights48_kernel<false>, dim3((unsigned)((total48 + 255) / 256)), dim3(256), 0, stream, a.w, (uint4*)out, cin,
                           a.CoutPad, split_ncb16(a), total48, cmax);
        hipError_t e48 = hipGetLastError();
        if (e48 != hipSuccess) {
            pp_set_error("split_weights48 launch failed: %s", hipGetErrorString(e48));
            return PP_ERR_HIP;
        }
        return PP_OK;
    }
    const size_t total = (size_t)(cin / 16) * taps * ncb * 64;
    if (a.split_f16)
        hipLaunchKernelGGL(split_weights_kernel<true>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a.w, (uint4*)out, cin,
                           taps, a.CoutPad, ncb, total, cmax);
    else
    hipLaunchKernelGGL(split_weights_kernel<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a.w, (uint4*)out, cin,
                       taps, a.CoutPad, ncb, total, cmax);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        pp_set_error("split_weights launch failed: %s", hipGetErrorString(e));
        return PP_ERR_HIP;
    }
    return PP_OK;
}

// ---- stand-alone running maximum (pp_amax.h): tensors whose producer has no fused epilogue (program inputs, pools, ...) -----------
__global__ __launch_bounds__(256) void amax_kernel(const float4* x, size_t quads, unsigned per, unsigned* amax) {
    const unsigned n = blockIdx.x / per, part = blockIdx.x - n * per;      // 1-D grid: the RoI head has 64 000 samples
    const float4* xs = x + (size_t)n * quads;
    float m = 0.f;
    for (size_t i = (size_t)part * 256 + threadIdx.x; i < quads; i += (size_t)per * 256) m = fmaxf(m, pp_abs4max(xs[i]));
    __shared__ float red[16];
    const int im[1] = {(int)n};
    const float mm[1] = {m};
    pp_amax_commit_wg<4, 1>(amax, im, mm, (int)n, (int)n, red);
}

int pp_launch_amax(const float* x, int n, size_t elems, unsigned* amax, hipStream_t stream) {
    if (n <= 0 || elems == 0) return PP_OK;
    if (elems % 4 != 0) {
        pp_set_error("amax: %zu floats per sample (a multiple of 4 is required)", elems);
        return PP_ERR_ARG;
    }
    const size_t quads = elems / 4;
    // ~16 KB per workgroup pass; enough workgroups to fill the chip even for a single large sample
    const unsigned per = (unsigned)std::min<size_t>(std::max<size_t>((quads + 1023) / 1024, 1), std::max<size_t>(2048 / (size_t)n, 1));
    hipLaunchKernelGGL(amax_kernel, dim3(per * (unsigned)n), dim3(256), 0, stream, (const float4*)x, quads, per, amax);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        pp_set_error("amax launch failed: %s", hipGetErrorString(e));
        return PP_ERR_HIP;
    }
    return PP_OK;
}

#ifdef PP_SPLIT_TIMELINE
// diagnosis builds: one record buffer, sized for the largest grid; tl_begin hands it to the launch, tl_end synchronises and appends
// "launch header + records" to the file named by $POSEPIPE_SPLIT_TIMELINE (only layers with Cin == $POSEPIPE_SPLIT_TIMELINE_CIN if set)
static unsigned long long* tl_buf = nullptr;
static const size_t TL_MAX_WG = 1 << 19;
static unsigned long long* tl_begin(const ConvArgs& a, hipStream_t stream) {
    static const char* path = getenv("POSEPIPE_SPLIT_TIMELINE");
    static const int only_cin = env_int("POSEPIPE_SPLIT_TIMELINE_CIN", 0);
    static const int max_launches = env_int("POSEPIPE_SPLIT_TIMELINE_MAX", 300);
    static int recorded = 0;
    if (!path || (only_cin && a.Cin != only_cin) || recorded >= max_launches) return nullptr;
    ++recorded;
    if (!tl_buf && hipMalloc(&tl_buf, TL_MAX_WG * 128) != hipSuccess) return nullptr;
    (void)hipMemsetAsync(tl_buf, 0, TL_MAX_WG * 128, stream);
    return tl_buf;
}
static void tl_end(const ConvArgs& a, const SplitArgs& s, const char* kernel, unsigned nwg, hipStream_t stream) {
    if (!s.dbg) return;
    (void)hipStreamSynchronize(stream);
    nwg = std::min<unsigned>(nwg, TL_MAX_WG);
    std::vector<unsigned long long> h((size_t)nwg * 16);
    (void)hipMemcpy(h.data(), tl_buf, h.size() * 8, hipMemcpyDeviceToHost);
    static int launch = 0;
    FILE* f = fopen(getenv("POSEPIPE_SPLIT_TIMELINE"), "ab");
    if (!f) return;
    long long hdr[16] = {0x54494d454c494e45ll, launch++, (long long)nwg, a.N, a.Hout, a.Wout, a.Cin, a.Cout, s.mode, s.nchunks, s.gx, s.gy,
                         a.res1 != nullptr, a.res2 != nullptr, 0, 0};
    memcpy(&hdr[14], kernel, std::min<size_t>(strlen(kernel), 15));
    fwrite(hdr, 8, 16, f);
    fwrite(h.data(), 8, h.size(), f);
    fclose(f);
}
#define PP_TL_BEGIN() s.dbg = tl_begin(a, stream)
#define PP_TL_END(name, nwg) tl_end(a, s, name, nwg, stream)
#else
#define PP_TL_BEGIN()
#define PP_TL_END(name, nwg)
#endif

// the reciprocals the kernels divide by (pp_udiv), from the launch's final geometry
static void fill_divisors(SplitArgs& s) {
    const bool stream_mode = s.mode == MODE_STREAM;
    make_magic(stream_mode ? (unsigned)(s.xp_h * s.PWp) : (unsigned)(s.H * s.W), s.dv_per);
    make_magic(stream_mode ? (unsigned)s.PWp : (unsigned)s.W, s.dv_row);
    make_magic((unsigned)std::max(s.tiles_x, 1), s.dv_tx);
    make_magic((unsigned)std::max(s.tiles_y, 1), s.dv_ty);
    make_magic((unsigned)std::max(s.PWp, 1), s.dv_pw);
    make_magic((unsigned)std::max(s.gy, 1), s.dv_ncol);
    make_magic((unsigned)std::max(s.gx >> 3, 1), s.dv_run0);
    make_magic((unsigned)((s.gx >> 3) + 1), s.dv_run1);
    make_magic((unsigned)std::max(s.ktaps, 1), s.dv_ktaps);
    make_magic((unsigned)std::max(s.KW, 1), s.dv_kw);
}

static int launch_stem7(const ConvArgs& a, hipStream_t stream) {
    if (!a.x_amax) {
        pp_set_error("conv_split: the fp16 form needs the per-sample maximum of its input (ConvArgs::x_amax)");
        return PP_ERR_STATE;
    }
    StemArgs s{};
    s.x = a.x; s.w = (const uint4*)a.wsplit; s.bias = a.bias; s.y = a.y;
    s.wscale = reinterpret_cast<const float*>(static_cast<const unsigned char*>(a.wsplit) + STEM_WBYTES);
    s.x_amax = a.x_amax; s.y_amax = a.y_amax;
    s.N = a.N; s.Hin = a.Hin; s.Win = a.Win; s.xp_h = a.Hin + a.x_pad; s.xp_w = a.Win + a.x_pad;
    s.Hout = a.Hout; s.Wout = a.Wout; s.y_pad = a.y_pad; s.relu = a.relu;
    s.tiles_x = (a.Wout + STEM_TW - 1) / STEM_TW;
    s.tiles_y = (a.Hout + STEM_TH - 1) / STEM_TH;
    s.ntiles = a.N * s.tiles_x * s.tiles_y;
    s.x_bytes = a.x_bytes;
    make_magic((unsigned)s.tiles_x, s.dv_tx);
    make_magic((unsigned)s.tiles_y, s.dv_ty);
    static const int ncu = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
    }();
    static PpPerDeviceOnce once;
    once.run([] { (void)hipFuncSetAttribute((const void*)conv_split_stem7_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, STEM_LDS); });
    const unsigned grid = (unsigned)std::min(s.ntiles, ncu);
    hipLaunchKernelGGL(conv_split_stem7_kernel, dim3(grid), dim3(512), STEM_LDS, stream, s);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        pp_set_error("conv_split_stem7 launch failed: %s", hipGetErrorString(e));
        return PP_ERR_HIP;
    }
    return PP_OK;
}

int pp_launch_conv_split(const ConvArgs& a, hipStream_t stream) {
    int taps = 1, cin = a.Cin, mode = 0;
    if (a.wsplit && split_stem7(a)) return launch_stem7(a, stream);
    if (!a.wsplit || !split_shape(a, &taps, &cin, &mode, true)) {
        pp_set_error("conv_split: layer not eligible");
        return PP_ERR_ARG;
    }
    const bool full = mode == MODE_GEMM && cin != a.Cin;
    SplitArgs s{};
    s.x = a.x; s.w = (const uint4*)a.wsplit; s.bias = a.bias; s.res1 = a.res1; s.res2 = a.res2; s.y = a.y;
    s.N = a.N; s.H = a.Hout; s.W = a.Wout; s.Cin = cin; s.Cout = a.Cout;
    s.ncb = split_ncb(a);
    s.xp_h = full ? 1 : a.Hin + a.x_pad;
    s.xp_w = full ? 1 : a.Win + a.x_pad;
    s.stride = a.stride;
    s.y_pad = a.y_pad; s.r1_pad = a.r1_pad; s.r2_pad = a.r2_pad; s.relu = a.relu;
    s.r1_shift = a.res1 ? a.res1_shift : 0;
    s.r1_H = a.res1 ? a.res1_H : a.Hout;
    s.r1_W = a.res1 ? a.res1_W : a.Wout;
    s.nchunks = cin / 16;
    s.ktaps = 1; s.KW = a.KW; s.pad = a.pad_h; s.Hin = a.Hin; s.Win = a.Win;
    if (mode == MODE_GEMM && taps > 1) {     // tap-gather product: a step per (chunk, tap), weights in the same [chunk][tap] order
        s.ktaps = taps;
        s.nchunks = (cin / 16) * taps;
    }
    s.x_bytes = a.x_bytes;
    s.xcd_remap = a.xcd_remap;
    const bool f16 = a.split_f16 != 0;
    const int xp = f16 ? 2 : 3;              // activation planes in LDS
    const int wpl = f16 ? 2 : 3;             // weight planes per fragment set
    s.wscale = f16 ? reinterpret_cast<const float*>(static_cast<const unsigned char*>(a.wsplit) + split_frag_bytes(a, true)) : nullptr;
    s.x_amax = a.x_amax;
    s.y_amax = a.y_amax;
    if (f16 && !a.x_amax) {
        pp_set_error("conv_split: the fp16 form needs the per-sample maximum of its input (ConvArgs::x_amax)");
        return PP_ERR_STATE;
    }
    PP_TL_BEGIN();
    // inputs beyond the Infinity Cache (256 MB): the columns of a tile back to back; smaller ones: column by column
    static const int colmaj_mb = env_int("POSEPIPE_SPLIT_COLMAJOR_MB", 256);
    s.col_major = (size_t)a.x_bytes <= (size_t)colmaj_mb << 20;
    if (const int g8 = gemm8_cfg(a, mode, cin)) {
        s.mode = MODE_GEMM;
        s.S = (long long)a.M;
        s.col_major = 0;                     // the product kernel walks tile by tile, all columns of a tile back to back
        const int BM = 256, BN = 128;
        dim3 grid((unsigned)((a.M + BM - 1) / BM), (unsigned)(a.Cout / BN));
        s.gx = (int)grid.x; s.gy = (int)grid.y;
        static const int gemm_remap = env_int("POSEPIPE_SPLIT_GEMM_REMAP", 1);
        s.xcd_remap = gemm_remap && s.gy > 1;
        if (s.xcd_remap) grid = dim3((unsigned)((s.gx + 7) / 8 * 8 * s.gy), 1);
        else s.xcd_remap = 0;
        // 1 = epilogue transposed through LDS (18 KiB per wave), 0 = from registers: two store orders of the SAME values (bit-identical,
        // tests/test_gpu_split.py A/Bs them in one process through pp_debug_knob("split_gemm_epilogue", ..)); POSEPIPE_SPLIT_GEMM_EPI is
        // read ONCE per process -- no environment scan per launch
        static const int epi_env = [] { const char* e = getenv("POSEPIPE_SPLIT_GEMM_EPI"); return e ? atoi(e) : 1; }();
        const int epi_knob = g_split_gemm_epi.load(std::memory_order_relaxed);
        s.epi_lds = epi_knob >= 0 ? epi_knob : epi_env;
        const int nwave = 4;
        fill_divisors(s);
        // fp16 form: three raw float32 pixel stages of BM x 64 B + three weight stages; six-product form: two stages of split planes + weights
        const size_t lds_loop = f16 ? (size_t)3 * BM * 64 + (size_t)3 * BN * 2 * wpl * 16 : (size_t)2 * (xp * 2 * (BM + 4) * 16 + BN * 2 * wpl * 16);
        const size_t lds = std::max<size_t>(lds_loop, s.epi_lds ? (size_t)nwave * (16384 + 2048) : 0);
        static PpPerDeviceOnce once;
        once.run([] {
            (void)hipFuncSetAttribute((const void*)conv_split_gemm_kernel<2, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
            (void)hipFuncSetAttribute((const void*)conv_split_gemm_kernel<2, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        });
        if (f16) hipLaunchKernelGGL((conv_split_gemm_kernel<2, 2, true>), grid, dim3(256), lds, stream, s);
        else hipLaunchKernelGGL((conv_split_gemm_kernel<2, 2, false>), grid, dim3(256), lds, stream, s);
        PP_TL_END("g256x128", grid.x * grid.y);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) {
            pp_set_error("conv_split_gemm launch failed: %s", hipGetErrorString(e));
            return PP_ERR_HIP;
        }
        return PP_OK;
    }
    // channel blocks (32 output channels) per wave = per workgroup column.  fp16 form, 4-wave ring kernels: THREE where the block
    // count allows and enough workgroups remain (96 / 192 channels of HRNet-W48): 10 fragment reads per 18 MFMAs and wave instead
    // of 6 per 6 (96 channels ran with ONE block per workgroup: three columns, each staging the patch again) or 8 per 12.
    // Same-box A/B (profile_net w48 128): 96 -> 96 at 48x36 292 -> 315 TFLOP/s, 192 -> 192 at 24x18 328 -> 349; 384 -> 384 at 12x9
    // LOSES (296 -> 227: 308 workgroups on 512 slots) -- hence the workgroup floor below.  FOUR blocks (128, 256, 512 channels: the
    // detector's 3x3 layers; 12 reads per 24 MFMAs, half the patch staging per output): 128 accumulator registers leave room for ONE
    // weight register set (WSETS; with two the K loop spilled: 442 -> 215 TFLOP/s) -- 256 -> 256 at 160x272 429 -> 455 TFLOP/s,
    // 80x136 409 -> 426, 40x68 386 -> 394.
    if (mode == MODE_TILE && a.stride == 2) {
        // strided-patch form: 128-pixel tiles, one pixel block per wave, 1 .. 4 channel blocks per wave (the widest the block count and
        // the LDS budget allow: every column stages the patch again)
        static const int cob2_env = env_int("POSEPIPE_SPLIT_S2_COB", 0);
        const TileGeom g = pick_tile_s2(a.Hout, a.Wout, 13 * 64);
        if (g.eff < 0) {
            pp_set_error("conv_split: no strided-patch tile for a %dx%d map", a.Hout, a.Wout);
            return PP_ERR_STATE;
        }
        s.mode = MODE_TILE;
        s.tiles_x = g.tiles_x; s.tiles_y = g.tiles_y; s.TH = g.TH; s.TW = g.TW; s.bw_log2 = g.bw_log2; s.gx_log2 = g.gx_log2;
        s.PWp = g.PWp; s.NP = g.NP; s.NPp = g.NPp;
        s.SP = (g.TH + 1) * g.PWp;
        const unsigned gx2 = (unsigned)(g.tiles_x * g.tiles_y * a.N);
        int cob2 = 1;
        for (int c : {4, 3, 2})
            if (s.ncb % c == 0 && (size_t)2 * 2 * s.NPp * 16 + (size_t)4 * c * 2 * 1024 <= 80 * 1024 && (cob2_env == 0 || cob2_env == c)) {
                cob2 = c;
                break;
            }
        const int nslot2 = (s.NP + 63) / 64;
        dim3 grid2(gx2, (unsigned)(s.ncb / cob2));
        s.gx = (int)grid2.x; s.gy = (int)grid2.y;
        if (s.xcd_remap) grid2 = dim3((unsigned)((s.gx + 7) / 8 * 8 * s.gy), 1);
        fill_divisors(s);
        make_magic((unsigned)s.SP, s.dv_row);          // (this form decodes a patch slot into (phase, row, column): dv_row = 1 / SP)
        const size_t lds2 = (size_t)2 * 2 * s.NPp * 16 + (size_t)4 * cob2 * 2 * 1024;
#define PP_SPLIT_LAUNCH_S2(NS_, COB_)                                                                                   \
    do {                                                                                                                \
        static PpPerDeviceOnce once;                                                                                    \
        once.run([] {                                                                                                   \
            (void)hipFuncSetAttribute((const void*)conv_split_kernel<9, NS_, COB_, 1, 4, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024); \
        });                                                                                                             \
        hipLaunchKernelGGL((conv_split_kernel<9, NS_, COB_, 1, 4, true, true, true>), grid2, dim3(256), lds2, stream, s); \
    } while (0)
#define PP_SPLIT_LAUNCH_S2_NS(COB_)                                                                                     \
    do {                                                                                                                \
        if (nslot2 <= 10) PP_SPLIT_LAUNCH_S2(10, COB_);                                                                 \
        else if (nslot2 == 11) PP_SPLIT_LAUNCH_S2(11, COB_);                                                            \
        else PP_SPLIT_LAUNCH_S2(13, COB_);                                                                              \
    } while (0)
        if (cob2 == 4) PP_SPLIT_LAUNCH_S2_NS(4);
        else if (cob2 == 3) PP_SPLIT_LAUNCH_S2_NS(3);
        else if (cob2 == 2) PP_SPLIT_LAUNCH_S2_NS(2);
        else PP_SPLIT_LAUNCH_S2_NS(1);
        PP_TL_END("s2p", grid2.x * grid2.y);
        hipError_t e2 = hipGetLastError();
        if (e2 != hipSuccess) {
            pp_set_error("conv_split (strided patch) launch failed: %s", hipGetErrorString(e2));
            return PP_ERR_HIP;
        }
        return PP_OK;
    }
    static const int cob_env = env_int("POSEPIPE_SPLIT_COB", 0);
    const bool cob_wide_ok = a.split_f16 != 0 && mode != MODE_GEMM && !split_c48(a);
    int cob = (s.ncb & 1) ? 1 : 2;
    if (cob_wide_ok && s.ncb % 3 == 0 && (cob_env == 0 || cob_env == 3)) cob = 3;
    else if (cob_wide_ok && s.ncb % 4 == 0 && (cob_env == 0 || cob_env == 4)) cob = 4;
    static const int stream_env = env_int("POSEPIPE_SPLIT_STREAM", -1);
    static const int nw8_min_blocks = env_int("POSEPIPE_SPLIT_NW8_MIN_BLOCKS", 512), nw8_min_chunks = env_int("POSEPIPE_SPLIT_NW8_MIN_CHUNKS", 16);
    unsigned gx = 0;
    const bool c48 = split_c48(a);
    // fp16 form: the 4-wave ring form everywhere.  With half the matrix work per tap both forms sit at the same ~420 TFLOP/s on the
    // long-K layers (LDS bandwidth: 8 b128 fragment reads per 12 MFMAs and wave), and two workgroups per CU hide each other's per-tap
    // barrier: same-box A/B (profile_net det 64) 256 -> 256 at 160x272 414.7 -> 418.9, 80x136 366 -> 400, 40x68 354 -> 371 TFLOP/s
    static const int nw8_f16 = env_int("POSEPIPE_SPLIT_NW8_F16", 0);
    int nw = (c48 || (f16 && !nw8_f16)) ? 4 : 8;
    // geometry for the 8-wave form (512-pixel tiles, one workgroup per CU); if that gives fewer than ~2 workgroups per CU (or
    // the layer is a one-tap product), the 4-wave form (256-pixel tiles, two per CU)
    for (;;) {
        const int tile_px = 64 * nw, pxb = nw / 2;
        if (mode == MODE_GEMM) {
            nw = 4;
            s.mode = MODE_GEMM;
            s.S = (long long)a.M;
            s.NP = 256;
            s.NPp = 260;
            gx = (unsigned)((a.M + 255) / 256);
            break;
        }
        const int max_np = nw == 8 ? 768 : 512;
        const TileGeom g = pick_tile(a.Hout, a.Wout, max_np, pxb);
        const int snp = tile_px + 2 + 2 * s.xp_w;
        const double stream_eff = (double)a.Hout * a.Wout / ((double)s.xp_h * s.xp_w);
        bool use_stream = a.x_pad >= 1 && snp <= max_np && (g.eff < 0 || stream_eff > g.eff + 0.02);
        if (stream_env == 0 && g.eff >= 0) use_stream = false;
        if (stream_env == 1 && a.x_pad >= 1 && snp <= max_np) use_stream = true;
        if (use_stream) {
            s.mode = MODE_STREAM;
            s.PWp = s.xp_w;
            s.S = (long long)a.N * s.xp_h * s.xp_w;
            s.NP = snp;
            s.NPp = snp;
            while (s.NPp % 8 != 4) ++s.NPp;
            gx = (unsigned)((s.S + tile_px - 1) / tile_px);
        } else {
            s.mode = MODE_TILE;
            s.tiles_x = g.tiles_x; s.tiles_y = g.tiles_y; s.TH = g.TH; s.TW = g.TW; s.bw_log2 = g.bw_log2; s.gx_log2 = g.gx_log2;
            s.PWp = g.PWp; s.NP = g.NP; s.NPp = g.NPp;
            gx = (unsigned)(g.tiles_x * g.tiles_y * a.N);
        }
        // measured on real activations (zeros clock ~15 % higher and mislead): 256 -> 256 at 160x272 207 -> 227 TFLOP/s, at 40x68
        // 186 -> 199; but 128 -> 128 189 -> 182 and HRNet's 48 / 96-channel layers -8 %: the per-tap barrier and the three-DMA
        // start-up want >= 16 chunks to pay off
        if (nw == 4 || ((long)gx * (s.ncb / cob) >= nw8_min_blocks && s.nchunks >= nw8_min_chunks)) break;
        nw = 4;
    }
    const int nslot = (s.NP + 16 * nw - 1) / (16 * nw);     // exactly: only the last patch slot of a thread can be partly outside
    // (three / four blocks per wave exist for the 4-wave ring kernels only: back to one / two where that form does not apply)
    static const int ring4_env0 = env_int("POSEPIPE_SPLIT_RING4", 1);
    if (cob >= 3 && !(ring4_env0 && nw == 4 && (size_t)xp * 2 * s.NPp * 16 + (size_t)4 * cob * wpl * 1024 <= 80 * 1024 &&
                      (cob_env >= 3 || (long)gx * (s.ncb / cob) >= 400)))
        cob = (s.ncb & 1) ? 1 : 2;
    dim3 grid(gx, (unsigned)(s.ncb / cob));
    s.gx = (int)grid.x; s.gy = (int)grid.y;
    if (s.xcd_remap) grid = dim3((unsigned)((s.gx + 7) / 8 * 8 * s.gy), 1);
    if (c48) {
        // three 16-channel blocks per workgroup on v_mfma_f32_16x16x32_bf16; Cout / 48 channel columns
        s.ncb = split_ncb16(a);
        s.gy = s.ncb / 3;
        if (s.xcd_remap) grid = dim3((unsigned)((s.gx + 7) / 8 * 8 * s.gy), 1);
        else grid = dim3(gx, (unsigned)s.gy);
        fill_divisors(s);
        // fp16 form: weights through a 4-slot LDS ring (3 blocks x 2 planes = 6 KB per pair-step) where patch + ring leave two workgroups per CU
        static const int ring48_env = env_int("POSEPIPE_SPLIT_RING48", 1);
        const size_t patch48 = (size_t)2 * xp * 2 * s.NPp * 16;
        const bool ring48 = f16 && ring48_env && patch48 + (size_t)4 * 3 * wpl * 1024 <= 80 * 1024;
        const size_t lds48 = patch48 + (ring48 ? (size_t)4 * 3 * wpl * 1024 : 0);
#define PP_SPLIT48_LAUNCH(NS_)                                                                                          \
    do {                                                                                                                \
        static PpPerDeviceOnce once;                                                                                     \
        once.run([] {                                                                                       \
            (void)hipFuncSetAttribute((const void*)conv_split48_kernel<NS_, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024); \
            (void)hipFuncSetAttribute((const void*)conv_split48_kernel<NS_, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024); \
            (void)hipFuncSetAttribute((const void*)conv_split48_kernel<NS_, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024); \
        });                                                                                                             \
        if (ring48) hipLaunchKernelGGL((conv_split48_kernel<NS_, true, true>), grid, dim3(256), lds48, stream, s);      \
        else if (f16) hipLaunchKernelGGL((conv_split48_kernel<NS_, true, false>), grid, dim3(256), lds48, stream, s);   \
        else hipLaunchKernelGGL((conv_split48_kernel<NS_, false, false>), grid, dim3(256), lds48, stream, s);           \
    } while (0)
        if (nslot <= 5) PP_SPLIT48_LAUNCH(5);
        else if (nslot == 6) PP_SPLIT48_LAUNCH(6);
        else if (nslot == 7) PP_SPLIT48_LAUNCH(7);
        else PP_SPLIT48_LAUNCH(8);
        PP_TL_END("c48", grid.x * grid.y);
        hipError_t e48 = hipGetLastError();
        if (e48 != hipSuccess) {
            pp_set_error("conv_split48 launch failed: %s", hipGetErrorString(e48));
            return PP_ERR_HIP;
        }
        return PP_OK;
    }
    // 4 waves, 3x3: weights through the LDS ring with a single-buffered patch (RING4), two workgroups per CU
    static const int ring4_env = env_int("POSEPIPE_SPLIT_RING4", 1);
    const bool ring4 = ring4_env && nw == 4 && mode != MODE_GEMM && (size_t)xp * 2 * s.NPp * 16 + (size_t)4 * cob * wpl * 1024 <= 80 * 1024;
    const size_t lds = (size_t)(ring4 ? 1 : 2) * xp * 2 * s.NPp * 16 + ((nw == 8 || ring4) ? (size_t)4 * cob * wpl * 1024 : 0);
    fill_divisors(s);
#define PP_SPLIT_LAUNCH_H(T_, NS_, NW_, R4_, H_)                                                                        \
    do {                                                                                                                \
        static PpPerDeviceOnce once;                                                                                     \
        once.run([] {     /* > 64 KB of dynamic LDS has to be allowed per kernel */                         \
            (void)hipFuncSetAttribute((const void*)conv_split_kernel<T_, NS_, 2, 2, NW_, R4_, H_>, hipFuncAttributeMaxDynamicSharedMemorySize, (NW_ == 8 ? 160 : 100) * 1024); \
            (void)hipFuncSetAttribute((const void*)conv_split_kernel<T_, NS_, 1, 2, NW_, R4_, H_>, hipFuncAttributeMaxDynamicSharedMemorySize, (NW_ == 8 ? 160 : 100) * 1024); \
        });                                                                                                             \
        if (cob == 2)                                                                                                   \
            hipLaunchKernelGGL((conv_split_kernel<T_, NS_, 2, 2, NW_, R4_, H_>), grid, dim3(64 * NW_), lds, stream, s); \
        else                                                                                                            \
            hipLaunchKernelGGL((conv_split_kernel<T_, NS_, 1, 2, NW_, R4_, H_>), grid, dim3(64 * NW_), lds, stream, s); \
    } while (0)
#define PP_SPLIT_LAUNCH_WIDE(NS_, COB_)                                                                                 \
    do {                                                                                                                \
        static PpPerDeviceOnce once;                                                                                     \
        once.run([] {                                                                                       \
            (void)hipFuncSetAttribute((const void*)conv_split_kernel<9, NS_, COB_, 2, 4, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024); \
        });                                                                                                             \
        hipLaunchKernelGGL((conv_split_kernel<9, NS_, COB_, 2, 4, true, true>), grid, dim3(256), lds, stream, s);       \
    } while (0)
#define PP_SPLIT_LAUNCH(T_, NS_, NW_, R4_)                                                                              \
    do {                                                                                                                \
        if (f16) PP_SPLIT_LAUNCH_H(T_, NS_, NW_, R4_, true);                                                            \
        else PP_SPLIT_LAUNCH_H(T_, NS_, NW_, R4_, false);                                                               \
    } while (0)
    if (mode == MODE_GEMM)
        PP_SPLIT_LAUNCH(1, 4, 4, false);
    else if (nw == 8) {
        if (nslot <= 5) PP_SPLIT_LAUNCH(9, 5, 8, false);
        else PP_SPLIT_LAUNCH(9, 6, 8, false);
    } else if (ring4 && cob == 3) {
        if (nslot <= 5) PP_SPLIT_LAUNCH_WIDE(5, 3);
        else if (nslot == 6) PP_SPLIT_LAUNCH_WIDE(6, 3);
        else if (nslot == 7) PP_SPLIT_LAUNCH_WIDE(7, 3);
        else PP_SPLIT_LAUNCH_WIDE(8, 3);
    } else if (ring4 && cob == 4) {
        if (nslot <= 5) PP_SPLIT_LAUNCH_WIDE(5, 4);
        else if (nslot == 6) PP_SPLIT_LAUNCH_WIDE(6, 4);
        else if (nslot == 7) PP_SPLIT_LAUNCH_WIDE(7, 4);
        else PP_SPLIT_LAUNCH_WIDE(8, 4);
    } else if (ring4) {
        if (nslot <= 5) PP_SPLIT_LAUNCH(9, 5, 4, true);
        else if (nslot == 6) PP_SPLIT_LAUNCH(9, 6, 4, true);
        else if (nslot == 7) PP_SPLIT_LAUNCH(9, 7, 4, true);
        else PP_SPLIT_LAUNCH(9, 8, 4, true);
    } else if (nslot <= 5)
        PP_SPLIT_LAUNCH(9, 5, 4, false);
    else if (nslot == 6)
        PP_SPLIT_LAUNCH(9, 6, 4, false);
    else if (nslot == 7)
        PP_SPLIT_LAUNCH(9, 7, 4, false);
    else
        PP_SPLIT_LAUNCH(9, 8, 4, false);
    PP_TL_END(mode == MODE_GEMM ? "gemm4" : nw == 8 ? "nw8" : ring4 ? "ring4" : "nw4", grid.x * grid.y);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        pp_set_error("conv_split launch failed: %s", hipGetErrorString(e));
        return PP_ERR_HIP;
    }
    return PP_OK;
}
