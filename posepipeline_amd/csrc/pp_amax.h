// Per-sample running maximum of |activation| (device helpers) -- what the fp16 form of the split convolutions scales its input by.
//
// conv_split.hip (round 5): x s = h0 + 2^-11 h1 holds 22 significand bits wherever |x s| is a NORMAL float16, so the scale s has to
// follow the data: s = 2^k PER SAMPLE with max |x| s in [2^14, 2^15).  A per-sample (not per-batch) scale keeps a frame's result
// independent of what else is in the batch (tests/test_gpu_sharded.py compares shards with the whole clip bit for bit).  The maximum
// is tracked where the tensor is PRODUCED: every kernel that stores a tensor a fp16-form convolution reads folds max |v| of what it
// stores into amax[sample] (unsigned bit pattern of a non-negative float: integer order = float order) with one atomic per wave;
// pp_launch_amax is the stand-alone pass for tensors produced elsewhere (program inputs, ops without the fused epilogue).
#pragma once
#include <hip/hip_runtime.h>

// scale 2^k with amax 2^k in [2^14, 2^15) and its inverse, from the bit pattern of amax >= 0 (0, subnormal and non-finite maxima are
// clamped: the exponent field is held to [15, 254], i.e. k in [-113, 126])
__device__ __forceinline__ unsigned pp_amax_exp(unsigned amax_bits) {
    const unsigned e = amax_bits >> 23;
    return e < 15u ? 15u : (e > 254u ? 254u : e);
}
__device__ __forceinline__ float pp_act_scale(unsigned amax_bits) { return __uint_as_float((268u - pp_amax_exp(amax_bits)) << 23); }
__device__ __forceinline__ float pp_act_unscale(unsigned amax_bits) { return __uint_as_float((pp_amax_exp(amax_bits) - 14u) << 23); }

__device__ __forceinline__ float pp_abs4max(const float4 v) { return fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))); }

// fold the lanes' maxima m (>= 0; NaNs were dropped by fmaxf) into amax[img]: ONE atomic per wave when every lane holds the same
// sample (the usual case: a wave's pixels lie in one image), TWO when the wave's pixels straddle an image boundary (two masked
// reductions), one per lane otherwise (maps of a few pixels; the RoI head, where every pixel is its own sample -- distinct
// addresses).  Atomics of one wave on ONE address serialise at the L2 (~16 ns each, measured: the first version, one per lane in
// every straddling wave, cost the detector's 512 -> 2048 layers 8 ms per 64 frames), hence the reductions.
// Lanes without data pass m = 0 and any valid img.  Call in wave-uniform control flow.
__device__ __forceinline__ float pp_wave_max(float m) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    return m;
}
__device__ __forceinline__ void pp_amax_commit(unsigned* amax, int img, float m) {
    const int img0 = __builtin_amdgcn_readfirstlane(img);
    const unsigned long long other = __builtin_amdgcn_ballot_w64(img != img0);
    const bool lane0 = (threadIdx.x & 63) == 0;
    if (other == 0) {
        m = pp_wave_max(m);
        if (lane0 && m > 0.f) atomicMax(amax + img0, __float_as_uint(m));
        return;
    }
    const int img1 = __builtin_amdgcn_readlane(img, __builtin_ctzll(other));
    if (__builtin_amdgcn_ballot_w64(img != img0 && img != img1) == 0) {
        const float m0 = pp_wave_max(img == img0 ? m : 0.f), m1 = pp_wave_max(img == img1 ? m : 0.f);
        if (lane0 && m0 > 0.f) atomicMax(amax + img0, __float_as_uint(m0));
        if (lane0 && m1 > 0.f) atomicMax(amax + img1, __float_as_uint(m1));
        return;
    }
    if (m > 0.f) atomicMax(amax + img, __float_as_uint(m));
}

// Workgroup form -- what the convolution epilogues use.  Device-scope atomics are resolved at the memory side (eight XCDs, eight
// L2s) and those on ONE address serialise at ~0.6 us each (measured: one atomic per wave in PP_OP_UPSAMPLE_ADD, 1300 per sample and
// launch, took the kernel from 75 to 770 us), so a workgroup folds its waves' maxima through LDS first and issues ONE atomic per
// sample it touches.  img_first .. img_last: the samples of the workgroup's pixels (workgroup-uniform); up to four are reduced,
// wider spans (maps of a few pixels, the RoI head) fall back to pp_amax_commit per wave.  img[i] / m[i]: the lane's NPB pixel
// blocks.  lds: NW * 4 floats nobody else touches between the two barriers inside.  Call from ALL threads of the workgroup.
template <int NW, int NPB>
__device__ __forceinline__ void pp_amax_commit_wg(unsigned* amax, const int (&img)[NPB], const float (&m)[NPB], int img_first,
                                                  int img_last, float* lds) {
    const int nimg = img_last - img_first + 1;
    if (nimg > 4) {
#pragma unroll
        for (int i = 0; i < NPB; ++i) pp_amax_commit(amax, img[i], m[i]);
        return;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();                                  // (the K loop's last LDS reads of every wave are done)
    for (int k = 0; k < nimg; ++k) {
        float mk = 0.f;
#pragma unroll
        for (int i = 0; i < NPB; ++i) mk = fmaxf(mk, img[i] == img_first + k ? m[i] : 0.f);
        mk = pp_wave_max(mk);
        if (lane == 0) lds[wave * 4 + k] = mk;
    }
    __syncthreads();
    if ((int)threadIdx.x < nimg) {
        float r = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) r = fmaxf(r, lds[w * 4 + threadIdx.x]);
        if (r > 0.f) atomicMax(amax + img_first + threadIdx.x, __float_as_uint(r));
    }
}

// stand-alone pass: amax[n] = max |x[n][0 .. elems)| for n < N (x: N contiguous samples of `elems` floats, elems % 4 == 0; a zero
// halo does not move a maximum).  The slots must be zero on entry (the kernel only raises them).
int pp_launch_amax(const float* x, int n, size_t elems, unsigned* amax, hipStream_t stream);
// the device array pp_net_input_amax hands out for program input `buf`, WITHOUT promising anything (null: no fp16-form reader)
struct pp_net;
unsigned* pp_net_input_amax_slot(pp_net* net, int buf);
