// Microbenchmark: what does v_mfma_f32_16x16x4_f32 sustain on this part?
//   mode 0: register-only MFMA loop (CT x PT independent accumulators, operands fixed in VGPRs)
//   mode 1: same loop, operands re-read from LDS with ds_read_b128 every 4 steps (the conv kernel's
//           inner loop without its fill, barriers or address math)
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o tools/mfma_peak ; run: tools/mfma_peak
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

//   mode 2: mode 1 + two __syncthreads per 32-k step (the conv kernel's barrier structure, still no fill work)
// THREADS = 256 or 128: how many SIMDs one block's barrier couples
//   mode 3: mode 1 + NV independent full-rate VALU instructions per 32-k step (do VALU and MFMA overlap on a SIMD?)
template <int CT, int PT, int MODE, int THREADS = 256, int NV = 0>
__global__ __launch_bounds__(THREADS) void mfma_loop(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float smem[(16 * CT + 64 * PT) * 32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ __attribute__((aligned(16))) float scratch[8 * THREADS];
    for (int i = tid; i < (16 * CT + 64 * PT) * 32; i += THREADS) smem[i] = (float)(i & 7) * 0.125f;
    __syncthreads();
    f32x4 acc[CT][PT];
    for (int ct = 0; ct < CT; ++ct)
        for (int pt = 0; pt < PT; ++pt) acc[ct][pt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* wrd = smem + (lane & 15) * 32 + (lane >> 4) * 8;
    const float* xrd = smem + 16 * CT * 32 + (wave * 16 * PT + (lane & 15)) * 32 + (lane >> 4) * 8;
    f32x4 av[CT], bv[PT];
    for (int ct = 0; ct < CT; ++ct) av[ct] = *reinterpret_cast<const f32x4*>(wrd + ct * 512);
    for (int pt = 0; pt < PT; ++pt) bv[pt] = *reinterpret_cast<const f32x4*>(xrd + pt * 512);
    unsigned vx[4] = {(unsigned)tid, (unsigned)tid * 3u, (unsigned)tid * 5u, (unsigned)tid * 7u};
    unsigned vy[4] = {1u, 2u, 3u, 4u};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 3) {
#pragma unroll
            for (int v = 0; v < NV / 2; ++v) {
                vx[v & 3] += vy[v & 3];
                vy[v & 3] ^= vx[v & 3];
            }
        }
        if (MODE == 4) {   // NV extra ds_write_b32 per step (conflict-free, private slots): what does an LDS write cost?
#pragma unroll
            for (int v = 0; v < NV; ++v) scratch[(v & 7) * THREADS + tid] = (float)it;
        }
        if (MODE == 5) {   // NV/4 extra ds_write_b128
#pragma unroll
            for (int v = 0; v < NV / 4; ++v)
                *reinterpret_cast<f32x4*>(scratch + ((v & 1) * THREADS + tid) * 4) = f32x4{(float)it, 0.f, 1.f, 2.f};
        }
        if (MODE == 2) {
            __syncthreads();
            if (it == 0x7fffffff) smem[tid] = 1.f;   // never true: keeps the barriers from merging
            __syncthreads();
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (MODE >= 1) {
                // volatile-ish: the offset depends on the loop counter so the reads stay in the loop
                const int so = ((it + h) & 1) * 4;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) av[ct] = *reinterpret_cast<const f32x4*>(wrd + ct * 512 + so);
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) bv[pt] = *reinterpret_cast<const f32x4*>(xrd + pt * 512 + so);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt)
                        acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ct][s], bv[pt][s], acc[ct][pt], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int ct = 0; ct < CT; ++ct)
        for (int pt = 0; pt < PT; ++pt) s += acc[ct][pt][0] + acc[ct][pt][1] + acc[ct][pt][2] + acc[ct][pt][3];
    if (MODE >= 4) s += scratch[(tid * 7) % (8 * THREADS)];
    if (MODE == 3) s += (float)(vx[0] ^ vx[1] ^ vx[2] ^ vx[3] ^ vy[0] ^ vy[1] ^ vy[2] ^ vy[3]);
    if (s == 12345.678f) out[0] = s;   // keep the loop alive
}

template <int CT, int PT, int MODE, int THREADS = 256, int NV = 0>
void run(const char* name, int blocks_per_cu, float* d_out) {
    const int iters = 4000;
    const int blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((mfma_loop<CT, PT, MODE, THREADS, NV>), dim3(blocks), dim3(THREADS), 0, 0, d_out, iters);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((mfma_loop<CT, PT, MODE, THREADS, NV>), dim3(blocks), dim3(THREADS), 0, 0, d_out, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double flops = (double)blocks * (THREADS / 64) * iters * 8.0 * CT * PT * (16 * 16 * 4 * 2);
    printf("%-28s blocks/CU %d  %8.3f ms  %7.2f TFLOP/s\n", name, blocks_per_cu, best, flops / best / 1e9);
}

int main() {
    float* d_out;
    hipMalloc(&d_out, 4);
    for (int b = 1; b <= 4; ++b) {
        run<3, 2, 0>("reg-only CT3 PT2", b, d_out);
        run<3, 2, 1>("lds-read CT3 PT2", b, d_out);
    }
    run<1, 2, 0>("reg-only CT1 PT2", 4, d_out);
    run<1, 2, 1>("lds-read CT1 PT2", 4, d_out);
    run<4, 2, 0>("reg-only CT4 PT2", 4, d_out);
    run<4, 2, 1>("lds-read CT4 PT2", 4, d_out);
    run<2, 1, 1>("lds-read CT2 PT1", 4, d_out);
    for (int b = 2; b <= 4; ++b) run<3, 2, 2>("barriers CT3 PT2 256thr", b, d_out);
    for (int b = 4; b <= 8; b += 2) run<3, 2, 2, 128>("barriers CT3 PT2 128thr", b, d_out);
    for (int b = 8; b <= 16; b += 4) run<3, 2, 2, 64>("barriers CT3 PT2 64thr", b, d_out);
    run<3, 2, 3, 256, 32>("lds-read + 32 VALU", 4, d_out);
    run<3, 2, 3, 256, 64>("lds-read + 64 VALU", 4, d_out);
    run<3, 2, 3, 256, 128>("lds-read + 128 VALU", 4, d_out);
    run<3, 2, 3, 256, 256>("lds-read + 256 VALU", 4, d_out);
    run<3, 2, 4, 256, 16>("lds-read + 16 ds_write_b32", 4, d_out);
    run<3, 2, 4, 256, 32>("lds-read + 32 ds_write_b32", 4, d_out);
    run<3, 2, 5, 256, 16>("lds-read + 4 ds_write_b128", 4, d_out);
    run<3, 2, 5, 256, 32>("lds-read + 8 ds_write_b128", 4, d_out);
    run<1, 2, 2>("barriers CT1 PT2 256thr", 4, d_out);
    run<4, 2, 2>("barriers CT4 PT2 256thr", 4, d_out);
    // sustained: 40 back-to-back launches (~power/clock steady state)
    {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        const int blocks = 1024, iters = 4000, n = 40;
        hipEventRecord(e0, 0);
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL((mfma_loop<3, 2, 0>), dim3(blocks), dim3(256), 0, 0, d_out, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)n * blocks * 4 * iters * 8.0 * 6 * 2048;
        printf("sustained reg-only CT3 PT2: %d launches %8.2f ms  %7.2f TFLOP/s\n", n, ms, flops / ms / 1e9);
    }
    return 0;
}
