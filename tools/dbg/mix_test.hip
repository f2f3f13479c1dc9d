#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__global__ void k(const float* x, float s, unsigned* ref0, unsigned* ref1, unsigned* got0, unsigned* got1, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x0 = x[2 * i], x1 = x[2 * i + 1];
    // reference: the sequence of split4h
    const f32x2_t a = f32x2_t{x0, x1} * s;
    const f16x2_t a0 = __builtin_convertvector(a, f16x2_t);
    const f32x2_t ra = a - __builtin_convertvector(a0, f32x2_t);
    const f16x2_t a1 = __builtin_convertvector(ra, f16x2_t);
    ref0[i] = __builtin_bit_cast(unsigned, a0);
    ref1[i] = __builtin_bit_cast(unsigned, a1);
    unsigned d = 0, e = 0;
    asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(d) : "v"(x0), "v"(s));
    asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(d) : "v"(x1), "v"(s));
    asm volatile("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "+v"(e) : "v"(x0), "v"(s), "v"(d));
    asm volatile("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(e) : "v"(x1), "v"(s), "v"(d));
    got0[i] = d;
    got1[i] = e;
}
int main() {
    const int n = 1 << 20;
    std::vector<float> h(2 * n);
    srand(1);
    for (int i = 0; i < 2 * n; ++i) {
        unsigned u = ((unsigned)rand() << 16) ^ (unsigned)rand();
        float f;
        memcpy(&f, &u, 4);
        if (!(f == f) || f > 1e30f || f < -1e30f) f = (float)(rand() % 2000 - 1000) * 1e-3f;
        h[i] = (i % 3 == 0) ? f : (float)(rand() % 200001 - 100000) * (i % 5 == 0 ? 1e-7f : 1e-2f);
    }
    h[0] = 0.f; h[1] = -0.f; h[2] = 65504.f; h[3] = -70000.f; h[4] = 1e-8f; h[5] = 6.1e-5f;
    float* dx; unsigned *r0, *r1, *g0, *g1;
    hipMalloc(&dx, 2 * n * 4); hipMalloc(&r0, n * 4); hipMalloc(&r1, n * 4); hipMalloc(&g0, n * 4); hipMalloc(&g1, n * 4);
    hipMemcpy(dx, h.data(), 2 * n * 4, hipMemcpyHostToDevice);
    std::vector<unsigned> a(n), b(n), c(n), d(n);
    for (float s : {1.f, 0.25f, 16384.f, 1.52587890625e-05f}) {
        k<<<n / 256, 256>>>(dx, s, r0, r1, g0, g1, n);
        hipMemcpy(a.data(), r0, n * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), r1, n * 4, hipMemcpyDeviceToHost);
        hipMemcpy(c.data(), g0, n * 4, hipMemcpyDeviceToHost); hipMemcpy(d.data(), g1, n * 4, hipMemcpyDeviceToHost);
        long bad0 = 0, bad1 = 0, zsign = 0;
        for (int i = 0; i < n; ++i) {
            if (a[i] != c[i]) { if (((a[i] ^ c[i]) & 0x7fff7fffu) == 0) ++zsign; else ++bad0; }
            if (b[i] != d[i]) { if (((b[i] ^ d[i]) & 0x7fff7fffu) == 0) ++zsign; else { if (bad1 < 3) printf("  i=%d x=%g,%g ref %08x got %08x (h0 %08x)\n", i, h[2*i], h[2*i+1], b[i], d[i], a[i]); ++bad1; } }
        }
        printf("scale %g: h0 mismatches %ld, h1 mismatches %ld, sign-of-zero only %ld of %d\n", s, bad0, bad1, zsign, n);
    }
    return 0;
}
