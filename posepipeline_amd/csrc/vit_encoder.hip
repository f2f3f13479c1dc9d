// ViT encoder (ViTPose backbone) on the bf16 matrix cores: LayerNorm, fused multi-head attention, the encoder
// executor behind PP_OP_VIT_ENCODER, and the depth-to-space step of the deconvolution head.
//
// ViTPose is not in the reference tree (SURVEY.md 8d "C5 ... out of contract"); BASELINE.json configs[4] names it, and it
// plugs into the same top-down slot as the HRNet programs (wrappers/mmpose.py:57,75 would load it through the same
// init_pose_model / inference_top_down_pose_model calls).  Architecture as published (ViTPose, mmpose 0.x fork:
// ViT backbone, patch 16 / padding 2, pre-norm blocks, final LayerNorm, two 4x4 stride-2 deconvolutions + 1x1 conv).
//
// Numerics: the residual stream stays fp32; the inputs of every contraction (LayerNorm output, qkv, softmax
// numerators, attention output, GELU output) are rounded to bf16 (RNE) and accumulated in fp32 on the MFMA.
#include "pp_internal.h"

#include <memory>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

namespace {

__device__ __forceinline__ unsigned short bf16_rne(float f) {
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    return (unsigned)bf16_rne(lo) | ((unsigned)bf16_rne(hi) << 16);
}

// ---- LayerNorm: one wave per row, the row lives in registers (two-pass mean / variance) ---------------------
template <int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ pos,
                                                        int pos_mod, float* x_out, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int rows, int dim, float eps,
                                                        void* y, int out_bf16) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int nvec = dim >> 2;
    const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * dim);
    const float4* pr = pos ? reinterpret_cast<const float4*>(pos + (size_t)(row % pos_mod) * dim) : nullptr;
    float4 v[NV];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = i * 64 + lane;
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < nvec) {
            v[i] = xr[c];
            if (pr) {
                const float4 p = pr[c];
                v[i].x += p.x; v[i].y += p.y; v[i].z += p.z; v[i].w += p.w;
            }
            if (x_out) reinterpret_cast<float4*>(x_out + (size_t)row * dim)[c] = v[i];
            sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum / (float)dim;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (i * 64 + lane < nvec) {
            const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            sq += (a * a + b * b) + (c * c + d * d);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    const float rstd = 1.0f / sqrtf(sq / (float)dim + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = i * 64 + lane;
        if (c < nvec) {
            const float4 g = reinterpret_cast<const float4*>(gamma)[c], b = reinterpret_cast<const float4*>(beta)[c];
            const float o0 = (v[i].x - mean) * rstd * g.x + b.x, o1 = (v[i].y - mean) * rstd * g.y + b.y;
            const float o2 = (v[i].z - mean) * rstd * g.z + b.z, o3 = (v[i].w - mean) * rstd * g.w + b.w;
            if (out_bf16)
                reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(y) + (size_t)row * dim)[c] =
                    make_uint2(pack_bf16(o0, o1), pack_bf16(o2, o3));
            else
                reinterpret_cast<float4*>(reinterpret_cast<float*>(y) + (size_t)row * dim)[c] = make_float4(o0, o1, o2, o3);
        }
    }
}

// ---- attention: one block per (sample, head); K and V^T of the head in LDS, everything else in registers ------
// S^T = K . Q^T puts, in each lane, 4 consecutive keys of one query per 16x16 tile -- exactly the k-slots of the next
// MFMA's B operand when V^T's k-slots are read with the same key permutation, so P never leaves the registers.
template <int T, int HD>
__global__ __launch_bounds__(256, 2) void attention_kernel(const unsigned short* __restrict__ qkv, unsigned short* __restrict__ out,
                                                        int heads, float scale, int head_major) {
    constexpr int HDP = (HD + 31) / 32 * 32;   // head dim padded to the MFMA K step (zeros)
    constexpr int KSTR = HDP * 2 + 16;         // bytes per K row; rows 0..15 land in 16 distinct 16-byte bank groups
    constexpr int VSTR = T * 2 + 16;           // bytes per V^T row
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sK = smem;
    unsigned char* sV = smem + T * KSTR;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / heads, h = blockIdx.x - b * heads;
    const int D = heads * HD;
    // packed: rows of [3][heads][HD] per token (the standalone ABI);  head-major: [3][batch][heads][T][HD], what the encoder's
    // qkv GEMM writes (GemmArgs::qkv_tokens), so that every (sample, head) slice is one contiguous 30 KB slab
    const int ld = head_major ? HD : 3 * D;
    const size_t slab = (size_t)T * HD, nbh = (size_t)gridDim.x;
    const unsigned short* qp = head_major ? qkv + (size_t)blockIdx.x * slab : qkv + (size_t)b * T * ld + h * HD;
    const unsigned short* kp = head_major ? qp + nbh * slab : qp + D;
    const unsigned short* vp = head_major ? kp + nbh * slab : kp + D;

    for (int idx = tid; idx < T * (HDP / 8); idx += 256) {
        const int t = idx / (HDP / 8), c = idx - t * (HDP / 8);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (c < HD / 8) v = *reinterpret_cast<const uint4*>(kp + (size_t)t * ld + c * 8);
        *reinterpret_cast<uint4*>(sK + t * KSTR + c * 16) = v;
    }
    // V^T: lane = token (consecutive 2-byte LDS columns: conflict-free transposed writes); each wave owns T / 4 tokens and
    // walks the HD / 8 chunks of their rows back to back, so the 2 cache lines of a row are fetched once and hit in L1
    // for the other chunks (a token-major sweep over all 192 tokens re-fetched every line HD / 8 times)
    for (int c = 0; c < HD / 8; ++c) {
        const int t = wave * (T / 4) + lane;
        if (lane >= T / 4) continue;
        const uint4 v = *reinterpret_cast<const uint4*>(vp + (size_t)t * ld + c * 8);
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            *reinterpret_cast<unsigned short*>(sV + (c * 8 + 2 * j) * VSTR + t * 2) = (unsigned short)(w[j] & 0xffffu);
            *reinterpret_cast<unsigned short*>(sV + (c * 8 + 2 * j + 1) * VSTR + t * 2) = (unsigned short)(w[j] >> 16);
        }
    }
    __syncthreads();

    const int r16 = lane & 15, kg = lane >> 4;
#pragma unroll 1
    for (int qf = wave; qf < T / 16; qf += 4) {
        const unsigned short* qrow = qp + (size_t)(qf * 16 + r16) * ld;
        bf16x8_t fq[HDP / 32];
#pragma unroll
        for (int kk = 0; kk < HDP / 32; ++kk) {
            const int col = kk * 32 + kg * 8;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (col < HD) v = *reinterpret_cast<const uint4*>(qrow + col);
            fq[kk] = *reinterpret_cast<bf16x8_t*>(&v);
        }
        f32x4_t s[T / 16];
#pragma unroll
        for (int f = 0; f < T / 16; ++f) {
            // keep the scheduler from hoisting all 36 K-fragment reads (378 registers, 1 block / CU without the fences;
            // 95 registers and 2 blocks / CU with them)
            if ((f & 3) == 0) __builtin_amdgcn_sched_barrier(0);
            s[f] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < HDP / 32; ++kk) {
                const bf16x8_t fk = *reinterpret_cast<const bf16x8_t*>(sK + (f * 16 + r16) * KSTR + (kk * 32 + kg * 8) * 2);
                s[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk, fq[kk], s[f], 0, 0, 0);
            }
        }
        // softmax over the keys of query r16: this lane holds keys f * 16 + kg * 4 + reg
        float mx = -INFINITY;
#pragma unroll
        for (int f = 0; f < T / 16; ++f) mx = fmaxf(fmaxf(fmaxf(s[f][0], s[f][1]), fmaxf(s[f][2], s[f][3])), mx);
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int f = 0; f < T / 16; ++f)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float e = __expf((s[f][j] - mx) * scale);
                s[f][j] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);

        f32x4_t o[HD / 16];
#pragma unroll
        for (int df = 0; df < HD / 16; ++df) o[df] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < T / 32; ++kb) {
            __builtin_amdgcn_sched_barrier(0);
            uint4 pp;
            pp.x = pack_bf16(s[2 * kb][0], s[2 * kb][1]);
            pp.y = pack_bf16(s[2 * kb][2], s[2 * kb][3]);
            pp.z = pack_bf16(s[2 * kb + 1][0], s[2 * kb + 1][1]);
            pp.w = pack_bf16(s[2 * kb + 1][2], s[2 * kb + 1][3]);
            const bf16x8_t fp = *reinterpret_cast<bf16x8_t*>(&pp);
#pragma unroll
            for (int df = 0; df < HD / 16; ++df) {
                const unsigned char* vrow = sV + (df * 16 + r16) * VSTR;
                const uint2 lo = *reinterpret_cast<const uint2*>(vrow + ((2 * kb) * 16 + kg * 4) * 2);
                const uint2 hi = *reinterpret_cast<const uint2*>(vrow + ((2 * kb + 1) * 16 + kg * 4) * 2);
                uint4 vv = make_uint4(lo.x, lo.y, hi.x, hi.y);
                o[df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8_t*>(&vv), fp, o[df], 0, 0, 0);
            }
        }
        const float inv = 1.0f / sum;
        unsigned short* orow = out + (size_t)(b * T + qf * 16 + r16) * D + h * HD;
#pragma unroll
        for (int df = 0; df < HD / 16; ++df)
            *reinterpret_cast<uint2*>(orow + df * 16 + kg * 4) =
                make_uint2(pack_bf16(o[df][0] * inv, o[df][1] * inv), pack_bf16(o[df][2] * inv, o[df][3] * inv));
    }
}

__global__ __launch_bounds__(256) void depth_to_space_kernel(const float4* __restrict__ x, float4* __restrict__ y, int n, int h,
                                                             int w, int c4) {
    const size_t total = (size_t)n * h * w * 4 * c4;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int cc = (int)(i % c4);
    size_t r = i / c4;
    const int ox = (int)(r % (2 * w)); r /= 2 * w;
    const int oy = (int)(r % (2 * h));
    const int b = (int)(r / (2 * h));
    const int g = (oy & 1) * 2 + (ox & 1);
    y[i] = x[(((size_t)b * h + (oy >> 1)) * w + (ox >> 1)) * 4 * c4 + g * c4 + cc];
}

}  // namespace

int pp_launch_layernorm(const float* x, const float* pos, int pos_mod, float* x_out, const float* gamma, const float* beta,
                        int rows, int dim, float eps, void* y, int out_bf16, hipStream_t stream) {
    PP_REQUIRE(rows > 0 && dim > 0 && (dim & 3) == 0 && dim <= 2048, "layernorm: dim = %d must be a multiple of 4, <= 2048", dim);
    PP_REQUIRE(!pos || pos_mod > 0, "layernorm: pos needs pos_mod > 0");
    const dim3 grid((rows + 3) / 4), block(256);
    const int nv = (dim / 4 + 63) / 64;
#define PP_LN(NV) hipLaunchKernelGGL((layernorm_kernel<NV>), grid, block, 0, stream, x, pos, pos_mod, x_out, gamma, beta, rows, dim, eps, y, out_bf16)
    switch (nv) {
        case 1: PP_LN(1); break;
        case 2: PP_LN(2); break;
        case 3: PP_LN(3); break;
        case 4: PP_LN(4); break;
        case 5: PP_LN(5); break;
        case 6: PP_LN(6); break;
        case 7: PP_LN(7); break;
        default: PP_LN(8); break;
    }
#undef PP_LN
    PP_HIP_CHECK(hipGetLastError());
    return PP_OK;
}

template <int T, int HD>
static int launch_attention(const void* qkv, int batch, int heads, void* out, int head_major, hipStream_t stream) {
    constexpr int HDP = (HD + 31) / 32 * 32;
    constexpr size_t lds = (size_t)T * (HDP * 2 + 16) + (size_t)HD * (T * 2 + 16);
    static PpPerDeviceOnce configured;
    configured.run([&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_kernel<T, HD>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
    const float scale = 1.0f / sqrtf((float)HD);
    hipLaunchKernelGGL((attention_kernel<T, HD>), dim3(batch * heads), dim3(256), lds, stream,
                       reinterpret_cast<const unsigned short*>(qkv), reinterpret_cast<unsigned short*>(out), heads, scale, head_major);
    PP_HIP_CHECK(hipGetLastError());
    return PP_OK;
}

int pp_launch_attention(const void* qkv, int batch, int tokens, int heads, int head_dim, void* out, hipStream_t stream,
                        int head_major) {
    PP_REQUIRE(batch > 0 && heads > 0, "attention: empty problem");
    if (tokens == 192 && head_dim == 80) return launch_attention<192, 80>(qkv, batch, heads, out, head_major, stream);
    if (tokens == 192 && head_dim == 64) return launch_attention<192, 64>(qkv, batch, heads, out, head_major, stream);
    pp_set_error("attention: (tokens %d, head_dim %d) is not built; available: (192, 80), (192, 64)", tokens, head_dim);
    return PP_ERR_UNSUPPORTED;
}

int pp_launch_depth_to_space(const float* x, float* y, int n, int h, int w, int c, hipStream_t stream) {
    PP_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0 && (c & 3) == 0, "depth_to_space: c = %d must be a multiple of 4", c);
    const size_t total = (size_t)n * h * w * c;   // float4 count = n * h * w * 4 * (c / 4)
    hipLaunchKernelGGL(depth_to_space_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                       reinterpret_cast<const float4*>(x), reinterpret_cast<float4*>(y), n, h, w, c / 4);
    PP_HIP_CHECK(hipGetLastError());
    return PP_OK;
}

// ---- encoder executor ---------------------------------------------------------------------------------------
size_t pp_vit_param_floats(int tokens, int dim, int depth, int hidden) {
    const size_t D = dim, H = hidden;
    const size_t per_block = 2 * D + 3 * D * D + 3 * D + D * D + D + 2 * D + H * D + H + D * H + D;
    return (size_t)tokens * D + per_block * depth + 2 * D;
}

struct VitBlock {
    const float *ln1_g, *ln1_b, *bqkv, *bproj, *ln2_g, *ln2_b, *b1, *b2;
    const unsigned short *wqkv, *wproj, *w1, *w2;
};

struct pp_vit_encoder {
    int tokens = 0, dim = 0, depth = 0, heads = 0, hidden = 0, max_batch = 0;
    int head_major = [] { const char* v = getenv("POSEPIPE_VIT_HEAD_MAJOR"); return v && atoi(v) != 0; }();
    const float* pos = nullptr;
    const float *lnf_g = nullptr, *lnf_b = nullptr;
    std::vector<VitBlock> blocks;
    unsigned short* wbf16 = nullptr;   // all matrices, bf16
    void* scratch = nullptr;           // X fp32 | h bf16 | qkv bf16 | att bf16 | mlp bf16
    float* X = nullptr;
    unsigned short *hbuf = nullptr, *qkv = nullptr, *att = nullptr, *mlp = nullptr;
    // optional per-kernel-family timing (bench.py's roofline leg): one event after every launch
    bool timing = false;
    std::vector<hipEvent_t> ev;
    std::vector<int> ev_kind;          // family of the launch that ENDS at event i: 0 gemm, 1 layernorm, 2 attention
    size_t ev_used = 0;
    int n_gemm = 0;
    void mark(int kind, hipStream_t s) {
        if (!timing) return;
        if (ev_used == ev.size()) {
            hipEvent_t e = nullptr;
            if (hipEventCreate(&e) != hipSuccess) return;
            ev.push_back(e);
            ev_kind.push_back(kind);
        }
        ev_kind[ev_used] = kind;
        (void)hipEventRecord(ev[ev_used++], s);
    }
    ~pp_vit_encoder() {
        for (auto e : ev) (void)hipEventDestroy(e);
        if (wbf16) (void)hipFree(wbf16);
        if (scratch) (void)hipFree(scratch);
    }
};

void pp_vit_encoder_set_timing(pp_vit_encoder* e, int enable) { e->timing = enable != 0; }

// {gemm, layernorm, attention} milliseconds and the GEMM launch count of the last run (the stream must be idle)
int pp_vit_encoder_get_timing(pp_vit_encoder* e, float ms3[3], int* n_gemm) {
    ms3[0] = ms3[1] = ms3[2] = 0.f;
    for (size_t i = 1; i < e->ev_used; ++i) {
        float ms = 0.f;
        PP_HIP_CHECK(hipEventElapsedTime(&ms, e->ev[i - 1], e->ev[i]));
        ms3[e->ev_kind[i]] += ms;
    }
    if (n_gemm) *n_gemm = e->n_gemm;
    return PP_OK;
}

void pp_vit_encoder_destroy(pp_vit_encoder* e) { delete e; }

// params: DEVICE pointer to the fp32 parameter block (layout in include/posepipe_hip.h, PP_OP_VIT_ENCODER)
int pp_vit_encoder_create(const float* params, int tokens, int dim, int depth, int heads, int hidden, int max_batch,
                          hipStream_t stream, pp_vit_encoder** out) {
    PP_REQUIRE(params && out, "vit encoder: NULL argument");
    PP_REQUIRE(dim % 128 == 0 && hidden % 128 == 0 && heads > 0 && dim % heads == 0 && depth > 0 && tokens > 0,
               "vit encoder: dim %d / hidden %d must be multiples of 128, dim divisible by heads %d", dim, hidden, heads);
    const int hd = dim / heads;
    PP_REQUIRE(tokens == 192 && (hd == 80 || hd == 64), "vit encoder: attention is built for 192 tokens and head_dim 64 / 80 (got %d, %d)",
               tokens, hd);
    std::unique_ptr<pp_vit_encoder> e(new pp_vit_encoder());
    e->tokens = tokens; e->dim = dim; e->depth = depth; e->heads = heads; e->hidden = hidden; e->max_batch = max_batch;
    const size_t D = dim, H = hidden;
    const size_t mat_per_block = 3 * D * D + D * D + H * D + D * H;
    PP_HIP_CHECK(hipMalloc((void**)&e->wbf16, mat_per_block * depth * sizeof(unsigned short)));
    const float* p = params;
    e->pos = p; p += (size_t)tokens * D;
    unsigned short* wb = e->wbf16;
    auto conv = [&](const float* src, size_t n) -> const unsigned short* {
        unsigned short* dst = wb;
        wb += n;
        return pp_launch_f32_to_bf16(src, dst, n, stream) == PP_OK ? dst : nullptr;
    };
    for (int i = 0; i < depth; ++i) {
        VitBlock b{};
        b.ln1_g = p; p += D; b.ln1_b = p; p += D;
        b.wqkv = conv(p, 3 * D * D); p += 3 * D * D; b.bqkv = p; p += 3 * D;
        b.wproj = conv(p, D * D); p += D * D; b.bproj = p; p += D;
        b.ln2_g = p; p += D; b.ln2_b = p; p += D;
        b.w1 = conv(p, H * D); p += H * D; b.b1 = p; p += H;
        b.w2 = conv(p, D * H); p += D * H; b.b2 = p; p += D;
        PP_REQUIRE(b.wqkv && b.wproj && b.w1 && b.w2, "vit encoder: weight conversion failed");
        e->blocks.push_back(b);
    }
    e->lnf_g = p; p += D; e->lnf_b = p;
    const size_t M = (size_t)max_batch * tokens;
    const size_t bytes = M * D * 4 + M * D * 2 + M * 3 * D * 2 + M * D * 2 + M * H * 2;
    PP_HIP_CHECK(hipMalloc(&e->scratch, bytes));
    char* s = static_cast<char*>(e->scratch);
    e->X = reinterpret_cast<float*>(s); s += M * D * 4;
    e->hbuf = reinterpret_cast<unsigned short*>(s); s += M * D * 2;
    e->qkv = reinterpret_cast<unsigned short*>(s); s += M * 3 * D * 2;
    e->att = reinterpret_cast<unsigned short*>(s); s += M * D * 2;
    e->mlp = reinterpret_cast<unsigned short*>(s);
    // one throw-away pass on the scratch buffers: sets the kernels' dynamic-LDS attributes now, so that a later first
    // launch inside a hipGraph capture (pp_net_capture) does not have to
    PP_HIP_CHECK(hipMemsetAsync(e->scratch, 0, bytes, stream));
    if (pp_gemm_bf16_prepare() != PP_OK) return PP_ERR_HIP;
    int rc = pp_vit_encoder_run(e.get(), e->X, e->X, 1, stream);
    if (rc != PP_OK) return rc;
    *out = e.release();
    return PP_OK;
}

// in: [batch][tokens][dim] fp32 patch embeddings (before the position embedding); out: same shape, after the final LayerNorm
int pp_vit_encoder_run(pp_vit_encoder* e, const float* in, float* out, int batch, hipStream_t stream) {
    PP_REQUIRE(e && in && out && batch > 0 && batch <= e->max_batch, "vit encoder: bad batch %d", batch);
    const int M = batch * e->tokens, D = e->dim, H = e->hidden;
    const float eps = 1e-6f;
    int rc;
    e->ev_used = 0;
    e->n_gemm = 0;
    e->mark(1, stream);   // start of the first interval
    auto gemm = [&](const void* A, const void* W, const float* bias, const float* res, void* C, int N, int K, int act, int obf,
                    int qkv_tokens = 0, int qkv_hd = 0) {
        GemmArgs g{};
        g.A = A; g.B = W; g.bias = bias; g.res = res; g.C = C; g.M = M; g.N = N; g.K = K; g.act = act; g.out_bf16 = obf;
        g.qkv_tokens = qkv_tokens; g.qkv_hd = qkv_hd;
        const int r = pp_launch_gemm_bf16(g, stream);
        e->mark(0, stream);
        ++e->n_gemm;
        return r;
    };
    auto ln = [&](const float* x, const float* pos, float* x_out, const float* g, const float* b, void* y, int obf) {
        const int r = pp_launch_layernorm(x, pos, pos ? e->tokens : 0, x_out, g, b, M, D, eps, y, obf, stream);
        e->mark(1, stream);
        return r;
    };
    for (int i = 0; i < e->depth; ++i) {
        const VitBlock& b = e->blocks[i];
        if (i == 0) rc = ln(in, e->pos, e->X, b.ln1_g, b.ln1_b, e->hbuf, 1);   // x = patch_embed + pos, h = LN1(x)
        else rc = ln(e->X, nullptr, nullptr, b.ln1_g, b.ln1_b, e->hbuf, 1);
        if (rc != PP_OK) return rc;
        // head-major q / k / v (POSEPIPE_VIT_HEAD_MAJOR=1) makes the attention loads contiguous (3.66 -> 3.38 ms per step of
        // 128 passes) but the scattering qkv epilogue costs more than that (GEMMs 34.3 -> 35.5 ms): off by default
        const int hm = e->head_major;
        if ((rc = gemm(e->hbuf, b.wqkv, b.bqkv, nullptr, e->qkv, 3 * D, D, 0, 1, hm ? e->tokens : 0, hm ? D / e->heads : 0)) != PP_OK) return rc;
        if ((rc = pp_launch_attention(e->qkv, batch, e->tokens, e->heads, D / e->heads, e->att, stream, hm)) != PP_OK) return rc;
        e->mark(2, stream);
        if ((rc = gemm(e->att, b.wproj, b.bproj, e->X, e->X, D, D, 0, 0)) != PP_OK) return rc;
        if ((rc = ln(e->X, nullptr, nullptr, b.ln2_g, b.ln2_b, e->hbuf, 1)) != PP_OK) return rc;
        if ((rc = gemm(e->hbuf, b.w1, b.b1, nullptr, e->mlp, H, D, 1, 1)) != PP_OK) return rc;
        if ((rc = gemm(e->mlp, b.w2, b.b2, e->X, e->X, D, H, 0, 0)) != PP_OK) return rc;
    }
    return ln(e->X, nullptr, nullptr, e->lnf_g, e->lnf_b, out, 0);
}

// ---- C ABI: the building blocks on their own (device pointers) ------------------------------------------------
extern "C" {

int pp_f32_to_bf16(pp_ctx* ctx, const float* x, uint16_t* y, size_t n) {
    PP_REQUIRE(ctx && x && y, "pp_f32_to_bf16: NULL argument");
    return pp_launch_f32_to_bf16(x, y, n, ctx->stream);
}

int pp_gemm_bf16(pp_ctx* ctx, const uint16_t* a, const uint16_t* w, const float* bias, const float* res, int res_mod,
                 void* c, int m, int n, int k, int act, int out_bf16) {
    PP_REQUIRE(ctx && a && w && c, "pp_gemm_bf16: NULL argument");
    PP_REQUIRE(act == 0 || act == 1, "pp_gemm_bf16: act must be 0 (none) or 1 (GELU)");
    GemmArgs g{};
    g.A = a; g.B = w; g.bias = bias; g.res = res; g.res_mod = res_mod; g.C = c; g.M = m; g.N = n; g.K = k;
    g.act = act; g.out_bf16 = out_bf16;
    return pp_launch_gemm_bf16(g, ctx->stream);
}

int pp_layernorm(pp_ctx* ctx, const float* x, const float* gamma, const float* beta, int rows, int dim, float eps, void* y,
                 int out_bf16) {
    PP_REQUIRE(ctx && x && gamma && beta && y, "pp_layernorm: NULL argument");
    return pp_launch_layernorm(x, nullptr, 0, nullptr, gamma, beta, rows, dim, eps, y, out_bf16, ctx->stream);
}

int pp_attention_bf16(pp_ctx* ctx, const uint16_t* qkv, int batch, int tokens, int heads, int head_dim, uint16_t* out) {
    PP_REQUIRE(ctx && qkv && out, "pp_attention_bf16: NULL argument");
    return pp_launch_attention(qkv, batch, tokens, heads, head_dim, out, ctx->stream);
}

}  // extern "C"
