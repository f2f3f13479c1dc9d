// Internal declarations shared by the translation units of libposepipe_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <mutex>

#include "posepipe_hip.h"

void pp_set_error(const char* fmt, ...);

#define PP_HIP_CHECK(expr)                                                              \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess) {                                                         \
            pp_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                         __LINE__);                                                     \
            return PP_ERR_HIP;                                                          \
        }                                                                               \
    } while (0)

#define PP_REQUIRE(cond, ...)            \
    do {                                 \
        if (!(cond)) {                   \
            pp_set_error(__VA_ARGS__);   \
            return PP_ERR_ARG;           \
        }                                \
    } while (0)

// Per-kernel function attributes (hipFuncAttributeMaxDynamicSharedMemorySize) belong to the DEVICE's copy of the code object: a
// process that drives several devices has to set them once per device, not once per process.  One object per call site.
struct PpPerDeviceOnce {
    std::mutex m;
    unsigned long long done = 0;          // bit d: set on device d
    template <class F> void run(F&& f) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::lock_guard<std::mutex> g(m);
        if ((done >> (dev & 63)) & 1ull) return;
        f();
        done |= 1ull << (dev & 63);
    }
};

struct pp_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = true;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    // frame upload path: a second (copy) stream so that host->device transfers of the next chunk overlap compute
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_upload = nullptr;     // recorded after the last pp_upload_begin
    hipEvent_t ev_consumed = nullptr;   // recorded by pp_upload_release: compute no longer reads the buffer
    // grow-only device scratch used to stage PP_MEM_HOST arguments
    void* scratch = nullptr;
    size_t scratch_bytes = 0;
    int ensure_scratch(size_t bytes);
};

// Simple bump allocator over ctx scratch for one call.
struct ScratchCursor {
    pp_ctx* ctx;
    size_t off = 0;
    explicit ScratchCursor(pp_ctx* c) : ctx(c) {}
    static size_t align(size_t b) { return (b + 255) & ~size_t(255); }
    template <class T>
    T* take(size_t count) {
        T* p = reinterpret_cast<T*>(static_cast<char*>(ctx->scratch) + off);
        off += align(count * sizeof(T));
        return p;
    }
};

// ---- roctx ranges around the stages (rocprofv3 --marker-trace shows them on the host timeline, each enclosing the
// launches of one stage; the per-stage DEVICE times are the HIP-event figures of pp_detector_timing / pp_topdown_timing).  The
// marker library is looked up at run time (librocprofiler-sdk-roctx.so, else libroctx64.so): absent = no-ops, no link dependency.
void pp_range_push(const char* name);
void pp_range_pop();
struct PpRange {
    explicit PpRange(const char* name) { pp_range_push(name); }
    ~PpRange() { pp_range_pop(); }
    PpRange(const PpRange&) = delete;
    PpRange& operator=(const PpRange&) = delete;
};
// consecutive stages of one function: next(name) closes the current range and opens the next; balanced on every return path
struct PpStages {
    bool open = false;
    void next(const char* name) {
        if (open) pp_range_pop();
        open = name != nullptr;
        if (open) pp_range_push(name);
    }
    ~PpStages() { next(nullptr); }
};

// ---- convolution (conv_igemm.hip) ----------------------------------------------------------
struct ConvArgs {
    const float* x;
    const float* w;
    const float* bias;
    const float* res1;
    const float* res2;
    float* y;
    int N, Hin, Win, Cin, Hout, Wout, Cout, CoutPad;
    int KH, KW, stride, pad_h, pad_w, dil_h, dil_w;
    int K, Kpad, M, HWout;
    // division by the invariants HWout and Wout (Granlund-Montgomery): q = (t + ((n - t) >> s1)) >> s2, t = mulhi(m, n)
    unsigned div_hw_m, div_hw_s1, div_hw_s2, div_w_m, div_w_s1, div_w_s2;
    unsigned div_c_m, div_c_s1, div_c_s2, div_kw_m, div_kw_s1, div_kw_s2;   // / Cin, / KW
    int xcd_remap;   // XCD-aware tile order (performance only)
    unsigned x_bytes; // size of the input tensor of this launch (buffer-load range check)
    int relu, up_log2, out_nchw;
    int res1_shift, res1_off_w, res1_H, res1_W;
    int y_stride, y_coff;   // channels per pixel of the out buffer, first channel written (0 / 0: y_stride = Cout)
    // variant 3 (conv_igemm_p3.hip, TAB): per (16-k step, k-quad) {packed tap (dh << 16 | dw) or 0x7fff7fff, byte offset} on the
    // device, built once per layer geometry by pp_launch_conv; null: the kernel walks the taps itself
    const uint2* tap_table;
    int x_pad, y_pad, r1_pad, r2_pad;   // zero halo (right columns / bottom rows) of the input, output and residual buffers (pp_buf.pad)
    int no_bounds;          // set by pp_launch_conv: no tap can leave the image (no padding) and the tensor is < 2 GiB
    // conv_split.hip: the weights pre-split into bf16 planes in fragment order (pp_conv_split_weights); null: pp_launch_conv
    // builds a temporary copy when it picks the split kernel (single-op API; never under graph capture)
    const void* wsplit;
    // 0: single-op API -- the process-wide setting (pp_conv_exact) decides; 1 / 2: an op of a net created exact / split (ABI 7: a
    // net's numerics never change after creation; with 2 the split kernel runs exactly where the net built split weights)
    int numerics;
    // split kernels: 0 = three bf16 planes, six products; 1 = the fp16 form (two activation planes, three products; round 5).  Fixed
    // with the split weights (pp_conv_split_bytes / _weights / pp_launch_conv_split must see the same value)
    int split_f16;
    // fp16 form: per-sample running maximum of |x| (pp_amax.h; required by the fp16 kernels) and, any conv kernel, where to fold max |y|
    // per sample of what this launch stores (null: nobody needs it).  Device pointers to N slots each.
    const unsigned* x_amax;
    unsigned* y_amax;
};
// fp32 convolution on the bf16 matrix cores (three-way split, six products; conv_split.hip)
bool pp_conv_split_eligible(const ConvArgs& a);
size_t pp_conv_split_bytes(const ConvArgs& a);
int pp_conv_split_weights(const ConvArgs& a, void* out, hipStream_t stream);
int pp_launch_conv_split(const ConvArgs& a, hipStream_t stream);   // `a` as prepared by pp_launch_conv, a.wsplit set
// true: the kernel this layer runs on folds max |y| per sample into ConvArgs::y_amax in its epilogue (pp_amax.h) -- the split
// kernels always, the float32 kernels in their plain NHWC epilogue; false: the caller takes the maximum in a pass of its own
bool pp_conv_tracks_amax(const ConvArgs& a, bool split);
bool pp_conv_split_enabled();       // false: POSEPIPE_CONV_EXACT=1 or an explicit exact variant
bool pp_conv_split_f16_default();   // the process-wide default split form (pp_conv_split_kind / POSEPIPE_SPLIT_F16)
// builds (and caches per device) the tap tables the pipelined kernel may use for this geometry; call outside graph capture
int pp_conv_prepare(const ConvArgs& a);
int pp_conv_out_dim(int in, int k, int stride, int pad, int dil);
int pp_launch_conv(const ConvArgs& a, hipStream_t stream);
// software-pipelined variant (conv_igemm_p3.hip): same results; `a` as prepared by pp_launch_conv
int pp_launch_conv_p3(const ConvArgs& a, int ct, int pt, hipStream_t stream, bool fake_addresses = false);

struct PoolArgs {
    const float* x;
    float* y;
    int N, Hin, Win, C, Hout, Wout, KH, KW, stride, pad_h, pad_w;
    int x_stride, x_coff, y_stride, y_coff;   // channel slices of wider buffers (0 stride: = C)
};
int pp_launch_maxpool(const PoolArgs& a, hipStream_t stream);
// nn.AvgPool2d((kh, kw), stride), no padding, NHWC, c % 4 == 0 (elementwise.hip)
int pp_launch_avgpool(const float* x, float* y, int n, int hin, int win, int c, int kh, int kw, int stride, hipStream_t stream);

// ---- ViT encoder pieces, bf16 MFMA path (gemm_bf16.hip, vit_encoder.hip) ---------------------------
struct GemmArgs {
    const void* A;       // [M][K] bf16 activations
    const void* B;       // [N][K] bf16 weights (nn.Linear layout)
    const float* bias;   // [N] or null
    const float* res;    // fp32 [M or res_mod][N] or null
    void* C;             // [M][N] fp32 or bf16
    int M, N, K;
    int act;             // 0 none, 1 GELU (erf)
    int out_bf16;
    int res_mod;         // > 0: residual row = m % res_mod (position embedding)
    int group_m;         // tile rows per L2 sweep group (0: default)
    // > 0 (bf16 output, N = 3 * dim): write C as [3][M / qkv_tokens][dim / qkv_hd][qkv_tokens][qkv_hd] (head-major q, k, v)
    int qkv_tokens, qkv_hd;
};
int pp_launch_gemm_bf16(const GemmArgs& a, hipStream_t stream);
int pp_gemm_bf16_prepare();
// ConvTranspose2d(4, 2, 1) + bias + ReLU as one bf16 GEMM + gather (deconv_bf16.hip); weights / bias are DEVICE pointers
struct pp_deconv_bf16;
int pp_deconv_bf16_create(const float* weights, const float* bias, int h, int w, int cin, int cout, int max_batch,
                          hipStream_t stream, pp_deconv_bf16** out);
void pp_deconv_bf16_destroy(pp_deconv_bf16* d);
int pp_deconv_bf16_run(pp_deconv_bf16* d, const float* x, float* out, int batch, int relu, hipStream_t stream);
int pp_launch_f32_to_bf16(const float* x, void* y, size_t n, hipStream_t stream);
// NV12 -> BGR u8 (nv12.hip)
int pp_launch_nv12_to_bgr(const unsigned char* nv12, unsigned char* bgr, int frames, int h, int w, hipStream_t stream);
// y = LayerNorm(x [+ pos[row % pos_mod]]) over the last dim; x_out (optional) receives x + pos in fp32
int pp_launch_layernorm(const float* x, const float* pos, int pos_mod, float* x_out, const float* gamma,
                        const float* beta, int rows, int dim, float eps, void* y, int out_bf16, hipStream_t stream);
// multi-head self-attention on a packed qkv tensor [batch * tokens][3][heads][head_dim] bf16 -> [batch * tokens][heads * head_dim]
// head_major: qkv is [3][batch][heads][tokens][head_dim] (GemmArgs::qkv_tokens) instead of packed token rows
int pp_launch_attention(const void* qkv, int batch, int tokens, int heads, int head_dim, void* out, hipStream_t stream,
                        int head_major = 0);
// [n][h][w][4 * c] (parity-major channel groups g = 2 * dy + dx) -> [n][2h][2w][c]
int pp_launch_depth_to_space(const float* x, float* y, int n, int h, int w, int c, hipStream_t stream);
// y[n][H][W][c] = act((((res1 + t[n][H >> u][W >> u][c]) + t2[.. >> u2]) + t3[.. >> u3]) + res2)   (elementwise.hip; res1 / res2 /
// t2 / t3 may be null)
int pp_launch_upsample_add(const float* t, const float* res1, const float* res2, float* y, int n, int H, int W, int c,
                           int up_log2, int relu, hipStream_t stream, const float* t2 = nullptr, int up2 = 0,
                           const float* t3 = nullptr, int up3 = 0, unsigned* y_amax = nullptr);   // y_amax: pp_amax.h
// encoder behind PP_OP_VIT_ENCODER; `params` is a DEVICE pointer into the program's fp32 weight blob
struct pp_vit_encoder;
size_t pp_vit_param_floats(int tokens, int dim, int depth, int hidden);
int pp_vit_encoder_create(const float* params, int tokens, int dim, int depth, int heads, int hidden, int max_batch,
                          hipStream_t stream, pp_vit_encoder** out);
void pp_vit_encoder_destroy(pp_vit_encoder* e);
int pp_vit_encoder_run(pp_vit_encoder* e, const float* in, float* out, int batch, hipStream_t stream);
void pp_vit_encoder_set_timing(pp_vit_encoder* e, int enable);
int pp_vit_encoder_get_timing(pp_vit_encoder* e, float ms3[3], int* n_gemm);

// ---- top-down pre/post (crop_affine.hip, dark_decode.hip) ----------------------------------------
struct PersonXform {
    double a00, a01, b0, a10, a11, b1;  // inverse map dst -> src (OpenCV warpAffine, after inversion)
    int frame;
    int valid;
};
// mmpose `_box2cs` + `get_affine_transform(rot=0)` + OpenCV's inversion; false for NaN boxes
// udp != 0: mmpose `TopDownAffine(use_udp=True)` (get_warp_matrix on image_size - 1) instead of the 3-point affine
bool pp_person_transform(const double* bbox_tlwh, int out_w, int out_h, float center_scale[4], PersonXform* t, int udp = 0);
// all pointers are device pointers; nothing is synchronised
int pp_enqueue_crop(hipStream_t s, const uint8_t* frames, int h, int w, const PersonXform* xf, int n_person,
                    int out_w, int out_h, const float* lut, const int32_t chan_map[3], int flip, float* out,
                    uint8_t* crop_u8);
int pp_enqueue_flip_w(hipStream_t s, const float* src, float* dst, int n, int h, int w);  // NHWC4 mirror

struct DecodeParams {
    int n, k, h, w;
    int shift_heatmap, post, blur_kernel;
};
int pp_enqueue_decode(hipStream_t s, const DecodeParams& p, const float* hm, const float* hm_flip,
                      const int32_t* flip_perm, const float* center_scale, float* kpts, float* merged);
