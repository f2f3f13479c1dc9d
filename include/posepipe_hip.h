/*
 * posepipe_hip.h -- C ABI of libposepipe_hip.so: the MI355X (gfx950) hot path of
 * PosePipe's detect/track -> top-down 2D -> 3D lifting cascade.
 *
 * The reference (peabody124/PosePipeline) has no FFI for this path: its boundary is the
 * Python calling convention between DataJoint `make()` methods and the wrapper functions
 *   pose_pipeline/wrappers/mmtrack.py:8      mmtrack_bounding_boxes(file_path, method)
 *   pose_pipeline/wrappers/mmpose.py:26      mmpose_top_down_person(key, method)
 *   pose_pipeline/wrappers/videopose3d.py:19 process_videopose3d(key, batch_size, transform_coco)
 * whose arithmetic lives in un-vendored third-party packages (mmpose / mmdet / mmtrack /
 * VideoPose3D / OpenCV).  Each entry point below names the reference line(s) or third-party
 * op it replaces.  posepipeline_amd/wrappers/ holds the drop-in Python callables that bind
 * these symbols through ctypes (see INTEGRATION.md for the stub a maintainer would add).
 *
 * Conventions
 *   - every function returns 0 on success, a negative pp_status on failure; the message is
 *     available from pp_last_error() (thread-local).
 *   - all array arguments are caller-owned, contiguous, with explicit dims.  `mem` says whether
 *     the pointers are host (PP_MEM_HOST: the library stages them through its own device
 *     buffers) or device (PP_MEM_DEVICE: used in place, e.g. a torch tensor's data_ptr()).
 *   - one HIP stream per ctx; calls on one ctx are not thread-safe, distinct ctxs are.
 *   - no torch / C++ types cross this boundary.
 */
#ifndef POSEPIPE_HIP_H
#define POSEPIPE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PP_ABI_VERSION 10  /* 2: pp_op gained out_c_off / in_c_off / pad_end; PP_ACT_* activations
                              3: PP_OP_VIT_ENCODER / PP_OP_DEPTH_TO_SPACE, bf16 building blocks, UDP top-down (post 2)
                              4: pp_memcpy_d2d, pp_net_create_mem (weights already on the device, e.g. an RCCL broadcast)
                              5: pp_buf.pad (zero halo of conv-only buffers), PP_OP_AVGPOOL, pp_crop_resize_bilinear,
                                 pp_conv_force / pp_conv_variant
                              6: pp_conv_exact, pp_net_conv_kinds; pp_conv_variant accepts 4 (fp32 convolutions on the bf16 matrix cores by a
                                 three-way operand split are the default where a layer is eligible)
                              7: pp_net_create_ex / pp_net_numerics: the numerics of a program are fixed when it is created (a per-net
                                 property, no longer read from the process-wide switch at launch time); pp_op gained in2 / in3 /
                                 up2_log2 / up3_log2 (sizeof(pp_op) 104 -> 120)
                              8: pp_upload_begin_nv12 / pp_nv12_to_bgr (NV12 frame source)
                              9: PP_NET_NUMERICS_SPLIT_BF16 / _F16, pp_conv_split_kind, pp_net_split_kind: the split convolutions have a
                                 second form -- two float16 terms per operand, three products (conv_split.hip, round 5);
                                 pp_detector_constants
                              10: pp_detector_enable_margins / pp_detector_margins (per-frame decision margins of the detection path);
                                 pp_net_input_amax is a one-shot promise */

typedef enum {
    PP_OK = 0,
    PP_ERR_ARG = -1,      /* bad argument */
    PP_ERR_HIP = -2,      /* HIP runtime error (no GPU, launch failure, OOM ...) */
    PP_ERR_STATE = -3,    /* object used in the wrong state */
    PP_ERR_UNSUPPORTED = -4
} pp_status;

typedef enum { PP_MEM_HOST = 0, PP_MEM_DEVICE = 1 } pp_mem_kind;

typedef struct pp_ctx pp_ctx;   /* device + stream + scratch */
typedef struct pp_net pp_net;   /* a compiled layer program + resident weights/activations */
typedef struct pp_tracker pp_tracker; /* host-side multi-object tracker state */
typedef struct pp_topdown pp_topdown; /* fused top-down 2D stage (crop -> backbone -> decode) */

/* ---- library / context ------------------------------------------------------------------ */
int pp_abi_version(void);
const char* pp_last_error(void);
int pp_device_count(void);                         /* 0 when no HIP device is visible */
int pp_ctx_create(int device, pp_ctx** out);       /* fails with PP_ERR_HIP when there is no GPU */
void pp_ctx_destroy(pp_ctx* ctx);
int pp_ctx_set_stream(pp_ctx* ctx, void* hip_stream); /* adopt an external hipStream_t (e.g. torch's) */
int pp_ctx_synchronize(pp_ctx* ctx);
/* HIP-event timer on the ctx stream (bench.py's roofline leg): start/stop bracket launches. */
int pp_timer_start(pp_ctx* ctx);
int pp_timer_stop(pp_ctx* ctx, float* elapsed_ms); /* synchronises on the stop event */
/* A/B knobs of tests and profiles that select between two kernels giving IDENTICAL bits (never numerics): "decode_generic"
 * (1: the generic flip-merge / decode kernel instead of the fast one), "split_gemm_epilogue" (1: the product kernel's epilogue
 * transposed through LDS, 0: from registers).  value -1: back to the environment's choice (POSEPIPE_DECODE_GENERIC /
 * POSEPIPE_SPLIT_GEMM_EPI, read once per process).  Process-wide, thread-safe. */
int pp_debug_knob(const char* name, int value);
/* raw device memory helpers so that a host language without a HIP binding can keep data resident */
int pp_malloc(pp_ctx* ctx, size_t bytes, void** dptr);
int pp_free(pp_ctx* ctx, void* dptr);
int pp_memcpy_h2d(pp_ctx* ctx, void* dst, const void* src, size_t bytes);
int pp_memcpy_d2h(pp_ctx* ctx, void* dst, const void* src, size_t bytes);
/* device -> device on the ctx stream, asynchronous (ordered with the other work of the ctx) */
int pp_memcpy_d2d(pp_ctx* ctx, void* dst, const void* src, size_t bytes);
/* Frame upload path (replaces the per-frame cv2.VideoCapture.read() -> model H2D of wrappers/mmtrack.py:38-45 and
 * wrappers/mmpose.py:60-75 with a chunked, double-buffered transfer): page-locked host staging buffers and an
 * asynchronous host->device copy on the context's copy stream, so chunk k+1 is uploaded while chunk k computes.
 *   pp_upload_begin   enqueue the copy (waits, on the device, for the last pp_upload_release);
 *   pp_upload_wait    make the compute stream wait for the last upload (host_sync != 0: also block the host);
 *   pp_upload_release mark, in compute-stream order, that the device buffer may be overwritten again. */
int pp_host_alloc(pp_ctx* ctx, size_t bytes, void** host_ptr);
int pp_host_free(pp_ctx* ctx, void* host_ptr);
int pp_upload_begin(pp_ctx* ctx, void* dst_device, const void* src_host, size_t bytes);
/* ABI 8 -- NV12 frame source (a decoder's native output: Y plane [h][w], then interleaved UV [h/2][w]; half the bytes of
 * the BGR frames cv2.VideoCapture.read() hands the reference at pose_pipeline/pipeline.py:47-87, wrappers/mmtrack.py:38-45,
 * wrappers/mmpose.py:55-75).  h and w even.  The conversion is OpenCV's cvtColor(COLOR_YUV2BGR_NV12) bit for
 * bit (ITU-R BT.601 limited range, 20-bit fixed point; oracle/nv12.py).
 *   pp_upload_begin_nv12  pp_upload_begin for `frames` NV12 frames: host -> tmp_nv12_device on the copy stream, then the
 *                         conversion into dst_bgr_device ([frames][h][w][3] u8) on the same stream; pp_upload_wait /
 *                         pp_upload_release as for pp_upload_begin.  tmp_nv12_device: frames * h * w * 3 / 2 bytes, may be
 *                         the same buffer for every call (the copy stream is in order).
 *   pp_nv12_to_bgr        the conversion alone, device -> device on the ctx stream. */
int pp_upload_begin_nv12(pp_ctx* ctx, void* dst_bgr_device, void* tmp_nv12_device, const void* src_nv12_host, int frames,
                         int height, int width);
int pp_nv12_to_bgr(pp_ctx* ctx, const uint8_t* nv12_device, int frames, int height, int width, uint8_t* bgr_device);
int pp_upload_wait(pp_ctx* ctx, int host_sync);
int pp_upload_release(pp_ctx* ctx);

/* ---- layer programs (backbones) ----------------------------------------------------------
 * A network is a straight-line program of ops over numbered NHWC fp32 activation buffers.
 * Replaces the third-party forward passes reached from
 *   wrappers/mmpose.py:75  (mmpose TopDown.forward_test -> HRNet, arch spec
 *                           3rdparty/mmpose/config/top_down/darkpose/coco/hrnet_w48_coco_384x288_dark.py:44-79)
 *   wrappers/mmtrack.py:45 (mmdet FasterRCNN, 3rdparty/mmtracking/_base_/models/faster_rcnn_r50_fpn.py:1-112)
 *   wrappers/videopose3d.py:82 (VideoPose3D TemporalModel, wrappers/videopose3d.py:46-50)
 * Convolutions run as implicit GEMM on v_mfma_f32_16x16x4_f32 with BatchNorm folded into
 * (weight, bias) and bias / residual adds / ReLU / nearest-upsample-accumulate fused in the
 * epilogue.  The fp32 MFMA accumulates each output as a k-ordered fmaf chain over
 * k = (kh, kw, cin), which is exactly what oracle/conv_ref.c does, so results are bit-exact.
 */
typedef enum {
    PP_OP_CONV = 1,      /* conv2d (conv1d when H == 1), see pp_op fields */
    PP_OP_MAXPOOL = 2,   /* kh x kw max pool, -inf padding (ResNet stem 3x3 s2 p1; FPN P6 1x1 s2) */
    PP_OP_ROIALIGN = 3,  /* reserved for the detector program */
    PP_OP_COPY = 4,      /* buffer copy (same dims) */
    PP_OP_VIT_ENCODER = 5,   /* ViT encoder on the bf16 matrix cores, see "ViT encoder" below */
    PP_OP_DEPTH_TO_SPACE = 6,/* in [h][w][4*cout] (channel groups g = 2*dy + dx) -> out [2h][2w][cout]; with four 2x2
                                convolutions writing the groups this is ConvTranspose2d(k=4, s=2, p=1) */
    PP_OP_DECONV_BF16 = 8,   /* ConvTranspose2d(4, 2, 1) + bias + act on the bf16 matrix cores: in [h][w][cin] fp32 ->
                                out [2h][2w][cout] fp32; w_off: fp32 W_all[16][cout][cin], block ((a*2+b)*2+r)*2+s =
                                w[:, :, 3-a-2r, 3-b-2s]^T (output parity (a, b), tap (r, s)); b_off: bias[cout];
                                cin % 64 == 0, cout % 8 == 0; relu PP_RELU_NONE / PP_RELU_LAST.  One GEMM with N = 16 cout
                                and a 4-term gather; operands rounded to bf16, fp32 accumulation (tolerance-based parity) */
    PP_OP_AVGPOOL = 9,       /* nn.AvgPool2d((kh, kw), stride), no padding: float32 sum in (kh, kw) order / (kh * kw)  (the
                                GlobalAveragePooling neck of mmtrack's ReID model, mot/deepsort/deepsort_*.py:27) */
    PP_OP_UPSAMPLE_ADD = 7,  /* out[y][x] = act(((((res1[y][x] + in[y >> up_log2][x >> up_log2]) + in2[y >> up2_log2][x >> up2_log2])
                                + in3[y >> up3_log2][x >> up3_log2]) + res2[y][x]): nearest upsample + accumulate of an HRNet
                                fuse layer in mmpose's summation order (res1, in2, in3, res2 optional; relu: PP_RELU_NONE /
                                PP_RELU_LAST); same additions, same order as a chain of convs with up_log2, in one pass */
} pp_op_type;

#define PP_RELU_NONE 0
#define PP_RELU_LAST 1   /* y = relu(conv + bias + res...) */
#define PP_RELU_FIRST 2  /* y = res + relu(conv + bias)   (VideoPose3D blocks) */
/* activations of the DeepSortYOLOv4 path (wrappers/deep_sort_yolov4/yolo4/model.py:25-75, tools/freeze_model.py),
 * applied like PP_RELU_FIRST: y = res + act(conv + bias).  Transcendentals are evaluated in double precision and
 * rounded to float once (Mish: tanh(softplus(x)) in double, rounded, then a float product with x). */
#define PP_ACT_LEAKY 3   /* LeakyReLU(alpha = 0.1f) */
#define PP_ACT_MISH 4    /* x * tanh(softplus(x)) */
#define PP_ACT_ELU 5     /* x > 0 ? x : exp(x) - 1 */
#define PP_ACT_SWISH 6   /* x * sigmoid(x)  (mmcv Swish: YOLOX of the ByteTrack config, _base_/models/yolox_x_8x8.py) */

typedef struct pp_op {
    int32_t type;
    int32_t in, out;          /* buffer ids */
    int32_t res1, res2;       /* residual buffer ids, -1 = none; order: ((acc+bias) + res1) + res2 */
    int32_t cin, cout;        /* cin % 4 == 0 (pad the blob); cout arbitrary */
    int32_t kh, kw, stride, pad_h, pad_w, dil_h, dil_w;
    int32_t relu;             /* PP_RELU_* */
    int32_t up_log2;          /* write each output to a 2^up x 2^up patch of `out` (nearest upsample) */
    int32_t out_nchw;         /* 1: write `out` as [n][cout][H][W] planes (heatmaps) */
    int32_t res1_shift;       /* read res1 at (h >> s, w >> s)  (FPN top-down add) */
    int32_t res1_off_w;       /* read res1 at w + off (VideoPose3D centre-cropped residual) */
    int32_t out_c_off;        /* write channels [out_c_off, out_c_off + cout) of a wider `out` buffer (Concatenate) */
    int32_t in_c_off;         /* max pool only: read channels [in_c_off, in_c_off + cin) of `in` */
    int32_t pad_end;          /* bit 0 / 1: one extra zero row at the bottom / column at the right (TensorFlow SAME);
                                 bit 2 / 3: pad_h / pad_w apply in front only: the last output row / column of the
                                 symmetric-padding result is not computed (the 2x2 sub-convolutions of a deconvolution) */
    int64_t w_off, b_off;     /* float offsets into the weight blob: W (layout below), bias[cout_pad16] */
    /* ABI 7 -- PP_OP_UPSAMPLE_ADD only (every other op: -1 / 0): two more coarse inputs, so that a whole HRNet fuse sum
     * y_i = relu(((partial + up(t_a)) + up(t_b)) + up(t_c)) is ONE pass over the fine map (mmpose's `y += ...` order) */
    int32_t in2, in3;         /* buffer ids, -1 = none */
    int32_t up2_log2, up3_log2;
} pp_op;

typedef struct pp_buf {
    int32_t h, w, c;          /* per-sample dims */
    int32_t pad;              /* > 0: the buffer is stored [h + pad][w + pad][c] per sample with `pad` ZERO columns at the right of
                                 every row and `pad` zero rows below every image (never written).  A padded 3x3 convolution
                                 reading it needs no bounds test: a tap left of column 0 lands in the previous row's zero
                                 columns, a tap above row 0 in the previous image's zero rows (or before the buffer, where the
                                 buffer descriptor returns zeros).  Only PP_OP_CONV may touch a padded buffer; buffers that
                                 pp_net_buffer / pp_net_forward expose must have pad = 0. */
} pp_buf;

/* weights: host pointer to the flat fp32 blob (copied to the device once).  max_batch fixes the
 * size of the resident activation arena. */
int pp_net_create(pp_ctx* ctx, const pp_op* ops, int n_ops, const pp_buf* bufs, int n_bufs,
                  const float* weights, size_t n_weights, int max_batch, pp_net** out);
/* same, with `weights` in `weights_mem` memory: PP_MEM_DEVICE takes a blob that is already resident (the tensor an
 * RCCL broadcast delivered -- wrappers/mmtrack.py:30 / wrappers/mmpose.py:57 load a checkpoint file per process instead)
 * and copies it device-to-device into the program's own allocation; no host round trip. */
int pp_net_create_mem(pp_ctx* ctx, const pp_op* ops, int n_ops, const pp_buf* bufs, int n_bufs,
                      const float* weights, size_t n_weights, int weights_mem, int max_batch, pp_net** out);
/* same, with the program's convolution numerics chosen explicitly (they are a property of the net, fixed here: split weights
 * are built at creation, and every later launch of this net -- from any thread, under any later pp_conv_exact call -- runs the
 * same kernels):  PP_NET_NUMERICS_DEFAULT = what pp_conv_exact / POSEPIPE_CONV_EXACT select at this moment (what pp_net_create
 * and pp_net_create_mem do), PP_NET_NUMERICS_EXACT = every layer on the float32 MFMA kernels (bit-identical to
 * oracle/conv_ref.c), PP_NET_NUMERICS_SPLIT = eligible layers on the bf16 matrix cores (three-way split, see pp_conv_exact). */
#define PP_NET_NUMERICS_DEFAULT 0
#define PP_NET_NUMERICS_EXACT 1
#define PP_NET_NUMERICS_SPLIT 2        /* the split form pp_conv_split_kind selects at this moment */
#define PP_NET_NUMERICS_SPLIT_BF16 3   /* three bfloat16 terms per operand, six products (rounds 2 - 4) */
#define PP_NET_NUMERICS_SPLIT_F16 4    /* two float16 terms per operand (per-channel / per-sample power-of-two scales), three products */
int pp_net_create_ex(pp_ctx* ctx, const pp_op* ops, int n_ops, const pp_buf* bufs, int n_bufs,
                     const float* weights, size_t n_weights, int weights_mem, int max_batch, int numerics, pp_net** out);
/* PP_NET_NUMERICS_EXACT or PP_NET_NUMERICS_SPLIT: what the net was created with (never DEFAULT) */
int pp_net_numerics(pp_net* net);
/* which split form a PP_NET_NUMERICS_SPLIT net runs (fixed at creation): PP_NET_NUMERICS_SPLIT_BF16 or PP_NET_NUMERICS_SPLIT_F16;
 * PP_NET_NUMERICS_EXACT for an exact net */
int pp_net_split_kind(pp_net* net);
void pp_net_destroy(pp_net* net);
/* device address of activation buffer `buf` (batch-major, then the pp_buf layout) */
int pp_net_buffer(pp_net* net, int buf, void** dptr, size_t* bytes_per_sample);
/* fp16-form programs scale every convolution input per sample by a power of two taken from the running maximum of the tensor
 * (pp_conv_split_kind).  For tensors produced inside the program the producing kernels track it; for a program INPUT the run takes
 * one extra pass over the buffer -- unless the caller's own producer supplies the maxima: *dptr receives the device array
 * (max_batch uint32: the bit pattern of max |x| over sample n, any upper bound >= it is valid), or NULL when no fp16-form
 * convolution reads `buf` (nothing to do).  The call is a ONE-SHOT promise: it covers the NEXT pp_net_run / pp_net_profile /
 * pp_net_capture whose op range reads `buf` -- the caller fills the array on the ctx stream before that run and calls this again
 * before every further one; a run without a fresh promise takes the maxima itself, and pp_net_forward (which overwrites the
 * buffer) voids a pending promise.  The address is stable for the life of the net.  (The detector's preprocess kernel and
 * RoIAlign do this.) */
int pp_net_input_amax(pp_net* net, int buf, void** dptr);
/* run ops [first, last) for `batch` samples (last < 0: to the end); inputs must already be in their buffers */
int pp_net_run(pp_net* net, int batch, int first_op, int last_op);
/* convenience: copy `in` into buffer in_buf, run everything, copy buffer out_buf to `out` */
int pp_net_forward(pp_net* net, int batch, int in_buf, const float* in, int out_buf, float* out,
                   int mem);
/* Programs with >= 8 ops run on 4 HIP streams ("lanes"): independent ops (HRNet branches, FPN / RPN levels)
 * overlap, dependencies become event waits.  enable = 0 launches every op on the ctx stream in program order
 * (profiling: per-kernel durations are additive only then).  Results are identical either way. */
int pp_net_set_lanes(pp_net* net, int enable);
/* number of lanes (default 4, POSEPIPE_NET_LANES): re-plans the op -> stream assignment of an existing net (synchronises; a captured
 * graph is dropped).  1 = every op on the ctx stream.  How many pays depends on the program: branches of small maps fill each other's
 * tails (HRNet at 256x192), large maps that fill the chip on their own only disturb each other (measured round 6: the 1080p cascade
 * runs 3 % faster on 2 lanes than on 4).  Results are identical for any count. */
int pp_net_set_lane_count(pp_net* net, int n_lanes);
/* capture ops [0, n_ops) at `batch` into a hipGraph and replay it on later pp_net_run calls */
int pp_net_capture(pp_net* net, int batch);
/* per-op elapsed time of the last profiled run (HIP events around each op), ms; NULL-safe */
int pp_net_profile(pp_net* net, int batch, float* ms_per_op);
/* which kernel family each op of this net launches (fixed at creation): 0 = not a convolution, 1 = float32 MFMA kernels
 * (conv_igemm*.hip), 2 = bf16-split kernel (conv_split.hip).  kinds: n_ops ints.  (bench.py prices the two families
 * against their own peaks.) */
int pp_net_conv_kinds(pp_net* net, int* kinds);

/* Tile configuration of the convolution kernel for all later launches of this process: ct = output-channel tile / 16
 * (1..4), pt = pixel tile / 64 (1, 2); 0 = automatic (posepipeline_amd/conv_tuning.txt, then the built-in heuristic).
 * Results do not depend on it; tools/autotune_conv.py uses it to measure every configuration per layer. */
int pp_conv_force(int ct, int pt);
/* Kernel variant for all later launches: 0 = two-barrier K step (conv_igemm.hip), 1 = three-stage software pipeline
 * (conv_igemm_p3.hip), 3 = 1 with per-geometry tap tables, 4 = the split kernel below where eligible,
 * -1 = default (POSEPIPE_CONV_VARIANT, else: the split kernel where eligible, 0 / 3 elsewhere).
 * 0, 1 and 3 give bit-identical results (the k-ordered float32 FMA chain of oracle/conv_ref.c). */
int pp_conv_variant(int variant);
/* Process-wide DEFAULT numerics: what nets created LATER with PP_NET_NUMERICS_DEFAULT get, and what the single-op entry point
 * pp_conv2d uses.  Nets that already exist are not affected (ABI 7: their numerics were fixed at creation); safe to call from
 * any thread.
 * Default (0, or -1 with POSEPIPE_CONV_EXACT unset): 3x3 / stride-1 convolutions, 1x1 convolutions from 256 input channels with
 * a multiple of 128 output channels (any from 1024), and full-cover 'valid' convolutions (the RoI head's FCs) run on
 * v_mfma_f32_32x32x16_bf16 -- every float32 operand is split EXACTLY into three bfloat16 values and the six
 * partial products down to 2^-16 relative weight are accumulated in float32 (conv_split.hip).  The dropped terms are <= 2^-23
 * of a product, one float32 rounding; measured against a float64 convolution the result is as accurate as the float32 FMA
 * chain (tests/test_gpu_split.py), but it is not bit-identical to it.
 * exact = 1 (or POSEPIPE_CONV_EXACT=1): every layer on the float32 MFMA kernels, bit-identical to oracle/conv_ref.c.
 * exact = -1: back to the environment's choice. */
int pp_conv_exact(int exact);
/* Process-wide DEFAULT split form: what PP_NET_NUMERICS_SPLIT (and DEFAULT, when it resolves to split) nets created LATER and the
 * single-op entry point get.  f16 = 1 (the default since round 5): two float16 terms per operand and three
 * v_mfma_f32_32x32x16_f16 products per term (half the matrix work of the six-product form; same float32 accumulation).  Weights are
 * normalised per output channel, activations PER SAMPLE by a power of two taken from the running maximum of the tensor -- tracked by
 * the kernel that produces it, or by one extra pass for tensors that come from outside the program -- so nothing saturates at any
 * magnitude of the data and a sample's result does not depend on what else is in the batch.  The precision is RELATIVE TO THE
 * SAMPLE'S MAXIMUM m (one scale per sample of the whole tensor, not per channel): elements down to 2^-17 m keep 22 significand
 * bits; below that the residual term is a float16 subnormal and the absolute error is <= 2^-39 m -- a large RELATIVE error for a
 * channel or pixel whose values are all below ~1e-5 of the sample's maximum (BN-folded nets can have such channels; the bf16 form
 * below has no such range limit, and PP_NET_NUMERICS_EXACT none at all).  Error against a float64 convolution: that of the float32
 * FMA chain or lower on inputs of one magnitude per sample, 1e-30 to 1e7 (tests/test_gpu_split.py; the test with channel ranges
 * spread over 2^20 holds the absolute bound above).  0: three bfloat16 terms, six products (rounds 2 - 4); -1: the environment's choice
 * (POSEPIPE_SPLIT_F16, default 1). */
int pp_conv_split_kind(int f16);

/* single convolution on caller-provided device/host buffers (tests, VideoPose3D, FC layers).
 * x: [n][hin][win][cin]; bias: [cout_pad16]; y per op flags.
 * w: [ceil(K/32)][cout_pad16][32], K = kh*kw*cin, k = (kh_i*kw + kw_i)*cin + c; within a chunk of 32 k's the
 * element 8*g + s of an output channel holds k = 32*chunk + 4*s + g (MFMA operand order; zero padded). */
int pp_conv2d(pp_ctx* ctx, const pp_op* op, int n, int hin, int win, const float* x,
              const float* w, const float* bias, const float* res1, const float* res2, float* y,
              int res1_h, int res1_w, int mem);

/* Crops for mmtrack's ReID branch (SortTracker.crop_imgs: `F.interpolate(img[:, :, y1:y2, x1:x2], (256, 128), 'bilinear',
 * align_corners=False)` on the DETECTOR's normalised input tensor, mot/deepsort/deepsort_*.py:43-49).  src: DEVICE [n_src][src_h]
 * [src_w][4] float32; rects5: HOST int32 [n][5] = (source image, x1, y1, x2, y2), x2 > x1, y2 > y1, inside the image;
 * out: DEVICE [n][out_h][out_w][4].  PyTorch's float32 upsample_bilinear2d arithmetic. */
int pp_crop_resize_bilinear(pp_ctx* ctx, const float* src_nhwc4, int n_src, int src_h, int src_w, const int32_t* rects5, int n,
                            int out_h, int out_w, float* out_nhwc4);

/* ---- ViT encoder (bf16 MFMA path) -----------------------------------------------------------
 * BASELINE.json configs[4] "ViTPose-H backbone (bf16 MFMA path)".  ViTPose is NOT in the reference tree; it fills the
 * same slot as the HRNet programs (the model `init_pose_model` builds at wrappers/mmpose.py:57 and
 * `inference_top_down_pose_model` runs at :75).  Architecture as published for ViTPose (mmpose 0.x fork): ViT with
 * patch 16 / padding 2, learned position embedding (cls slot folded in), pre-norm blocks (LayerNorm eps 1e-6, qkv bias,
 * erf GELU), final LayerNorm; head: two ConvTranspose2d(k4, s2, p1) + BN + ReLU and a 1x1 conv.
 *
 * PP_OP_VIT_ENCODER: `in`  [h][w][dim] fp32 = patch embeddings (a PP_OP_CONV with kh = kw = stride = 16, pad 2),
 *                    `out` [h][w][dim] fp32 = tokens after the final LayerNorm;  h * w = tokens (192 = 16 x 12).
 *   cin = cout = dim, kh = depth, kw = heads, stride = hidden / dim (MLP ratio); dim / heads in {64, 80}.
 *   w_off: fp32 parameter block  pos[tokens][dim],
 *          depth x { ln1_g[dim], ln1_b[dim], Wqkv[3 dim][dim], bqkv[3 dim], Wproj[dim][dim], bproj[dim],
 *                    ln2_g[dim], ln2_b[dim], W1[hidden][dim], b1[hidden], W2[dim][hidden], b2[dim] },
 *          lnf_g[dim], lnf_b[dim]          (matrices in nn.Linear layout [out][in]; qkv rows = (q|k|v, head, head_dim)).
 *   The matrices are converted to bf16 (RNE) once at pp_net_create.  Numerics: fp32 residual stream; LayerNorm
 *   output, qkv, softmax numerators exp(s - max), attention output and GELU output are rounded to bf16; every
 *   contraction accumulates in fp32 (v_mfma_f32_16x16x32_bf16).  Not bit-reproducible against a CPU restatement
 *   (the MFMA's internal summation order is not architected): parity tolerance is stated in tests/test_gpu_vit.py.
 *
 * The building blocks are exported on their own (tests, other callers); ALL pointers are device pointers:
 *   pp_gemm_bf16      C[m][n] = act(A[m][k] . W[n][k]^T + bias[n]) + res[row][n];  A, W bf16; bias / res fp32 or NULL;
 *                     act 0 none / 1 GELU(erf); res row = m % res_mod when res_mod > 0; C fp32 or bf16 (out_bf16);
 *                     k % 64 == 0, n % 128 == 0.
 *   pp_layernorm      y = (x - mean) * rsqrt(var + eps) * gamma + beta over the last dim (biased variance), x fp32
 *   pp_attention_bf16 qkv [batch * tokens][3][heads][head_dim] bf16 -> out [batch * tokens][heads * head_dim] bf16,
 *                     softmax(q k^T / sqrt(head_dim)) v;  (tokens, head_dim) in {(192, 80), (192, 64)}
 *   pp_f32_to_bf16    round-to-nearest-even conversion
 */
/* HIP-event timing of the encoder's kernel families (bench.py's roofline leg).  enable != 0: record one event after
 * every launch of later runs.  ms3 (may be NULL): {bf16 GEMMs, LayerNorms, attention} milliseconds of the last recorded
 * run, n_gemm its GEMM launch count; synchronises the ctx stream. */
int pp_net_vit_timing(pp_net* net, int enable, float* ms3, int* n_gemm);
int pp_f32_to_bf16(pp_ctx* ctx, const float* x, uint16_t* y, size_t n);
int pp_gemm_bf16(pp_ctx* ctx, const uint16_t* a, const uint16_t* w, const float* bias, const float* res, int res_mod,
                 void* c, int m, int n, int k, int act, int out_bf16);
int pp_layernorm(pp_ctx* ctx, const float* x, const float* gamma, const float* beta, int rows, int dim, float eps,
                 void* y, int out_bf16);
int pp_attention_bf16(pp_ctx* ctx, const uint16_t* qkv, int batch, int tokens, int heads, int head_dim, uint16_t* out);

/* ---- top-down pre-processing --------------------------------------------------------------
 * Replaces mmpose `_box2cs` + `TopDownAffine` (cv2.warpAffine INTER_LINEAR, border 0) + ToTensor +
 * NormalizeTensor reached from wrappers/mmpose.py:75 (pipeline spec
 * hrnet_w48_coco_384x288_dark.py:87-90,129-144); in-tree twin utils/bounding_box.py:32-53.
 * frames: [n_frames][h][w][3] u8.  For person i: frame_idx[i], bbox_tlwh[i] (x,y,w,h doubles;
 * NaN = absent -> the output sample is all zeros and valid[i] = 0).
 * lut: [3][256] fp32 = ((v/255) - mean[c]) / std[c] computed by the caller in fp32.
 * chan_map[c] = source channel feeding output channel c (reproduces the reference's double
 * BGR<->RGB swap, wrappers/mmpose.py:73).
 * out: [n_out][out_h][out_w][4] fp32 NHWC (4th channel 0); with flip != 0, n_out = 2*n_person and
 * sample n_person+i is sample i mirrored in W (`img.flip(3)` of flip_test).
 * center_scale: [n_person][4] fp32 = (cx, cy, sx, sy) as mmpose `_box2cs` returns them.
 * crop_u8 (optional, may be NULL): [n_person][out_h][out_w][3] the warped u8 crop (parity tests).
 */
int pp_crop_affine_normalize(pp_ctx* ctx, const uint8_t* frames, int n_frames, int h, int w,
                             const int32_t* frame_idx, const double* bbox_tlwh, int n_person,
                             int out_w, int out_h, const float* lut, const int32_t* chan_map,
                             int flip, float* out, float* center_scale, uint8_t* crop_u8,
                             int32_t* valid, int mem);

/* ---- flip-merge + heatmap decode ------------------------------------------------------------
 * Replaces mmpose head `inference_model` flip_back/shift/average (test_cfg
 * hrnet_w48_coco_384x288_dark.py:81-85) and `keypoints_from_heatmaps` reached from
 * wrappers/mmpose.py:75; in-tree statement of the DARK maths: utils/inference.py:27-114.
 * hm: [n][k][h][w] fp32; hm_flip: same for the mirrored input or NULL (no flip test).
 * flip_perm: [k] channel permutation applied to hm_flip (COCO left/right pairs).
 * post: 0 = 'default' (+-0.25 px), 1 = 'unbiased' (DARK, blur_kernel e.g. 17),
 *       2 = UDP (`use_udp=True`, ViTPose configs; not in the reference tree): post_dark_udp (cv2.GaussianBlur with
 *       modulate kernel e.g. 11 and reflect-101 borders, clip [0.001, 50], log, edge-replicated stencil, Newton step
 *       with inv(Hessian + eps I)) and the UDP back-mapping x * scale * 200 / (w - 1).  In pp_topdown_* post 2 also
 *       selects the UDP crop (TopDownAffine(use_udp=True)); pp_crop_affine_normalize takes it as bit 1 of `flip`.
 * center_scale: [n][4] as above.  kpts: [n][k][3] fp32 = (x_px, y_px, maxval).
 * merged (optional): [n][k][h][w] the averaged heatmap (parity tests).
 */
int pp_flip_merge_decode(pp_ctx* ctx, const float* hm, const float* hm_flip, int n, int k, int h,
                         int w, const int32_t* flip_perm, int shift_heatmap, int post,
                         int blur_kernel, const float* center_scale, float* kpts, float* merged,
                         int mem);

/* ---- fused top-down 2D stage -------------------------------------------------------------------
 * The batched body of the per-frame loop at pose_pipeline/wrappers/mmpose.py:60-76: for every
 * (frame, bbox) pair crop + normalise (+ mirrored copy), run the backbone program `net`, flip-merge
 * and decode.  All parameters stay resident; one stream synchronisation per call.
 * flip_perm == NULL disables flip_test.  Absent persons (NaN bbox) produce zero rows
 * (wrappers/mmpose.py:67-69) and valid[i] = 0.
 */
int pp_topdown_create(pp_net* net, int in_buf, int out_buf, int num_joints, const int32_t* flip_perm,
                      int shift_heatmap, int post, int blur_kernel, const float* lut,
                      const int32_t* chan_map, pp_topdown** out);
void pp_topdown_destroy(pp_topdown* t);
/* frames [n_frames][h][w][3] u8 (host or device per frames_mem); kpts [n_person][K][3] fp32 */
int pp_topdown_run(pp_topdown* t, const uint8_t* frames, int n_frames, int h, int w, int frames_mem,
                   const int32_t* frame_idx, const double* bbox_tlwh, int n_person, float* kpts,
                   int kpts_mem, int32_t* valid);
/* BASELINE config 2: the person crops are already normalised tensors [n][in_h][in_w][4] fp32;
 * center_scale [n][4] is a host array. */
int pp_topdown_run_precropped(pp_topdown* t, const float* x_nhwc4, int x_mem, const float* center_scale,
                              int n_person, float* kpts, int kpts_mem);

/* HIP-event stage times of the last pp_topdown_run*: ms3 = {pre (copies, crop / mirror), backbone program,
 * flip-merge + decode}, measured on the ctx stream. */
int pp_topdown_timing(pp_topdown* t, float* ms3);

/* ---- NMS -------------------------------------------------------------------------------------
 * Replaces mmcv-full `nms` / `batched_nms` reached from the Faster-RCNN RPN and RoI head
 * (faster_rcnn_r50_fpn.py:101-109) -- convention 0: boxes [n][4] x1y1x2y2 float32, scores [n]
 * float32; suppress IoU > thr, area = w*h.
 * convention 1 = in-tree deep_sort preprocessing.non_max_suppression
 * (wrappers/deep_sort_yolov4/deep_sort/preprocessing.py:5-70): boxes [n][4] (x, y, w, h) FLOAT64,
 * scores [n] FLOAT64, +1 areas, overlap = inter / area_other.
 * convention 2 = tf.image.non_max_suppression as called by wrappers/deep_sort_yolov4/yolo4/model.py:278-281:
 * boxes [n][4] (y1, x1, y2, x2) float32 with corners in any order, IoU = inter / (Sa + Sb - inter), 0 when an
 * area is <= 0; the caller truncates to max_output_size.
 * keep[n] receives the surviving indices in descending score order, *n_keep their count; n <= 8192.
 */
int pp_nms(pp_ctx* ctx, const void* boxes, const void* scores, int n, double iou_thr, int convention,
           int32_t* keep, int32_t* n_keep, int mem);

/* ---- tracker (host) ---------------------------------------------------------------------------
 * Replaces the association stage of wrappers/mmtrack.py:45 (mmtrack SortTracker) with the one
 * tracker whose source is in the reference tree: wrappers/deep_sort_yolov4/deep_sort/
 * (tracker.py:10-131, track.py, kalman_filter.py:14-217, linear_assignment.py:14-186,
 * iou_matching.py:7-84).  float64 throughout, scipy-compatible Hungarian tie-breaking.
 * mode 0 = in-tree DeepSORT semantics without appearance features (IoU-only association);
 * mode 1 = mmtrack-style SORT (ids from 0, tentative handling as SURVEY.md A6; unpinned).
 */
/* mode 0: (feat_dim >= 1, max_iou_distance .7, max_cosine_distance .3, max_age 30, n_init 3) are the
 * defaults of tracker.py:40 / parser.py:35-47.  mode 1: max_iou_distance carries match_iou_thr (.5)
 * and max_cosine_distance carries obj_score_thr (.5); feat_dim, max_age, n_init are ignored. */
int pp_tracker_create(int mode, int feat_dim, double max_iou_distance, double max_cosine_distance,
                      int max_age, int n_init, pp_tracker** out);
void pp_tracker_destroy(pp_tracker* t);
/* One frame.  dets_tlwh: [n_det][4] float64 (x, y, w, h), conf: [n_det], feats: [n_det][feat_dim]
 * (mode 0 only, may be NULL in mode 1).  Outputs (capacity cap):
 *   mode 0: every live track after the update, as parser.py:76-86 emits them: track_id, tlwh (from the
 *           Kalman mean), info[4] = (state, hits, age, time_since_update);
 *   mode 1: boxes are the detector's (x1, y1, x2, y2) rows (float32 values widened to double); every kept
 *           detection (score > thr) comes back with its id: tlwh = the input row, info[1] = its index. */
int pp_tracker_step(pp_tracker* t, const double* dets_tlwh, const double* conf, const double* feats,
                    int n_det, int cap, int64_t* track_id, double* tlwh, int32_t* info, int32_t* n_out);
/* debugging / parity: dump all live tracks (id, state, hits, age, time_since_update, mean[8], cov[64]) */
int pp_tracker_dump(pp_tracker* t, int cap, int64_t* ids, int32_t* state4, double* mean8,
                    double* cov64, int32_t* n_out);
/* kalman_filter.py:31-217 entry points on their own (float64; mean[8], cov[8][8], z = (x, y, a, h)) */
int pp_kalman_initiate(const double* z4, double* mean8, double* cov64);
int pp_kalman_predict(double* mean8, double* cov64);
int pp_kalman_update(double* mean8, double* cov64, const double* z4);
int pp_kalman_gating_distance(const double* mean8, const double* cov64, const double* z4, int n,
                              double* out);
/* scipy.optimize.linear_sum_assignment restated (rectangular, float64); row4col/col4row sized
 * by n_rows / n_cols; returns assignment pairs sorted by row like scipy. */
int pp_linear_sum_assignment(const double* cost, int n_rows, int n_cols, int32_t* rows,
                             int32_t* cols, int32_t* n_pairs);

/* ---- detector ------------------------------------------------------------------------------------
 * Faster-RCNN R50-FPN person detector = the detection half of `mmtrack.apis.inference_mot`
 * (wrappers/mmtrack.py:45), batched over frames.  net_a: image program (input [Hp][Wp][4]; outputs:
 * 5 RPN objectness maps [H][W][3], 5 RPN delta maps [H][W][12], FPN P2..P5 [H][W][256]); net_b: RoI-head
 * program (input [7][7][256] per RoI; outputs cls [1][1][2], reg [1][1][4]; max_batch >= 1000 per frame).
 * bufs_a = {input, cls0..4, reg0..4, p2..p5} (15 ids), bufs_b = {roi_in, cls, reg}.  cls_l == reg_l (all levels) selects the fused
 * head layout: ONE map [H][W][16] per level, objectness in channels 0 - 2, deltas in 3 - 14 (rpn_cls and rpn_reg as one convolution).
 * lut: [3][256] fp32 = (v - mean[c]) * (1/std[c]) (mmcv imnormalize); channel c of the decoded BGR frame
 * feeds tensor channel c (the reference's double BGR<->RGB swap, wrappers/mmtrack.py:43 + to_rgb=True).
 * base_anchors: [5][3][4] fp32 (AnchorGenerator scales [8], ratios [.5,1,2], strides 4..64).
 * Test-time constants are those of faster_rcnn_r50_fpn.py:101-109.
 */
typedef struct pp_detector pp_detector;
/* mmcv rescale_size for img_scale (1088,1088) + Pad(size_divisor=32): resized and padded input dims */
int pp_detector_input_size(int src_h, int src_w, int32_t* nh, int32_t* nw, int32_t* hp, int32_t* wp);
/* The same mmcv test pipeline on its own (Resize keep_ratio -> Normalize -> Pad(size_divisor, pad_val)), used by the
 * YOLOX detector of the ByteTrack config (mot/bytetrack/ *.py: img_scale (800, 1440), mean 0 / std 1, pad 114):
 * pp_rescale_size = mmcv.rescale_size + the padded dims; pp_resize_pad_normalize writes device [n][hp][wp][4] fp32
 * (lut[c][v] applied to channel c of the BGR frame, 4th channel 0, padding = pad_val). */
int pp_rescale_size(int src_h, int src_w, int max_long, int max_short, int divisor, int32_t* nh, int32_t* nw,
                    int32_t* hp, int32_t* wp);
int pp_resize_pad_normalize(pp_ctx* ctx, const uint8_t* frames, int n, int src_h, int src_w, int frames_mem, int nh,
                            int nw, int hp, int wp, const float* lut, float pad_val, float* out_device);
/* The detector's test-time constants, for pinning against the vendored configs (faster_rcnn_r50_fpn.py:101-109, mot_challenge.py:33-47):
 * out[15] = {rpn nms_pre, rpn NMS IoU, rpn max_per_img, rcnn score_thr, rcnn NMS IoU, rcnn max_per_img, img_scale long, img_scale
 * short, Pad size_divisor, RoIAlign output_size, SingleRoIExtractor finest_scale, bbox_head target_stds[4]}.  Returns 15. */
int pp_detector_constants(double* out, int cap);
int pp_detector_create(pp_net* net_a, pp_net* net_b, const int32_t* bufs_a, const int32_t* bufs_b, int src_h,
                       int src_w, const float* lut, const float* base_anchors, pp_detector** out);
void pp_detector_destroy(pp_detector* d);
/* frames: [n_frames][src_h][src_w][3] u8 BGR (host or device per frames_mem; NULL = the program input
 * buffer was filled by the caller).  dets: host [n_frames][100][5] = (x1, y1, x2, y2, score) in source
 * pixels, n_dets: host [n_frames].  proposals / n_proposals (optional, host): [n_frames][1000][4], [n_frames]. */
int pp_detector_run(pp_detector* d, const uint8_t* frames, int n_frames, int frames_mem, float* dets,
                    int32_t* n_dets, float* proposals, int32_t* n_proposals);
/* HIP-event stage times of the last run, ms6 = {preprocess, image program, RPN proposals + NMS, RoIAlign,
 * RoI-head program, final decode + NMS} */
int pp_detector_timing(pp_detector* d, float* ms6);
/* The same pass in two halves (ABI 10): pp_detector_enqueue queues everything on the ctx stream -- resize, both programs, RPN / NMS /
 * RoIAlign, the copies of the results into the detector's page-locked staging -- and returns; pp_detector_collect waits for that pass
 * (an event: work queued BEHIND it on the stream is not waited for) and hands the results over.  One pass in flight per detector.
 * A caller that enqueues chunk k + 1 before it processes chunk k's detections on the host keeps the stream fed across its own host
 * work (tracker, bookkeeping) without a second stream or thread: posepipeline_amd/cascade.py.  Host frames (PP_MEM_HOST) must stay
 * valid until the collect.  pp_detector_run = enqueue + collect. */
int pp_detector_enqueue(pp_detector* d, const uint8_t* frames, int n_frames, int frames_mem, int want_proposals);
int pp_detector_collect(pp_detector* d, float* dets, int32_t* n_dets, float* proposals, int32_t* n_proposals);
/* Decision margins (round 6): how far every INTEGER decision of the detection path is from flipping, per frame.  The path takes
 * thousands of discrete decisions on float32 values (top-1000 per level, NMS 0.7, top-1000, RoI level, score > 0.05, NMS 0.5,
 * top-100: faster_rcnn_r50_fpn.py:101-109); an evaluation that is as accurate but not bit-identical -- this library's default
 * convolution numerics, or the reference's own cuDNN kernels -- may legitimately flip one that sits on its threshold.  With margins
 * enabled every pp_detector_run also returns, per frame, the distance of the CLOSEST decision of each class to its threshold, so the
 * caller knows which frames are decided (every margin above the evaluation's error) and which are not (re-run those on a
 * PP_NET_NUMERICS_EXACT detector: posepipeline_amd/cascade.py id_numerics="certified").  +inf = no decision of that class.
 *   PP_DET_MARGIN_RPN_CUT    score gap across the per-level top-k cut (min over the levels that have one)
 *   PP_DET_MARGIN_RPN_NMS    NMS 0.7, IoU units: min over kept proposals of (thr - IoU) to every kept predecessor, over suppressed
 *                            ones of their BEST suppressor's min(IoU - thr, score_weight * its score lead)
 *   PP_DET_MARGIN_RPN_TOP    score gap across the max_per_img cut of the kept proposals
 *   PP_DET_MARGIN_ROI_LEVEL  min |log2(sqrt(w h) / 56 + 1e-6) - b|, b in {1, 2, 3}, over the RoIs (which FPN level is sampled)
 *   PP_DET_MARGIN_SCORE_THR  min |score - 0.05| over the RoIs
 *   PP_DET_MARGIN_DET_NMS    NMS 0.5 on the scored boxes, as RPN_NMS
 *   PP_DET_MARGIN_DET_TOP    score gap across the top-100 cut
 *   PP_DET_MARGIN_DET_ORDER  smallest score gap between neighbouring OUTPUT rows (the tracker numbers new tracks in row order)
 * score_weight converts a score lead into IoU units inside the two NMS margins (a suppressor only counts while it stays ahead of
 * the box it suppresses): (IoU error bound) / (2 x score error bound) of the evaluation to be certified.  Costs ~1 ms per 64 frames;
 * off by default. */
#define PP_DET_N_MARGINS 8
#define PP_DET_MARGIN_RPN_CUT 0
#define PP_DET_MARGIN_RPN_NMS 1
#define PP_DET_MARGIN_RPN_TOP 2
#define PP_DET_MARGIN_ROI_LEVEL 3
#define PP_DET_MARGIN_SCORE_THR 4
#define PP_DET_MARGIN_DET_NMS 5
#define PP_DET_MARGIN_DET_TOP 6
#define PP_DET_MARGIN_DET_ORDER 7
int pp_detector_enable_margins(pp_detector* d, int enable, float score_weight);
/* margins [n_frames][PP_DET_N_MARGINS] of the most recent pp_detector_run (n_frames = that run's) */
int pp_detector_margins(pp_detector* d, int n_frames, float* margins);

/* ---- DeepSortYOLOv4 pre / post-processing ------------------------------------------------------------
 * tracking_method 0 (pipeline.py:519-523 -> wrappers/deep_sort_yolov4/parser.py:21): YOLOv4 detector + mars-small128
 * appearance encoder + the in-tree DeepSORT tracker (pp_tracker mode 0).  The two networks are layer programs
 * (posepipeline_amd/models/yolov4.py, mars.py); these entry points are the image and box arithmetic around them.
 *
 * pp_letterbox_bicubic: letterbox_image (yolo4/utils.py:21-32: PIL Image.resize BICUBIC = Pillow's 8-bit two-pass
 *   resampler, then paste on a (128,128,128) canvas) + yolo.py:93-95 float32 / 255, with the BGR->RGB swap of
 *   parser.py:55 folded in.  frames [n][src_h][src_w][3] u8 BGR; out: device [n][size_h][size_w][4] fp32 (R,G,B,0).
 *   xtab [nw][2+kx] / ytab [nh][2+ky] int32 (host): per output index (first source index, tap count, coefficients
 *   << 22) as Pillow's precompute_coeffs + normalize_coeffs_8bpc produce them (models/yolov4.py:pil_bicubic_table).
 * pp_yolo_decode: yolo_head + yolo_correct_boxes + box_confidence * class probability of ONE class
 *   (yolo4/model.py:193-254; yolo.py:112 keeps 'person' only).  feats: device [n][gh][gw][3*(5+num_classes)];
 *   boxes [n][gh*gw*3][4] (y1, x1, y2, x2 in image pixels), scores [n][gh*gw*3], host or device per out_mem.
 * pp_reid_patches: the cv2.resize(INTER_LINEAR) of extract_image_patch (tools/generate_detections.py:61-62) + the
 *   encoder graph's uint8->float cast and channel reversal (tools/freeze_model.py:239-255).  rects: host [n][5]
 *   int32 (frame, sx, sy, ex, ey) from the clipped box (:44-60, host side); out: device [n][ph][pw][4] fp32. */
int pp_letterbox_bicubic(pp_ctx* ctx, const uint8_t* frames, int n, int src_h, int src_w, int frames_mem,
                         const int32_t* xtab, int nw, int kx, const int32_t* ytab, int nh, int ky, int size_h,
                         int size_w, float* out_device);
int pp_yolo_decode(pp_ctx* ctx, const float* feats, int n, int gh, int gw, int num_classes, int cls,
                   const float* anchors3x2, int input_h, int input_w, int image_h, int image_w, float* boxes,
                   float* scores, int out_mem);
int pp_reid_patches(pp_ctx* ctx, const uint8_t* frames, int n_frames, int src_h, int src_w, int frames_mem,
                    const int32_t* rects, int n, int ph, int pw, float* out_device);

/* ---- 3D lifting -----------------------------------------------------------------------------
 * Replaces VideoPose3D TemporalModelOptimized1f + ChunkedGenerator windows reached from
 * wrappers/videopose3d.py:66-85, computed in the equivalent whole-clip dilated form.
 * net: a program built for the dilated form (posepipeline_amd.models.videopose3d).
 * in_buf is [1][T + 2*pad][>= in_features], out_buf [1][T][out_features]: the clip is processed in
 * chunks of T frames with a pad-frame halo, clamped to the clip (= np.pad mode 'edge').
 * kpts2d_norm: host [n_frames][in_features = 17*2] fp32 already screen-normalised
 * (videopose3d.py:26-37); out: host [n_frames][out_features = 17*3] fp32.
 */
int pp_videopose3d_lift(pp_net* net, int in_buf, int out_buf, const float* kpts2d_norm, int n_frames,
                        int in_features, int out_features, int pad, float* out);

#ifdef __cplusplus
}
#endif
#endif /* POSEPIPE_HIP_H */
