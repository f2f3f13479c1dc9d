// Internal interfaces of the detector post-processing (det_post.hip <-> detector.hip <-> nms.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

// Test-time constants of the configuration the reference selects (wrappers/mmtrack.py:16-19 ->
// 3rdparty/mmtracking/mot/deepsort/deepsort_faster-rcnn_fpn_4e_mot17-private-half.py on _base_/models/faster_rcnn_r50_fpn.py:101-109
// and _base_/datasets/mot_challenge.py:33-47).  pp_detector_constants exports them; tests/test_arch_configs.py compares that with
// the values evaluated from those files (tests/golden/arch_configs.json).
constexpr int PP_DET_RPN_NMS_PRE = 1000, PP_DET_RPN_MAX_PER_IMG = 1000, PP_DET_RCNN_MAX_PER_IMG = 100;
constexpr float PP_DET_RPN_NMS_IOU = 0.7f, PP_DET_RCNN_SCORE_THR = 0.05f, PP_DET_RCNN_NMS_IOU = 0.5f;
constexpr int PP_DET_IMG_SCALE_LONG = 1088, PP_DET_IMG_SCALE_SHORT = 1088, PP_DET_SIZE_DIVISOR = 32;
constexpr int PP_DET_ROI_SIZE = 7;                 // RoIAlign output_size = bbox_head.roi_feat_size (the kernels are written for 7 x 7 bins)
constexpr float PP_DET_FINEST_SCALE = 56.f;        // SingleRoIExtractor.map_roi_levels (mmdet's default; not in the config file)
#define PP_DET_RCNN_STDS {0.1f, 0.1f, 0.2f, 0.2f}  /* bbox_head.bbox_coder.target_stds (means 0) */

struct DetRpnArgs {
    const float* cls[5];   // RPN objectness logits per level [frame][H][W][3] -- or, pitch 16: channels 0 - 2 of the fused head's map
    const float* reg[5];   // RPN deltas per level [frame][H][W][12] -- or cls + 3 of the fused map (channels 3 - 14)
    int pitch;             // 0: two maps (3 and 12 channels); 16: ONE [frame][H][W][16] map per level (round 4: rpn_cls and rpn_reg as
                           // one 15-channel convolution: the 256-channel input is read once instead of twice)
    int h[5], w[5], stride[5];
    float base[5][3][4];   // AnchorGenerator base anchors (float32, computed by the host like mmdet)
    int nms_pre;           // 1000
    int max_n;             // 5 * nms_pre rounded up: row stride of boxes / scores
    float* score_scratch;  // [frame][scratch_stride] sigmoid scores
    int scratch_stride;
    float* cand_box;       // [frame][5][nms_pre][4]
    float* cand_score;     // [frame][5][nms_pre]
    int32_t* cand_cnt;     // [frame][5]
    float* boxes;          // [frame][max_n][4]
    float* boxes_nms;      // [frame][max_n][4]  (+ level offsets)
    float* scores;         // [frame][max_n]
    int32_t* n_boxes;      // [frame]
    // decision margins (pp_detector_enable_margins; all null / 0 otherwise): per (frame, level) cut gaps, and where their per-frame
    // minimum goes (cut_margin[frame * margin_stride])
    float* cut_gap = nullptr;
    float* cut_margin = nullptr;
    int margin_stride = 0;
};

struct DetFpnArgs {
    const float* feat[4];  // P2..P5 [frame][H][W][C]
    int h[4], w[4], stride[4];
    int c;
};

int det_enqueue_preprocess(hipStream_t s, const uint8_t* frames, int n_frames, int H, int W, int nh, int nw, int Hp,
                           int Wp, const int32_t* xtab, const int32_t* ytab, const float* lut, float pad_val, float* out, unsigned* amax = nullptr);
int det_enqueue_rpn(hipStream_t s, const DetRpnArgs& a, int n_frames);
int det_enqueue_gather(hipStream_t s, const float* boxes, const float* scores, int max_n, const int32_t* keep,
                       const int32_t* n_keep, int limit, float* out_box, float* out_score, int32_t* n_out, int out5,
                       int n_frames, float* top_gap = nullptr, float* order_gap = nullptr, int margin_stride = 0);
// separable != 0: the default-numerics form (same samples, separable evaluation order); 0: the oracle's (iy, ix) order
// amax (may be null; separable kernel only): per-RoI maxima for the RoI head's fp16-form input (pp_net_input_amax)
bool det_roi_align_separable(int separable);
int det_enqueue_roi_align(hipStream_t s, const DetFpnArgs& a, const float* rois, const int32_t* n_rois, int max_rois,
                          float* out, int n_frames, int separable, unsigned* amax = nullptr);
int det_enqueue_final_decode(hipStream_t s, const float* rois, const int32_t* n_rois, int max_rois, const float* cls,
                             const float* reg, float sfx, float sfy, float score_thr, float* boxes, float* scores,
                             int32_t* n_out, int n_frames, float* level_margin = nullptr, float* thr_margin = nullptr,
                             int margin_stride = 0);

// nms.hip: batched float32 NMS (mmcv convention).  boxes [frame][max_n][4] (used for the IoU test), scores
// [frame][max_n], n [frame] on the device.  keep [frame][max_n] receives indices in descending score order.
// scratch: pp_nms_batched_scratch_bytes(max_n, n_frames).
size_t pp_nms_batched_scratch_bytes(int max_n, int n_frames);
// margin (may be null; decision margins): margin[frame * margin_stride] receives how far the frame's NMS result is from changing,
// in IoU units -- min over the kept boxes of (thr - IoU) to every kept predecessor, and over the suppressed boxes of their BEST
// suppressor's min(IoU - thr, score_weight * (score lead over the suppressed box)); +inf with fewer than two boxes.
// kflag_scratch: n_frames * max_n bytes.
int pp_enqueue_nms_batched(hipStream_t s, const float* boxes, const float* scores, const int32_t* n, int max_n,
                           int n_frames, float thr, void* scratch, int32_t* keep, int32_t* n_keep, float* margin = nullptr,
                           int margin_stride = 0, float score_weight = 0.f, unsigned char* kflag_scratch = nullptr);
