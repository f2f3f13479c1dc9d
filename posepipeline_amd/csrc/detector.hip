// pp_detector: Faster-RCNN R50-FPN person detector, frames in -> per-frame [<=100][5] boxes out, everything
// between on the device (two layer programs + det_post.hip + nms.hip), one synchronisation per call.
//
// The detection half of `mmtrack.apis.inference_mot` as called per frame at
// pose_pipeline/wrappers/mmtrack.py:37-45, batched over frames (detection is independent per frame; only
// the association that follows is sequential).  Model / test-time constants:
// 3rdparty/mmtracking/_base_/models/faster_rcnn_r50_fpn.py:101-109 (rpn: nms_pre 1000, NMS .7, max 1000;
// rcnn: score_thr .05, NMS .5, max 100) and _base_/datasets/mot_challenge.py:33-47 (scale (1088,1088), /32).
#include <cmath>
#include <memory>

#include "pp_internal.h"
#include "det_internal.h"
#include "pp_amax.h"

int pp_net_dims(pp_net* net, int buf, int* h, int* w, int* c);
int pp_net_max_batch(pp_net* net);
pp_ctx* pp_net_ctx(pp_net* net);

struct pp_detector {
    pp_ctx* ctx = nullptr;
    pp_net* netA = nullptr;   // image -> FPN levels + RPN maps
    pp_net* netB = nullptr;   // RoI features -> (cls, reg)
    int in_buf = 0, cls_buf[5], reg_buf[5], fpn_buf[4], roi_in = 0, roi_cls = 0, roi_reg = 0;
    int rpn_pitch = 0;       // 16: cls_buf[l] == reg_buf[l] is the fused head's 16-channel map
    unsigned* in_amax = nullptr;    // per-frame maxima of the image program's input (its fp16-form stem reads them), written by the preprocess kernel
    unsigned* roi_amax = nullptr;   // per-RoI maxima of the RoI head's input (its fp16-form fc6 reads them), written by RoIAlign
    int H = 0, W = 0, nh = 0, nw = 0, Hp = 0, Wp = 0;
    float sfx = 1.f, sfy = 1.f;
    int max_frames = 0, nms_pre = PP_DET_RPN_NMS_PRE, max_rois = PP_DET_RPN_MAX_PER_IMG, max_det = PP_DET_RCNN_MAX_PER_IMG, max_n = 0;
    float rpn_iou = PP_DET_RPN_NMS_IOU, score_thr = PP_DET_RCNN_SCORE_THR, det_iou = PP_DET_RCNN_NMS_IOU;
    int lvl_h[5], lvl_w[5], lvl_stride[5];
    float base[5][3][4];
    int scratch_stride = 0;
    // device state
    int32_t *d_xtab = nullptr, *d_ytab = nullptr;
    float* d_lut = nullptr;
    uint8_t* d_frames = nullptr;
    size_t d_frames_bytes = 0;
    char* d_work = nullptr;   // one allocation carved below
    float *score_scratch, *cand_box, *cand_score, *boxes, *boxes_nms, *scores, *rois, *roi_scores, *fin_boxes, *fin_scores, *out_dets;
    int32_t *cand_cnt, *n_boxes, *keep, *n_keep, *n_rois, *n_fin, *keep2, *n_keep2, *n_out;
    void *nms_scratch1, *nms_scratch2;
    // decision margins (pp_detector_enable_margins): [frame][PP_DET_N_MARGINS] on the device, the last run's copy on the host
    int margins_on = 0;
    float score_weight = 0.f;
    float *d_margins = nullptr, *cut_gap = nullptr;
    unsigned char *kflag1 = nullptr, *kflag2 = nullptr;
    std::vector<float> h_margins;
    int h_margins_frames = 0;
    // asynchronous form (pp_detector_enqueue / pp_detector_collect): page-locked staging of the pass's outputs and its completion event
    float *h_dets = nullptr, *h_props = nullptr;
    int32_t *h_ndets = nullptr, *h_nprops = nullptr;
    hipEvent_t ev_done = nullptr;
    int pending_frames = 0;      // > 0: a pass is in flight (enqueued, not collected)
    bool pending_props = false;
    float ms[6];
    hipEvent_t ev[7];
};

// mmcv.rescale_size: keep the aspect ratio so that the image fits `scale`
static void rescale_size(int w, int h, int max_long, int max_short, int* nw, int* nh) {
    const double sf = std::min((double)max_long / std::max(h, w), (double)max_short / std::min(h, w));
    *nw = (int)(w * sf + 0.5);
    *nh = (int)(h * sf + 0.5);
}

// cv::resize(INTER_LINEAR) 8-bit coefficient tables for one axis: (source index, w0, w1), weights * 2048
static void resize_table(int src, int dst, std::vector<int32_t>& tab) {
    tab.resize((size_t)dst * 3);
    const double scale = 1.0 / ((double)dst / src);
    for (int d = 0; d < dst; ++d) {
        float fx = (float)((d + 0.5) * scale - 0.5);
        int s = (int)std::floor(fx);
        fx -= (float)s;
        if (s < 0) { fx = 0.f; s = 0; }
        if (s >= src - 1) { fx = 0.f; s = src - 1; }
        tab[3 * d] = s;
        tab[3 * d + 1] = (int32_t)std::lrintf((1.f - fx) * 2048.f);
        tab[3 * d + 2] = (int32_t)std::lrintf(fx * 2048.f);
    }
}

extern "C" {

int pp_detector_input_size(int src_h, int src_w, int32_t* nh, int32_t* nw, int32_t* hp, int32_t* wp) {
    PP_REQUIRE(src_h > 0 && src_w > 0 && nh && nw && hp && wp, "pp_detector_input_size: bad argument");
    int a, b;
    rescale_size(src_w, src_h, PP_DET_IMG_SCALE_LONG, PP_DET_IMG_SCALE_SHORT, &a, &b);
    *nw = a; *nh = b;
    *wp = (a + PP_DET_SIZE_DIVISOR - 1) / PP_DET_SIZE_DIVISOR * PP_DET_SIZE_DIVISOR;
    *hp = (b + PP_DET_SIZE_DIVISOR - 1) / PP_DET_SIZE_DIVISOR * PP_DET_SIZE_DIVISOR;
    return PP_OK;
}

int pp_detector_constants(double* out, int cap) {
    const float stds[4] = PP_DET_RCNN_STDS;
    const double v[15] = {PP_DET_RPN_NMS_PRE, PP_DET_RPN_NMS_IOU, PP_DET_RPN_MAX_PER_IMG, PP_DET_RCNN_SCORE_THR, PP_DET_RCNN_NMS_IOU,
                          PP_DET_RCNN_MAX_PER_IMG, PP_DET_IMG_SCALE_LONG, PP_DET_IMG_SCALE_SHORT, PP_DET_SIZE_DIVISOR, PP_DET_ROI_SIZE,
                          PP_DET_FINEST_SCALE, stds[0], stds[1], stds[2], stds[3]};
    PP_REQUIRE(out && cap >= 15, "pp_detector_constants: out must hold 15 doubles");
    for (int i = 0; i < 15; ++i) out[i] = v[i];
    return 15;
}

int pp_rescale_size(int src_h, int src_w, int max_long, int max_short, int divisor, int32_t* nh, int32_t* nw, int32_t* hp,
                    int32_t* wp) {
    PP_REQUIRE(src_h > 0 && src_w > 0 && max_long > 0 && max_short > 0 && divisor > 0 && nh && nw && hp && wp,
               "pp_rescale_size: bad argument");
    int a, b;
    rescale_size(src_w, src_h, max_long, max_short, &a, &b);
    *nw = a; *nh = b;
    *wp = (a + divisor - 1) / divisor * divisor;
    *hp = (b + divisor - 1) / divisor * divisor;
    return PP_OK;
}

int pp_resize_pad_normalize(pp_ctx* ctx, const uint8_t* frames, int n, int src_h, int src_w, int frames_mem, int nh, int nw,
                            int hp, int wp, const float* lut, float pad_val, float* out_device) {
    PP_REQUIRE(ctx && frames && lut && out_device, "pp_resize_pad_normalize: NULL argument");
    PP_REQUIRE(n > 0 && src_h > 0 && src_w > 0 && nh > 0 && nw > 0 && hp >= nh && wp >= nw, "pp_resize_pad_normalize: bad dims");
    std::vector<int32_t> xt, yt;
    resize_table(src_w, nw, xt);
    resize_table(src_h, nh, yt);
    const size_t frame_bytes = (size_t)n * src_h * src_w * 3;
    size_t need = ScratchCursor::align(xt.size() * 4) + ScratchCursor::align(yt.size() * 4) + ScratchCursor::align(768 * 4);
    if (frames_mem == PP_MEM_HOST) need += ScratchCursor::align(frame_bytes);
    int rc = ctx->ensure_scratch(need);
    if (rc != PP_OK) return rc;
    ScratchCursor cur(ctx);
    int32_t* d_xt = cur.take<int32_t>(xt.size());
    int32_t* d_yt = cur.take<int32_t>(yt.size());
    float* d_lut = cur.take<float>(768);
    hipStream_t s = ctx->stream;
    const uint8_t* df = frames;
    if (frames_mem == PP_MEM_HOST) {
        uint8_t* st = cur.take<uint8_t>(frame_bytes);
        PP_HIP_CHECK(hipMemcpyAsync(st, frames, frame_bytes, hipMemcpyHostToDevice, s));
        df = st;
    }
    PP_HIP_CHECK(hipMemcpyAsync(d_xt, xt.data(), xt.size() * 4, hipMemcpyHostToDevice, s));
    PP_HIP_CHECK(hipMemcpyAsync(d_yt, yt.data(), yt.size() * 4, hipMemcpyHostToDevice, s));
    PP_HIP_CHECK(hipMemcpyAsync(d_lut, lut, 768 * 4, hipMemcpyHostToDevice, s));
    rc = det_enqueue_preprocess(s, df, n, src_h, src_w, nh, nw, hp, wp, d_xt, d_yt, d_lut, pad_val, out_device);
    if (rc != PP_OK) return rc;
    PP_HIP_CHECK(hipStreamSynchronize(s));    // the host tables must outlive the copies
    return PP_OK;
}

int pp_detector_create(pp_net* netA, pp_net* netB, const int32_t* bufs_a, const int32_t* bufs_b, int src_h, int src_w,
                       const float* lut, const float* base_anchors, pp_detector** out) {
    PP_REQUIRE(netA && netB && bufs_a && bufs_b && lut && base_anchors && out, "pp_detector_create: NULL argument");
    *out = nullptr;
    std::unique_ptr<pp_detector> d(new pp_detector());
    d->ctx = pp_net_ctx(netA);
    PP_REQUIRE(pp_net_ctx(netB) == d->ctx, "pp_detector_create: both programs must live on the same context");
    d->netA = netA; d->netB = netB;
    d->H = src_h; d->W = src_w;
    rescale_size(src_w, src_h, PP_DET_IMG_SCALE_LONG, PP_DET_IMG_SCALE_SHORT, &d->nw, &d->nh);
    d->Wp = (d->nw + PP_DET_SIZE_DIVISOR - 1) / PP_DET_SIZE_DIVISOR * PP_DET_SIZE_DIVISOR;
    d->Hp = (d->nh + PP_DET_SIZE_DIVISOR - 1) / PP_DET_SIZE_DIVISOR * PP_DET_SIZE_DIVISOR;
    d->sfx = (float)((double)d->nw / src_w);
    d->sfy = (float)((double)d->nh / src_h);
    d->in_buf = bufs_a[0];
    int h, w, c;
    PP_REQUIRE(pp_net_dims(netA, d->in_buf, &h, &w, &c) == PP_OK && h == d->Hp && w == d->Wp && c == 4,
               "pp_detector_create: program input is %dx%dx%d, expected %dx%dx4 for a %dx%d source", h, w, c, d->Hp, d->Wp,
               src_h, src_w);
    for (int l = 0; l < 5; ++l) {
        d->cls_buf[l] = bufs_a[1 + l];
        d->reg_buf[l] = bufs_a[6 + l];
        d->lvl_stride[l] = 4 << l;
        int ch, cw, cc, rh, rw, rc_;
        if (d->cls_buf[l] == d->reg_buf[l]) {
            // fused RPN head: ONE 16-channel map per level (objectness 0 - 2, deltas 3 - 14, channel 15 unused); all levels alike
            PP_REQUIRE(pp_net_dims(netA, d->cls_buf[l], &ch, &cw, &cc) == PP_OK && cc == 16, "fused rpn map of level %d must have 16 channels", l);
            PP_REQUIRE(l == 0 || d->rpn_pitch == 16, "pp_detector_create: fused and separate rpn maps mixed");
            d->rpn_pitch = 16;
        } else {
        PP_REQUIRE(d->rpn_pitch == 0, "pp_detector_create: fused and separate rpn maps mixed");
        PP_REQUIRE(pp_net_dims(netA, d->cls_buf[l], &ch, &cw, &cc) == PP_OK && cc == 3, "rpn_cls level %d must have 3 channels", l);
        PP_REQUIRE(pp_net_dims(netA, d->reg_buf[l], &rh, &rw, &rc_) == PP_OK && rc_ == 12 && rh == ch && rw == cw,
                   "rpn_reg level %d shape mismatch", l);
        }
        d->lvl_h[l] = ch; d->lvl_w[l] = cw;
        d->scratch_stride += ch * cw * 3;
    }
    for (int l = 0; l < 4; ++l) {
        d->fpn_buf[l] = bufs_a[11 + l];
        PP_REQUIRE(pp_net_dims(netA, d->fpn_buf[l], &h, &w, &c) == PP_OK && c == 256 && h == d->lvl_h[l] && w == d->lvl_w[l],
                   "FPN level %d shape mismatch", l);
    }
    d->roi_in = bufs_b[0]; d->roi_cls = bufs_b[1]; d->roi_reg = bufs_b[2];
    PP_REQUIRE(pp_net_dims(netB, d->roi_in, &h, &w, &c) == PP_OK && h == PP_DET_ROI_SIZE && w == PP_DET_ROI_SIZE && c == 256, "RoI head input must be 7x7x256");
    PP_REQUIRE(pp_net_dims(netB, d->roi_cls, &h, &w, &c) == PP_OK && h * w == 1 && c == 2, "RoI head cls output must be 1x1x2");
    PP_REQUIRE(pp_net_dims(netB, d->roi_reg, &h, &w, &c) == PP_OK && h * w == 1 && c == 4, "RoI head reg output must be 1x1x4");
    d->max_frames = std::min(pp_net_max_batch(netA), pp_net_max_batch(netB) / d->max_rois);
    // a fp16-form RoI head scales its input per RoI: the separable RoIAlign kernel writes the maxima itself (no extra pass)
    if (det_roi_align_separable(pp_net_numerics(netB) == PP_NET_NUMERICS_SPLIT)) {
        d->roi_amax = pp_net_input_amax_slot(netB, d->roi_in);       // (the promise itself is made per run, pp_detector_run)
    }
    // likewise the image program's input, when its stem runs in the fp16 form: the preprocess kernel folds the maxima
    d->in_amax = pp_net_input_amax_slot(netA, d->in_buf);
    PP_REQUIRE(d->max_frames > 0, "pp_detector_create: RoI-head program needs max_batch >= %d", d->max_rois);
    memcpy(d->base, base_anchors, sizeof(d->base));
    d->max_n = 5 * d->nms_pre;
    hipStream_t s = d->ctx->stream;
    PP_HIP_CHECK(hipSetDevice(d->ctx->device));
    std::vector<int32_t> xt, yt;
    resize_table(src_w, d->nw, xt);
    resize_table(src_h, d->nh, yt);
    PP_HIP_CHECK(hipMalloc((void**)&d->d_xtab, xt.size() * 4));
    PP_HIP_CHECK(hipMalloc((void**)&d->d_ytab, yt.size() * 4));
    PP_HIP_CHECK(hipMalloc((void**)&d->d_lut, 768 * 4));
    PP_HIP_CHECK(hipMemcpy(d->d_xtab, xt.data(), xt.size() * 4, hipMemcpyHostToDevice));
    PP_HIP_CHECK(hipMemcpy(d->d_ytab, yt.data(), yt.size() * 4, hipMemcpyHostToDevice));
    PP_HIP_CHECK(hipMemcpy(d->d_lut, lut, 768 * 4, hipMemcpyHostToDevice));
    // carve the work area
    const int F = d->max_frames;
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off += ScratchCursor::align(bytes); return o; };
    const size_t o_ss = carve((size_t)F * d->scratch_stride * 4), o_cb = carve((size_t)F * 5 * d->nms_pre * 16),
                 o_cs = carve((size_t)F * 5 * d->nms_pre * 4), o_cc = carve((size_t)F * 5 * 4),
                 o_bx = carve((size_t)F * d->max_n * 16), o_bn = carve((size_t)F * d->max_n * 16),
                 o_sc = carve((size_t)F * d->max_n * 4), o_nb = carve((size_t)F * 4), o_kp = carve((size_t)F * d->max_n * 4),
                 o_nk = carve((size_t)F * 4), o_ro = carve((size_t)F * d->max_rois * 16), o_rs = carve((size_t)F * d->max_rois * 4),
                 o_nr = carve((size_t)F * 4), o_fb = carve((size_t)F * d->max_rois * 16), o_fs = carve((size_t)F * d->max_rois * 4),
                 o_nf = carve((size_t)F * 4), o_k2 = carve((size_t)F * d->max_rois * 4), o_n2 = carve((size_t)F * 4),
                 o_od = carve((size_t)F * d->max_det * 20), o_no = carve((size_t)F * 4),
                 o_s1 = carve(pp_nms_batched_scratch_bytes(d->max_n, F)), o_s2 = carve(pp_nms_batched_scratch_bytes(d->max_rois, F)),
                 o_mg = carve((size_t)F * PP_DET_N_MARGINS * 4), o_cg = carve((size_t)F * 5 * 4), o_k1 = carve((size_t)F * d->max_n),
                 o_kf2 = carve((size_t)F * d->max_rois);
    PP_HIP_CHECK(hipMalloc((void**)&d->d_work, off));
    PP_HIP_CHECK(hipMemset(d->d_work, 0, off));
    char* base = d->d_work;
    d->score_scratch = (float*)(base + o_ss); d->cand_box = (float*)(base + o_cb); d->cand_score = (float*)(base + o_cs);
    d->cand_cnt = (int32_t*)(base + o_cc); d->boxes = (float*)(base + o_bx); d->boxes_nms = (float*)(base + o_bn);
    d->scores = (float*)(base + o_sc); d->n_boxes = (int32_t*)(base + o_nb); d->keep = (int32_t*)(base + o_kp);
    d->n_keep = (int32_t*)(base + o_nk); d->rois = (float*)(base + o_ro); d->roi_scores = (float*)(base + o_rs);
    d->n_rois = (int32_t*)(base + o_nr); d->fin_boxes = (float*)(base + o_fb); d->fin_scores = (float*)(base + o_fs);
    d->n_fin = (int32_t*)(base + o_nf); d->keep2 = (int32_t*)(base + o_k2); d->n_keep2 = (int32_t*)(base + o_n2);
    d->out_dets = (float*)(base + o_od); d->n_out = (int32_t*)(base + o_no);
    d->nms_scratch1 = base + o_s1; d->nms_scratch2 = base + o_s2;
    d->d_margins = (float*)(base + o_mg); d->cut_gap = (float*)(base + o_cg);
    d->kflag1 = (unsigned char*)(base + o_k1); d->kflag2 = (unsigned char*)(base + o_kf2);
    d->h_margins.assign((size_t)F * PP_DET_N_MARGINS, 0.f);
    for (auto& e : d->ev) PP_HIP_CHECK(hipEventCreate(&e));
    PP_HIP_CHECK(hipEventCreateWithFlags(&d->ev_done, hipEventDisableTiming));
    PP_HIP_CHECK(hipHostMalloc((void**)&d->h_dets, (size_t)F * d->max_det * 5 * sizeof(float)));
    PP_HIP_CHECK(hipHostMalloc((void**)&d->h_ndets, (size_t)F * sizeof(int32_t)));
    PP_HIP_CHECK(hipHostMalloc((void**)&d->h_props, (size_t)F * d->max_rois * 16));
    PP_HIP_CHECK(hipHostMalloc((void**)&d->h_nprops, (size_t)F * sizeof(int32_t)));
    PP_HIP_CHECK(hipStreamSynchronize(s));
    *out = d.release();
    return PP_OK;
}

void pp_detector_destroy(pp_detector* d) {
    if (!d) return;
    if (d->ctx && d->ctx->stream) (void)hipStreamSynchronize(d->ctx->stream);
    for (void* p : {(void*)d->d_xtab, (void*)d->d_ytab, (void*)d->d_lut, (void*)d->d_frames, (void*)d->d_work})
        if (p) (void)hipFree(p);
    for (auto& e : d->ev)
        if (e) (void)hipEventDestroy(e);
    if (d->ev_done) (void)hipEventDestroy(d->ev_done);
    for (void* p : {(void*)d->h_dets, (void*)d->h_ndets, (void*)d->h_props, (void*)d->h_nprops})
        if (p) (void)hipHostFree(p);
    delete d;
}

// frames == NULL: the preprocessed input is already in the program's input buffer (skip the resize)
int pp_detector_enqueue(pp_detector* d, const uint8_t* frames, int n_frames, int frames_mem, int want_proposals) {
    PP_REQUIRE(d, "pp_detector_enqueue: detector is NULL");
    PP_REQUIRE(n_frames > 0 && n_frames <= d->max_frames, "pp_detector_enqueue: %d frames not in (0, %d]", n_frames, d->max_frames);
    PP_REQUIRE(d->pending_frames == 0, "pp_detector_enqueue: the previous pass has not been collected");
    float* dets = d->h_dets;
    int32_t* n_dets = d->h_ndets;
    float* proposals = want_proposals ? d->h_props : nullptr;
    int32_t* n_proposals = want_proposals ? d->h_nprops : nullptr;
    hipStream_t s = d->ctx->stream;
    const int F = n_frames;
    PP_HIP_CHECK(hipEventRecord(d->ev[0], s));
    PpStages stage;
    stage.next("det.preprocess");
    void* in_ptr = nullptr;
    int rc = pp_net_buffer(d->netA, d->in_buf, &in_ptr, nullptr);
    if (rc != PP_OK) return rc;
    if (frames) {
        const uint8_t* df = frames;
        if (frames_mem == PP_MEM_HOST) {
            const size_t bytes = (size_t)F * d->H * d->W * 3;
            if (bytes > d->d_frames_bytes) {
                if (d->d_frames) PP_HIP_CHECK(hipFree(d->d_frames));
                d->d_frames = nullptr;
                PP_HIP_CHECK(hipMalloc((void**)&d->d_frames, bytes));
                d->d_frames_bytes = bytes;
            }
            PP_HIP_CHECK(hipMemcpyAsync(d->d_frames, frames, bytes, hipMemcpyHostToDevice, s));
            df = d->d_frames;
        }
        if (d->in_amax) PP_HIP_CHECK(hipMemsetAsync(d->in_amax, 0, (size_t)F * sizeof(unsigned), s));
        rc = det_enqueue_preprocess(s, df, F, d->H, d->W, d->nh, d->nw, d->Hp, d->Wp, d->d_xtab, d->d_ytab, d->d_lut, 0.f,
                                    static_cast<float*>(in_ptr), d->in_amax);
        if (rc != PP_OK) return rc;
    } else if (d->in_amax) {      // the caller filled the input buffer itself: take the maxima in a pass of their own
        PP_HIP_CHECK(hipMemsetAsync(d->in_amax, 0, (size_t)F * sizeof(unsigned), s));
        rc = pp_launch_amax(static_cast<const float*>(in_ptr), F, (size_t)d->Hp * d->Wp * 4, d->in_amax, s);
        if (rc != PP_OK) return rc;
    }
    PP_HIP_CHECK(hipEventRecord(d->ev[1], s));
    stage.next("det.image_program");
    if (d->in_amax) {            // the maxima were written above: the promise covers this run only (pp_net_input_amax)
        void* am = nullptr;
        rc = pp_net_input_amax(d->netA, d->in_buf, &am);
        if (rc != PP_OK) return rc;
    }
    rc = pp_net_run(d->netA, F, 0, -1);
    if (rc != PP_OK) return rc;
    PP_HIP_CHECK(hipEventRecord(d->ev[2], s));
    stage.next("det.rpn_proposals");
    DetRpnArgs ra{};
    for (int l = 0; l < 5; ++l) {
        void *pc, *pr;
        pp_net_buffer(d->netA, d->cls_buf[l], &pc, nullptr);
        pp_net_buffer(d->netA, d->reg_buf[l], &pr, nullptr);
        ra.pitch = d->rpn_pitch;
        ra.cls[l] = (const float*)pc; ra.reg[l] = d->rpn_pitch == 16 ? (const float*)pc + 3 : (const float*)pr;
        ra.h[l] = d->lvl_h[l]; ra.w[l] = d->lvl_w[l]; ra.stride[l] = d->lvl_stride[l];
        memcpy(ra.base[l], d->base[l], sizeof(ra.base[l]));
    }
    ra.nms_pre = d->nms_pre; ra.max_n = d->max_n; ra.score_scratch = d->score_scratch; ra.scratch_stride = d->scratch_stride;
    ra.cand_box = d->cand_box; ra.cand_score = d->cand_score; ra.cand_cnt = d->cand_cnt;
    ra.boxes = d->boxes; ra.boxes_nms = d->boxes_nms; ra.scores = d->scores; ra.n_boxes = d->n_boxes;
    // decision margins: one float per frame and decision class, written by the kernels that take the decisions (det_post.hip, nms.hip)
    const int MS = PP_DET_N_MARGINS;
    float* mg = d->margins_on ? d->d_margins : nullptr;
    auto mcol = [&](int k) { return mg ? mg + k : nullptr; };
    if (mg) {
        ra.cut_gap = d->cut_gap; ra.cut_margin = mcol(PP_DET_MARGIN_RPN_CUT); ra.margin_stride = MS;
    }
    rc = det_enqueue_rpn(s, ra, F);
    if (rc != PP_OK) return rc;
    rc = pp_enqueue_nms_batched(s, d->boxes_nms, d->scores, d->n_boxes, d->max_n, F, d->rpn_iou, d->nms_scratch1, d->keep, d->n_keep,
                                mcol(PP_DET_MARGIN_RPN_NMS), MS, d->score_weight, d->kflag1);
    if (rc != PP_OK) return rc;
    rc = det_enqueue_gather(s, d->boxes, d->scores, d->max_n, d->keep, d->n_keep, d->max_rois, d->rois, d->roi_scores, d->n_rois, 0, F,
                            mcol(PP_DET_MARGIN_RPN_TOP), nullptr, MS);
    if (rc != PP_OK) return rc;
    PP_HIP_CHECK(hipEventRecord(d->ev[3], s));
    stage.next("det.roi_align");
    DetFpnArgs fa{};
    for (int l = 0; l < 4; ++l) {
        void* p;
        pp_net_buffer(d->netA, d->fpn_buf[l], &p, nullptr);
        fa.feat[l] = (const float*)p; fa.h[l] = d->lvl_h[l]; fa.w[l] = d->lvl_w[l]; fa.stride[l] = d->lvl_stride[l];
    }
    fa.c = 256;
    void* roi_in_ptr;
    pp_net_buffer(d->netB, d->roi_in, &roi_in_ptr, nullptr);
    rc = det_enqueue_roi_align(s, fa, d->rois, d->n_rois, d->max_rois, (float*)roi_in_ptr, F,
                               pp_net_numerics(d->netB) == PP_NET_NUMERICS_SPLIT, d->roi_amax);
    if (rc != PP_OK) return rc;
    PP_HIP_CHECK(hipEventRecord(d->ev[4], s));
    stage.next("det.roi_head_program");
    if (d->roi_amax) {
        void* am = nullptr;
        rc = pp_net_input_amax(d->netB, d->roi_in, &am);
        if (rc != PP_OK) return rc;
    }
    rc = pp_net_run(d->netB, F * d->max_rois, 0, -1);
    if (rc != PP_OK) return rc;
    PP_HIP_CHECK(hipEventRecord(d->ev[5], s));
    stage.next("det.final_decode_nms");
    void *pcls, *preg;
    pp_net_buffer(d->netB, d->roi_cls, &pcls, nullptr);
    pp_net_buffer(d->netB, d->roi_reg, &preg, nullptr);
    rc = det_enqueue_final_decode(s, d->rois, d->n_rois, d->max_rois, (const float*)pcls, (const float*)preg, d->sfx, d->sfy,
                                  d->score_thr, d->fin_boxes, d->fin_scores, d->n_fin, F, mcol(PP_DET_MARGIN_ROI_LEVEL),
                                  mcol(PP_DET_MARGIN_SCORE_THR), MS);
    if (rc != PP_OK) return rc;
    rc = pp_enqueue_nms_batched(s, d->fin_boxes, d->fin_scores, d->n_fin, d->max_rois, F, d->det_iou, d->nms_scratch2, d->keep2, d->n_keep2,
                                mcol(PP_DET_MARGIN_DET_NMS), MS, d->score_weight, d->kflag2);
    if (rc != PP_OK) return rc;
    rc = det_enqueue_gather(s, d->fin_boxes, d->fin_scores, d->max_rois, d->keep2, d->n_keep2, d->max_det, d->out_dets, nullptr, d->n_out, 1, F,
                            mcol(PP_DET_MARGIN_DET_TOP), mcol(PP_DET_MARGIN_DET_ORDER), MS);
    if (rc != PP_OK) return rc;
    d->h_margins_frames = 0;
    if (mg) {
        PP_HIP_CHECK(hipMemcpyAsync(d->h_margins.data(), mg, (size_t)F * MS * sizeof(float), hipMemcpyDeviceToHost, s));
        d->h_margins_frames = F;
    }
    PP_HIP_CHECK(hipEventRecord(d->ev[6], s));
    PP_HIP_CHECK(hipMemcpyAsync(dets, d->out_dets, (size_t)F * d->max_det * 5 * sizeof(float), hipMemcpyDeviceToHost, s));
    PP_HIP_CHECK(hipMemcpyAsync(n_dets, d->n_out, (size_t)F * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    if (proposals) PP_HIP_CHECK(hipMemcpyAsync(proposals, d->rois, (size_t)F * d->max_rois * 16, hipMemcpyDeviceToHost, s));
    if (n_proposals) PP_HIP_CHECK(hipMemcpyAsync(n_proposals, d->n_rois, (size_t)F * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    PP_HIP_CHECK(hipEventRecord(d->ev_done, s));
    d->pending_frames = F;
    d->pending_props = want_proposals != 0;
    return PP_OK;
}

int pp_detector_collect(pp_detector* d, float* dets, int32_t* n_dets, float* proposals, int32_t* n_proposals) {
    PP_REQUIRE(d && dets && n_dets, "pp_detector_collect: NULL argument");
    PP_REQUIRE(d->pending_frames > 0, "pp_detector_collect: no pass in flight");
    PP_REQUIRE(!(proposals || n_proposals) || d->pending_props, "pp_detector_collect: the pass was enqueued without proposals");
    const int F = d->pending_frames;
    PP_HIP_CHECK(hipEventSynchronize(d->ev_done));
    d->pending_frames = 0;
    memcpy(dets, d->h_dets, (size_t)F * d->max_det * 5 * sizeof(float));
    memcpy(n_dets, d->h_ndets, (size_t)F * sizeof(int32_t));
    if (proposals) memcpy(proposals, d->h_props, (size_t)F * d->max_rois * 16);
    if (n_proposals) memcpy(n_proposals, d->h_nprops, (size_t)F * sizeof(int32_t));
    for (int i = 0; i < 6; ++i) (void)hipEventElapsedTime(&d->ms[i], d->ev[i], d->ev[i + 1]);
    return PP_OK;
}

int pp_detector_run(pp_detector* d, const uint8_t* frames, int n_frames, int frames_mem, float* dets, int32_t* n_dets,
                    float* proposals, int32_t* n_proposals) {
    PP_REQUIRE(d && dets && n_dets, "pp_detector_run: NULL argument");
    PP_REQUIRE(n_frames >= 0 && n_frames <= d->max_frames, "pp_detector_run: %d frames exceed capacity %d", n_frames, d->max_frames);
    if (n_frames == 0) return PP_OK;
    int rc = pp_detector_enqueue(d, frames, n_frames, frames_mem, proposals || n_proposals);
    if (rc != PP_OK) return rc;
    return pp_detector_collect(d, dets, n_dets, proposals, n_proposals);
}

int pp_detector_enable_margins(pp_detector* d, int enable, float score_weight) {
    PP_REQUIRE(d, "pp_detector_enable_margins: detector is NULL");
    PP_REQUIRE(!enable || score_weight > 0.f, "pp_detector_enable_margins: score_weight must be positive");
    PP_HIP_CHECK(hipStreamSynchronize(d->ctx->stream));
    d->margins_on = enable != 0;
    d->score_weight = score_weight;
    d->h_margins_frames = 0;
    return PP_OK;
}

int pp_detector_margins(pp_detector* d, int n_frames, float* margins) {
    PP_REQUIRE(d && margins, "pp_detector_margins: NULL argument");
    PP_REQUIRE(d->margins_on, "pp_detector_margins: margins are not enabled (pp_detector_enable_margins)");
    PP_REQUIRE(n_frames == d->h_margins_frames, "pp_detector_margins: the last run had %d frames, %d asked for", d->h_margins_frames, n_frames);
    memcpy(margins, d->h_margins.data(), (size_t)n_frames * PP_DET_N_MARGINS * sizeof(float));   // (pp_detector_run synchronised after the copy)
    return PP_OK;
}

int pp_detector_timing(pp_detector* d, float* ms6) {
    PP_REQUIRE(d && ms6, "pp_detector_timing: NULL argument");
    memcpy(ms6, d->ms, sizeof(d->ms));
    return PP_OK;
}

}  // extern "C"
