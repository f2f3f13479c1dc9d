// pp_topdown: the fused top-down 2D stage -- crop/normalise (+ mirrored copy) -> backbone program ->
// flip-merge + decode -- with every parameter resident on the device and one synchronisation per
// call.  This is the per-frame body of pose_pipeline/wrappers/mmpose.py:60-76 (one
// `inference_top_down_pose_model` call per person-frame, batch 1) restructured as one batched pass.
#include <memory>

#include "pp_internal.h"

struct pp_topdown {
    pp_ctx* ctx = nullptr;
    pp_net* net = nullptr;
    int in_buf = 0, out_buf = 0;
    int in_h = 0, in_w = 0, hm_h = 0, hm_w = 0, k = 0;
    int flip = 0, shift_heatmap = 0, post = 0, blur_kernel = 0;
    int32_t chan_map[3] = {0, 1, 2};
    int max_person = 0;   // net max_batch / (flip ? 2 : 1)
    float* d_lut = nullptr;
    int32_t* d_perm = nullptr;
    PersonXform* d_xf = nullptr;
    float* d_cs = nullptr;
    float* d_kpts = nullptr;
    uint8_t* d_frames = nullptr;   // staging for host frames
    size_t d_frames_bytes = 0;
    // pinned host staging
    PersonXform* h_xf = nullptr;
    float* h_cs = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};   // start | after pre | after net | after decode
};

// implemented in api_core.hip
int pp_net_dims(pp_net* net, int buf, int* h, int* w, int* c);
int pp_net_max_batch(pp_net* net);
pp_ctx* pp_net_ctx(pp_net* net);

extern "C" void pp_topdown_destroy(pp_topdown* t);

extern "C" {

int pp_topdown_create(pp_net* net, int in_buf, int out_buf, int num_joints, const int32_t* flip_perm,
                      int shift_heatmap, int post, int blur_kernel, const float* lut, const int32_t* chan_map,
                      pp_topdown** out) {
    PP_REQUIRE(net && lut && chan_map && out, "pp_topdown_create: NULL argument");
    *out = nullptr;
    // every early return below (PP_REQUIRE / PP_HIP_CHECK) releases what has been allocated so far
    std::unique_ptr<pp_topdown, void (*)(pp_topdown*)> t(new pp_topdown(), pp_topdown_destroy);
    t->net = net;
    t->ctx = pp_net_ctx(net);
    int c = 0, oc = 0;
    PP_REQUIRE(pp_net_dims(net, in_buf, &t->in_h, &t->in_w, &c) == PP_OK && c == 4,
               "pp_topdown_create: input buffer must be NHWC with 4 channels");
    // the output buffer is declared (h, w, K) and written as [K][h][w] planes
    PP_REQUIRE(pp_net_dims(net, out_buf, &t->hm_h, &t->hm_w, &oc) == PP_OK && oc == num_joints,
               "pp_topdown_create: output buffer has %d channels, expected %d", oc, num_joints);
    t->in_buf = in_buf; t->out_buf = out_buf; t->k = num_joints;
    t->flip = flip_perm != nullptr; t->shift_heatmap = shift_heatmap; t->post = post; t->blur_kernel = blur_kernel;
    for (int i = 0; i < 3; ++i) {
        PP_REQUIRE(chan_map[i] >= 0 && chan_map[i] < 3, "chan_map[%d] out of range", i);
        t->chan_map[i] = chan_map[i];
    }
    t->max_person = pp_net_max_batch(net) / (t->flip ? 2 : 1);
    PP_REQUIRE(t->max_person > 0, "pp_topdown_create: net max_batch too small");
    hipStream_t s = t->ctx->stream;
    PP_HIP_CHECK(hipSetDevice(t->ctx->device));
    PP_HIP_CHECK(hipMalloc((void**)&t->d_lut, 768 * sizeof(float)));
    PP_HIP_CHECK(hipMemcpyAsync(t->d_lut, lut, 768 * sizeof(float), hipMemcpyHostToDevice, s));
    if (flip_perm) {
        for (int i = 0; i < num_joints; ++i)
            PP_REQUIRE(flip_perm[i] >= 0 && flip_perm[i] < num_joints, "flip_perm[%d] out of range", i);
        PP_HIP_CHECK(hipMalloc((void**)&t->d_perm, num_joints * sizeof(int32_t)));
        PP_HIP_CHECK(hipMemcpyAsync(t->d_perm, flip_perm, num_joints * sizeof(int32_t), hipMemcpyHostToDevice, s));
    }
    PP_HIP_CHECK(hipMalloc((void**)&t->d_xf, t->max_person * sizeof(PersonXform)));
    PP_HIP_CHECK(hipMalloc((void**)&t->d_cs, t->max_person * 4 * sizeof(float)));
    PP_HIP_CHECK(hipMalloc((void**)&t->d_kpts, (size_t)t->max_person * num_joints * 3 * sizeof(float)));
    PP_HIP_CHECK(hipHostMalloc((void**)&t->h_xf, t->max_person * sizeof(PersonXform), hipHostMallocDefault));
    PP_HIP_CHECK(hipHostMalloc((void**)&t->h_cs, t->max_person * 4 * sizeof(float), hipHostMallocDefault));
    for (auto& e : t->ev) PP_HIP_CHECK(hipEventCreate(&e));
    PP_HIP_CHECK(hipStreamSynchronize(s));
    *out = t.release();
    return PP_OK;
}

void pp_topdown_destroy(pp_topdown* t) {
    if (!t) return;
    if (t->ctx && t->ctx->stream) (void)hipStreamSynchronize(t->ctx->stream);
    if (t->d_lut) (void)hipFree(t->d_lut);
    if (t->d_perm) (void)hipFree(t->d_perm);
    if (t->d_xf) (void)hipFree(t->d_xf);
    if (t->d_cs) (void)hipFree(t->d_cs);
    if (t->d_kpts) (void)hipFree(t->d_kpts);
    if (t->d_frames) (void)hipFree(t->d_frames);
    if (t->h_xf) (void)hipHostFree(t->h_xf);
    if (t->h_cs) (void)hipHostFree(t->h_cs);
    for (auto& e : t->ev)
        if (e) (void)hipEventDestroy(e);
    delete t;
}

// check_valid: zero the rows of absent persons (h_xf[i].valid == 0) ON THE DEVICE, so that host and device outputs both
// keep the reference's zeros((K, 3)) contract (wrappers/mmpose.py:67-69)
static int topdown_tail(pp_topdown* t, int n_person, float* kpts, int kpts_mem, bool check_valid) {
    hipStream_t s = t->ctx->stream;
    const int batch = n_person * (t->flip ? 2 : 1);
    PP_HIP_CHECK(hipEventRecord(t->ev[1], s));
    PpStages stage;
    stage.next("topdown.backbone_program");
    int rc = pp_net_run(t->net, batch, 0, -1);
    if (rc != PP_OK) return rc;
    PP_HIP_CHECK(hipEventRecord(t->ev[2], s));
    stage.next("topdown.flip_merge_decode");
    void* hm_ptr = nullptr;
    size_t hm_bytes = 0;
    rc = pp_net_buffer(t->net, t->out_buf, &hm_ptr, &hm_bytes);
    if (rc != PP_OK) return rc;
    const float* hm = static_cast<const float*>(hm_ptr);
    const float* hm_flip = t->flip ? hm + (size_t)n_person * (hm_bytes / sizeof(float)) : nullptr;
    const DecodeParams dp{n_person, t->k, t->hm_h, t->hm_w, t->shift_heatmap, t->post, t->blur_kernel};
    float* dk = kpts_mem == PP_MEM_DEVICE ? kpts : t->d_kpts;
    rc = pp_enqueue_decode(s, dp, hm, hm_flip, t->d_perm, t->d_cs, dk, nullptr);
    if (rc != PP_OK) return rc;
    PP_HIP_CHECK(hipEventRecord(t->ev[3], s));
    if (check_valid) {
        const size_t row = (size_t)t->k * 3;
        for (int i = 0; i < n_person; ++i)
            if (!t->h_xf[i].valid) PP_HIP_CHECK(hipMemsetAsync(dk + (size_t)i * row, 0, row * sizeof(float), s));
    }
    if (kpts_mem == PP_MEM_HOST) {
        PP_HIP_CHECK(hipMemcpyAsync(kpts, dk, (size_t)n_person * t->k * 3 * sizeof(float), hipMemcpyDeviceToHost, s));
    }
    PP_HIP_CHECK(hipStreamSynchronize(s));
    return PP_OK;
}

int pp_topdown_run(pp_topdown* t, const uint8_t* frames, int n_frames, int h, int w, int frames_mem,
                   const int32_t* frame_idx, const double* bbox_tlwh, int n_person, float* kpts, int kpts_mem,
                   int32_t* valid) {
    PP_REQUIRE(t && frames && frame_idx && bbox_tlwh && kpts, "pp_topdown_run: NULL argument");
    PP_REQUIRE(n_person >= 0 && n_person <= t->max_person, "pp_topdown_run: n_person %d exceeds capacity %d", n_person,
               t->max_person);
    if (n_person == 0) return PP_OK;
    hipStream_t s = t->ctx->stream;
    for (int i = 0; i < n_person; ++i) {
        PP_REQUIRE(frame_idx[i] >= 0 && frame_idx[i] < n_frames, "frame_idx[%d]=%d out of range", i, frame_idx[i]);
        pp_person_transform(bbox_tlwh + 4 * i, t->in_w, t->in_h, t->h_cs + 4 * i, &t->h_xf[i], t->post == 2);
        t->h_xf[i].frame = frame_idx[i];
        if (valid) valid[i] = t->h_xf[i].valid;
    }
    PP_HIP_CHECK(hipEventRecord(t->ev[0], s));
    PP_HIP_CHECK(hipMemcpyAsync(t->d_xf, t->h_xf, n_person * sizeof(PersonXform), hipMemcpyHostToDevice, s));
    PP_HIP_CHECK(hipMemcpyAsync(t->d_cs, t->h_cs, n_person * 4 * sizeof(float), hipMemcpyHostToDevice, s));
    const uint8_t* dframes = frames;
    if (frames_mem == PP_MEM_HOST) {
        const size_t bytes = (size_t)n_frames * h * w * 3;
        if (bytes > t->d_frames_bytes) {
            if (t->d_frames) PP_HIP_CHECK(hipFree(t->d_frames));
            t->d_frames = nullptr;
            PP_HIP_CHECK(hipMalloc((void**)&t->d_frames, bytes));
            t->d_frames_bytes = bytes;
        }
        PP_HIP_CHECK(hipMemcpyAsync(t->d_frames, frames, bytes, hipMemcpyHostToDevice, s));
        dframes = t->d_frames;
    }
    void* in_ptr = nullptr;
    int rc = pp_net_buffer(t->net, t->in_buf, &in_ptr, nullptr);
    if (rc != PP_OK) return rc;
    rc = pp_enqueue_crop(s, dframes, h, w, t->d_xf, n_person, t->in_w, t->in_h, t->d_lut, t->chan_map, t->flip,
                         static_cast<float*>(in_ptr), nullptr);
    if (rc != PP_OK) return rc;
    return topdown_tail(t, n_person, kpts, kpts_mem, true);
}

int pp_topdown_run_precropped(pp_topdown* t, const float* x_nhwc4, int x_mem, const float* center_scale,
                              int n_person, float* kpts, int kpts_mem) {
    PP_REQUIRE(t && x_nhwc4 && center_scale && kpts, "pp_topdown_run_precropped: NULL argument");
    PP_REQUIRE(n_person >= 0 && n_person <= t->max_person, "pp_topdown_run_precropped: n_person %d exceeds capacity %d",
               n_person, t->max_person);
    if (n_person == 0) return PP_OK;
    hipStream_t s = t->ctx->stream;
    void* in_ptr = nullptr;
    size_t in_bytes = 0;
    int rc = pp_net_buffer(t->net, t->in_buf, &in_ptr, &in_bytes);
    if (rc != PP_OK) return rc;
    float* din = static_cast<float*>(in_ptr);
    PP_HIP_CHECK(hipEventRecord(t->ev[0], s));
    if (x_nhwc4 != din) {
        PP_HIP_CHECK(hipMemcpyAsync(din, x_nhwc4, (size_t)n_person * in_bytes,
                                    x_mem == PP_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, s));
    }
    memcpy(t->h_cs, center_scale, (size_t)n_person * 4 * sizeof(float));
    PP_HIP_CHECK(hipMemcpyAsync(t->d_cs, t->h_cs, n_person * 4 * sizeof(float), hipMemcpyHostToDevice, s));
    if (t->flip) {
        rc = pp_enqueue_flip_w(s, din, din + (size_t)n_person * (in_bytes / sizeof(float)), n_person, t->in_h, t->in_w);
        if (rc != PP_OK) return rc;
    }
    return topdown_tail(t, n_person, kpts, kpts_mem, false);
}

int pp_topdown_timing(pp_topdown* t, float* ms3) {
    PP_REQUIRE(t && ms3, "pp_topdown_timing: NULL argument");
    for (int i = 0; i < 3; ++i) PP_HIP_CHECK(hipEventElapsedTime(&ms3[i], t->ev[i], t->ev[i + 1]));
    return PP_OK;
}

}  // extern "C"
