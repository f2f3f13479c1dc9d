// pp_videopose3d_lift: 2D -> 3D temporal lifting of one keypoint track in the whole-clip dilated form.
//
// Replaces the per-window loop of pose_pipeline/wrappers/videopose3d.py:66-85: ChunkedGenerator builds
// one edge-replicated 243-frame window per output frame and TemporalModelOptimized1f (strided convs)
// reduces each to one frame -- 10.4x redundant.  Here the clip is cut into chunks of T output frames;
// each chunk reads its frames plus a `pad`-frame halo (clamped to the clip = edge replication) and the
// dilated program (posepipeline_amd/models/videopose3d.py, dilations 1,3,9,27,81) produces all T
// frames at once.  Same weights, same taps per output in the same (tap, channel) order, hence
// bit-identical to the strided form (oracle/nets.py VideoPose3DRef).
#include "pp_internal.h"

int pp_net_dims(pp_net* net, int buf, int* h, int* w, int* c);
int pp_net_max_batch(pp_net* net);
pp_ctx* pp_net_ctx(pp_net* net);

extern "C" int pp_videopose3d_lift(pp_net* net, int in_buf, int out_buf, const float* kpts2d_norm, int n_frames,
                                   int in_features, int out_features, int pad, float* out) {
    PP_REQUIRE(net && kpts2d_norm && out, "pp_videopose3d_lift: NULL argument");
    PP_REQUIRE(n_frames >= 0 && in_features > 0 && out_features > 0 && pad >= 0, "pp_videopose3d_lift: bad dims");
    if (n_frames == 0) return PP_OK;
    int ih, iw, ic, oh, ow, oc;
    PP_REQUIRE(pp_net_dims(net, in_buf, &ih, &iw, &ic) == PP_OK && pp_net_dims(net, out_buf, &oh, &ow, &oc) == PP_OK,
               "pp_videopose3d_lift: bad buffer id");
    PP_REQUIRE(ih == 1 && oh == 1 && ic >= in_features && oc == out_features && iw == ow + 2 * pad,
               "pp_videopose3d_lift: program shape (in %dx%dx%d, out %dx%dx%d) does not match pad=%d / features %d->%d",
               ih, iw, ic, oh, ow, oc, pad, in_features, out_features);
    pp_ctx* ctx = pp_net_ctx(net);
    const int T = ow;
    const int max_b = pp_net_max_batch(net);
    const int n_chunks = (n_frames + T - 1) / T;
    std::vector<float> hin, hout;
    for (int c0 = 0; c0 < n_chunks; c0 += max_b) {
        const int b = std::min(max_b, n_chunks - c0);
        hin.assign((size_t)b * iw * ic, 0.f);
        for (int ci = 0; ci < b; ++ci) {
            const int t0 = (c0 + ci) * T;
            for (int j = 0; j < iw; ++j) {
                int src = t0 - pad + j;                       // np.pad(..., 'edge') == clamp
                src = src < 0 ? 0 : (src >= n_frames ? n_frames - 1 : src);
                memcpy(&hin[((size_t)ci * iw + j) * ic], kpts2d_norm + (size_t)src * in_features,
                       (size_t)in_features * sizeof(float));
            }
        }
        hout.resize((size_t)b * T * oc);
        int rc = pp_net_forward(net, b, in_buf, hin.data(), out_buf, hout.data(), PP_MEM_HOST);
        if (rc != PP_OK) return rc;
        for (int ci = 0; ci < b; ++ci) {
            const int t0 = (c0 + ci) * T;
            const int cnt = std::min(T, n_frames - t0);
            memcpy(out + (size_t)t0 * oc, &hout[(size_t)ci * T * oc], (size_t)cnt * oc * sizeof(float));
        }
    }
    (void)ctx;
    return PP_OK;
}
