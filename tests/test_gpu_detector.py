"""GPU detector (pp_detector) vs the CPU oracle (oracle/detector.py), stage by stage and end to end."""
import numpy as np
import pytest

from oracle import detector as odet
from posepipeline_amd.models import faster_rcnn as fr
from posepipeline_amd.models import synth

pytestmark = pytest.mark.gpu


def synth_frame(rng, h, w):
    base = rng.integers(0, 256, (h // 6 + 1, w // 6 + 1, 3)).astype(np.uint8)
    img = np.repeat(np.repeat(base, 6, axis=0), 6, axis=1)[:h, :w].astype(np.int64)
    img[h // 4: 3 * h // 4, w // 3: w // 2] = rng.integers(100, 255, (3 * h // 4 - h // 4, w // 2 - w // 3, 3))
    return np.clip(img + rng.integers(-12, 13, img.shape), 0, 255).astype(np.uint8)


@pytest.fixture(scope="module")
def setup(ctx):
    rng = np.random.default_rng(2)                      # config index 2
    sd = synth.synth_state_dict(fr.faster_rcnn_param_shapes(), seed=2)
    # He-normal heads give saturated scores and e^4-sized boxes; tame them so that sorting, the
    # wh-ratio clamp, NMS and the score threshold all see a spread of values
    for k, g in (("detector.rpn_head.rpn_cls.weight", 0.5), ("detector.rpn_head.rpn_reg.weight", 0.1),
                 ("detector.roi_head.bbox_head.fc_reg.weight", 0.2)):
        sd[k] = (sd[k] * g).astype(np.float32)
    frames = np.stack([synth_frame(rng, 135, 240), synth_frame(rng, 135, 240)])
    # numerics named explicitly: a module-scoped fixture is created BEFORE conftest's function-scoped numerics fixture is entered, so the
    # process default at that moment depends on which module ran before (this module run on its own got the split kernels)
    det = fr.Detector(ctx, sd, 135, 240, max_frames=2, numerics="exact")
    assert det.net_a.numerics == "exact" and det.net_b.numerics == "exact"
    return sd, frames, det


def test_input_size_rule():
    assert fr.detector_input_size(1080, 1920) == (612, 1088, 640, 1088)
    assert fr.detector_input_size(480, 640) == (816, 1088, 832, 1088)
    assert odet.rescale_size(1920, 1080) == (1088, 612) and odet.rescale_size(640, 480) == (1088, 816)
    assert np.array_equal(fr.base_anchors()[2], odet.base_anchors(16))
    assert np.array_equal(fr.normalize_lut(), odet.normalize_lut())


def test_detector_matches_oracle(setup):
    sd, frames, det = setup
    dets, props = det.run(frames, want_proposals=True)
    model = odet.FasterRCNNRef(sd)
    for f in range(frames.shape[0]):
        ref, mid = odet.detect(model, frames[f][:, :, ::-1], want_intermediates=True)   # the wrapper hands mmtrack RGB
        # 1. resize + normalise + pad: bit-exact
        x = det.net_a.read("input", frames.shape[0])[f]
        assert np.array_equal(x[:, :, :3], mid["x"]) and not x[:, :, 3].any()
        # 2. backbone / FPN / RPN maps: bit-exact (fp32 MFMA == fmaf chain)
        for l in range(5):
            rpn = det.net_a.read(f"rpn{l}", 2)[f]                     # fused head: channels 0 - 2 objectness, 3 - 14 deltas
            assert rpn.shape[-1] == 16
            assert np.array_equal(rpn[..., :3], mid["cls_maps"][l][0]), f"rpn_cls level {l}"
            assert np.array_equal(rpn[..., 3:15], mid["reg_maps"][l][0]), f"rpn_reg level {l}"
        assert np.array_equal(det.net_a.read("p2", 2)[f], mid["feats"][0][0])
        # 3. proposals: same boxes in the same order
        assert props[f].shape == mid["proposals"].shape, (props[f].shape, mid["proposals"].shape)
        assert np.array_equal(props[f], mid["proposals"])
        # 4. final detections
        assert dets[f].shape == ref.shape, (dets[f].shape, ref.shape)
        assert ref.shape[0] > 0
        assert np.array_equal(dets[f], ref), np.abs(dets[f] - ref).max()


def test_roi_align_and_head_match_oracle(setup):
    sd, frames, det = setup
    dets, props = det.run(frames[:1], want_proposals=True)
    model = odet.FasterRCNNRef(sd)
    feats = [det.net_a.read(f"p{i}", 1) for i in range(2, 6)]
    n = min(40, props[0].shape[0])
    ref_feats, _ = odet.extract_roi_feats(feats, props[0][:n])
    got = det.net_b.read("roi_in", n)
    assert np.array_equal(got, ref_feats), np.abs(got - ref_feats).max()
    cls_ref, reg_ref = model.roi_head(ref_feats)
    assert np.array_equal(det.net_b.read("cls", n).reshape(n, 2), cls_ref)
    assert np.array_equal(det.net_b.read("reg", n).reshape(n, 4), reg_ref)


def test_direct_runs_of_the_detector_nets_take_their_own_input_maxima(ctx, setup):
    """pp_net_input_amax is a ONE-SHOT promise: after pp_detector_create and after a detector run, a direct forward /
    run / profile of net_a or net_b must scale its fp16-form stem / fc6 by the maxima of the input it is GIVEN -- a stale (or zero)
    maximum would push x * s out of the float16 range and the outputs to inf / NaN."""
    sd, frames, _ = setup
    det = fr.Detector(ctx, sd, 135, 240, max_frames=2, numerics="split")
    if det.net_a.split_kind != "split_f16":
        pytest.skip("the per-sample activation scale belongs to the fp16 form")
    rng = np.random.default_rng(5)
    h, w, c = det.prog_a.bufs[det.prog_a.named["input"]]
    x_small = (rng.standard_normal((2, h, w, c)) * 0.01).astype(np.float32)
    x_big = (rng.standard_normal((2, h, w, c)) * 300.0).astype(np.float32)
    x_small[..., 3] = 0
    x_big[..., 3] = 0
    # straight after creation (nothing has ever filled the maxima)
    y0 = det.net_a.forward(x_big, out_name="rpn0")
    assert np.isfinite(y0).all()
    # after a detector run whose frames have maxima ~2.6 (normalised pixels), then inputs 100x larger, by forward and by run
    det.run(frames)
    y1 = det.net_a.forward(x_big, out_name="rpn0")
    assert np.array_equal(y0, y1)
    det.run(frames)
    dptr, _, _ = det.net_a.buffer("input")
    ctx.h2d(dptr, x_big)
    det.net_a.run(2)
    assert np.array_equal(det.net_a.read("rpn0", 2), y0)
    # ... and 100x smaller ones keep their precision (a stale LARGE maximum would cost ~2^-14 of it)
    det.run(frames)
    ys = det.net_a.forward(x_small, out_name="p2")
    ref = fr_exact_p2(ctx, sd, x_small)
    assert np.abs(ys - ref).max() <= 2e-5 * np.abs(ref).max()
    # the detector itself is unaffected by the direct runs in between
    a = det.run(frames)
    b = fr.Detector(ctx, sd, 135, 240, max_frames=2, numerics="split").run(frames)
    assert all(np.array_equal(p, q) for p, q in zip(a, b))
    # net_b: RoI features 50x what RoIAlign last promised
    hb, wb, cb = det.prog_b.bufs[det.prog_b.named["roi_in"]]
    r = (rng.standard_normal((8, hb, wb, cb)) * 50.0).astype(np.float32)
    cls = det.net_b.forward(r, in_name="roi_in", out_name="cls")
    assert np.isfinite(cls).all()


def fr_exact_p2(ctx, sd, x):
    from posepipeline_amd.program import Net
    prog = fr.build_image_program(sd, x.shape[1], x.shape[2], "detector.")
    return Net(ctx, prog, max_batch=x.shape[0], numerics="exact").forward(x, out_name="p2")
