"""GPU parity of the DeepSortYOLOv4 path (tracking_method 0) against the CPU oracle (oracle/yolo.py, oracle/reid.py).

Integer stages (bicubic letterbox, patch resampling, NMS picks, int boxes, track ids) must be bit-exact.  The two
networks use Mish / ELU epilogues whose exp / log1p / tanh are evaluated in double precision by two different libms
(device ocml vs host glibc) and rounded to float32: a 1-ulp difference is possible with probability ~1e-9 per element,
so network outputs are compared with rtol 1e-5 and the exact-match fraction is asserted separately."""
import numpy as np
import pytest

from oracle import reid as oreid
from oracle import yolo as oyolo
from posepipeline_amd.models import mars, yolov4
from tests.test_gpu_detector import synth_frame

pytestmark = pytest.mark.gpu

NC = 3           # classes of the small test model (class 0 = person)
SIZE = 96        # network input of the small test model (grids 3, 6, 12)


def _frames(seed, n, h=135, w=240):
    rng = np.random.default_rng(seed)
    return np.stack([synth_frame(rng, h, w) for _ in range(n)])


@pytest.fixture(scope="module")
def yolo_sd():
    return yolov4.synth_params(yolov4.yolov4_param_shapes(NC), seed=4, head_bias=0.7)


def test_letterbox_bicubic_bit_exact(ctx, yolo_sd):
    frames = _frames(1, 2)
    det = yolov4.YoloV4Detector(ctx, yolo_sd, 135, 240, max_frames=2, size=SIZE, num_classes=NC)
    n = det.preprocess(frames)
    dptr, _, _ = det.net.buffer("input")
    got = np.empty((n, SIZE, SIZE, 4), np.float32)
    ctx.d2h(got, int(dptr))
    for f in range(n):
        ref = oyolo.network_input(np.ascontiguousarray(frames[f][..., ::-1]), (SIZE, SIZE))[0]
        assert np.array_equal(got[f, :, :, :3], ref)
        assert not got[f, :, :, 3].any()
    # frames already on the device + an upscaling geometry
    small = _frames(2, 1, 40, 30)
    det2 = yolov4.YoloV4Detector(ctx, yolo_sd, 40, 30, max_frames=1, size=SIZE, num_classes=NC)
    d = ctx.malloc(small.nbytes)
    ctx.h2d(d, small)
    det2.preprocess(None, frames_dev=(d, 1))
    dptr2, _, _ = det2.net.buffer("input")
    got2 = np.empty((1, SIZE, SIZE, 4), np.float32)
    ctx.d2h(got2, int(dptr2))
    assert np.array_equal(got2[0, :, :, :3], oyolo.network_input(np.ascontiguousarray(small[0][..., ::-1]), (SIZE, SIZE))[0])
    ctx.free(d)
    det.close()
    det2.close()


def test_yolov4_network_decode_and_boxes(ctx, yolo_sd):
    frames = _frames(3, 2)
    det = yolov4.YoloV4Detector(ctx, yolo_sd, 135, 240, max_frames=2, size=SIZE, num_classes=NC)
    got = det.run(frames)
    model = oyolo.YOLOv4Ref(yolo_sd, NC)
    total = exact = 0
    for f in range(2):
        x = oyolo.network_input(np.ascontiguousarray(frames[f][..., ::-1]), (SIZE, SIZE))
        ref_outs = model.forward(x)
        dev_outs = []
        for name, ref in zip(("y19", "y38", "y76"), ref_outs):
            dptr, _, _ = det.net.buffer(name)
            buf = np.empty((2,) + ref.shape[1:], np.float32)
            ctx.d2h(buf, int(dptr))
            dev = buf[f:f + 1]
            assert np.allclose(dev, ref, rtol=1e-5, atol=1e-6), name
            total += dev.size
            exact += int((dev == ref).sum())
            dev_outs.append(dev)
        # decode + NMS + int boxes: from the DEVICE head outputs both ways, so this part must be exact
        rb, rs = oyolo.person_detections(dev_outs, (135, 240), num_classes=NC)
        assert np.array_equal(got[f][0], rb) and np.array_equal(got[f][1], rs)
        assert len(rb) > 0                                      # the seeded head bias yields candidates
    assert exact / total > 0.9999, exact / total
    det.close()


def test_tf_nms_convention_matches_oracle(ctx):
    from posepipeline_amd import ops
    rng = np.random.default_rng(9)
    for n in (0, 1, 37, 300):
        c = rng.uniform(0, 200, (n, 2))
        wh = rng.uniform(1, 80, (n, 2))
        b = np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32)
        b[::7] = b[::7][:, [2, 3, 0, 1]]                        # flipped corners are legal for TensorFlow
        if n > 5:
            b[5, 2] = b[5, 0]                                   # zero-area box never suppresses / is never suppressed
        s = rng.uniform(0, 1, n).astype(np.float32)
        if n > 10:
            s[3] = s[9]                                         # tie -> lower index first
        for thr in (0.3, 0.5):
            got = ops.nms(ctx, b, s, thr, convention=2)
            assert np.array_equal(got, oyolo.tf_nms(b, s, 10 ** 6, thr)), (n, thr)


def test_reid_patches_and_encoder(ctx):
    frames = _frames(5, 2)
    sd = yolov4.synth_params(mars.mars_param_shapes(), seed=5)
    enc = mars.MarsEncoder(ctx, sd, 135, 240, max_patches=8)
    boxes = [np.array([[20, 10, 60, 110], [-4, 30, 50, 90], [200, 5, 60, 150]], np.int64),       # ints, like yolo.detect_image
             np.array([[100, 40, 30, 80], [0, 0, 240, 135], [239, 134, 5, 5]], np.int64)]      # last one: empty patch
    # patches: bit-exact against the restated cv2.resize
    rects = [(f,) + (mars.patch_rect(b, (135, 240)) or (0, 0, 0, 0)) for f in range(2) for b in boxes[f]]
    part = np.ascontiguousarray(np.array(rects, np.int32))
    din, _, _ = enc.net.buffer("input")
    from posepipeline_amd import _lib as L
    L.check(ctx.lib.pp_reid_patches(ctx.handle, L.ptr(np.ascontiguousarray(frames)), 2, 135, 240, L.PP_MEM_HOST, L.ptr(part),
                                    len(part), 128, 64, L.ptr(int(din))), "pp_reid_patches")
    got = np.empty((8, 128, 64, 4), np.float32)
    ctx.d2h(got, int(din))
    k = 0
    for f in range(2):
        for b in boxes[f]:
            p = oreid.extract_image_patch(frames[f], b)
            ref = np.zeros((128, 64, 3), np.float32) if p is None else p[..., ::-1].astype(np.float32)
            assert np.array_equal(got[k, :, :, :3], ref), (f, k)
            k += 1
    assert oreid.extract_image_patch(frames[1], boxes[1][2]) is None
    # features
    feats = enc.encode(frames, boxes)
    model = oreid.MarsSmall128Ref(sd)
    for f in range(2):
        ref = model.encode(frames[f], boxes[f])
        assert feats[f].dtype == np.float64 and feats[f].shape == ref.shape == (3, 128)
        assert np.allclose(feats[f], ref, rtol=1e-5, atol=1e-7)
        assert np.allclose(np.linalg.norm(feats[f], axis=1), 1.0, atol=1e-5)
        assert (feats[f] == ref).mean() > 0.99
    enc.close()


def test_tracking_bounding_boxes_wrapper(ctx, tmp_path, monkeypatch):
    """wrappers/deep_sort_yolov4/parser.py drop-in vs the oracle chain (YOLOv4 + encoder restatements feeding the
    fixture-pinned tracker): same tracks, ids and boxes."""
    monkeypatch.setenv("POSEPIPE_SYNTHETIC_WEIGHTS", "1")
    from posepipeline_amd import ops, video
    from posepipeline_amd.tracking import Tracker
    from posepipeline_amd.wrappers.deep_sort_yolov4 import parser
    frames = _frames(7, 3, 96, 128)
    path = str(tmp_path / "clip.ppvid")
    video.write_ppvid(path, frames)
    with pytest.raises(NotImplementedError):
        parser.tracking_bounding_boxes(path, outfile="x.avi")
    tracks = parser.tracking_bounding_boxes(path)
    assert len(tracks) == 3
    ysd = yolov4.seed_person_head(yolov4.synth_params(yolov4.yolov4_param_shapes(), seed=4))
    msd = yolov4.synth_params(mars.mars_param_shapes(), seed=5)
    ymodel, emodel = oyolo.YOLOv4Ref(ysd), oreid.MarsSmall128Ref(msd)
    trk = Tracker(mode=0, feat_dim=128, max_cosine_distance=0.3)
    n_tracks = 0
    for f in range(3):
        boxes, conf = oyolo.detect(ymodel, frames[f])
        feats = emodel.encode(frames[f], boxes)
        tlwh, sc = boxes.astype(np.float64), conf.astype(np.float64)
        keep = ops.nms(ctx, tlwh, sc, 1.0, convention=1) if len(tlwh) else np.zeros(0, np.int64)
        ids, t, info = trk.step(tlwh[keep], sc[keep], feats[keep])
        assert [d["track_id"] for d in tracks[f]] == [int(i) for i in ids]
        for d, b, s in zip(tracks[f], t, info):
            assert isinstance(d["track_id"], int) and set(d) == {"track_id", "tlhw", "tlbr", "time_since_update"}
            assert np.array_equal(d["tlhw"], b) and np.array_equal(d["tlbr"], np.concatenate([b[:2], b[:2] + b[2:]]))
            assert d["time_since_update"] == int(s[3])
        n_tracks += len(ids)
    assert n_tracks > 0
    parser._cache.clear()


def test_yolov4_full_size_416_80_classes(ctx):
    """the program `bench.py --workload track0 / cascade0` times -- 1080p frame, 416 x 416 letterbox, the 80-class
    COCO head (255 channels per scale) -- against the oracle: letterbox and the integer stages `==`, network outputs rtol 1e-5
    (Mish in double precision on two libms, see the module docstring).  wrappers/deep_sort_yolov4/yolo.py:18-129."""
    sd = yolov4.synth_params(yolov4.yolov4_param_shapes(80), seed=4, head_bias=0.7)
    frames = _frames(11, 1, 1080, 1920)
    det = yolov4.YoloV4Detector(ctx, sd, 1080, 1920, max_frames=1)            # defaults: size 416, 80 classes
    got = det.run(frames)
    x = oyolo.network_input(np.ascontiguousarray(frames[0][..., ::-1]), (416, 416))
    dptr, _, _ = det.net.buffer("input")
    din = np.empty((1, 416, 416, 4), np.float32)
    ctx.d2h(din, int(dptr))
    assert np.array_equal(din[0, :, :, :3], x[0]) and not din[0, :, :, 3].any()
    ref_outs = oyolo.YOLOv4Ref(sd, 80).forward(x)
    assert [r.shape for r in ref_outs] == [(1, 13, 13, 255), (1, 26, 26, 255), (1, 52, 52, 255)]
    total = exact = 0
    dev_outs = []
    for name, ref in zip(("y19", "y38", "y76"), ref_outs):
        dptr, _, _ = det.net.buffer(name)
        dev = np.empty(ref.shape, np.float32)
        ctx.d2h(dev, int(dptr))
        assert np.allclose(dev, ref, rtol=1e-5, atol=1e-6), name
        total += dev.size
        exact += int((dev == ref).sum())
        dev_outs.append(dev)
    rb, rs = oyolo.person_detections(dev_outs, (1080, 1920), num_classes=80)
    assert np.array_equal(got[0][0], rb) and np.array_equal(got[0][1], rs)
    print(f"YOLOv4 416 / 80 classes: {exact} of {total} head outputs bit-identical, {len(rb)} person boxes")
    assert exact / total > 0.9999 and len(rb) > 0
    det.close()
