"""GPU parity of the ByteTrack configuration (YOLOX-X program + pre/post-processing + wrapper) against the CPU oracle.
Swish epilogues evaluate exp() in double precision with two different libms, so network outputs are compared with a
relative tolerance and the exact-match fraction is asserted separately; resize / pad, NMS picks and ids must be exact."""
import numpy as np
import pytest

from oracle import bytetrack as obt
from oracle import yolox as oyx
from posepipeline_amd.models import synth, yolox
from tests.test_gpu_detector import synth_frame

pytestmark = pytest.mark.gpu


def _sd(seed=6, obj_bias=-3.0):
    # keep the number of candidates (objectness * class score >= 0.01) small and non-zero
    return yolox.seed_synthetic_head(synth.synth_state_dict(yolox.yolox_param_shapes(), seed=seed), obj_bias)


def test_yolox_preprocess_network_and_detections(ctx):
    rng = np.random.default_rng(12)
    frames = np.stack([synth_frame(rng, 135, 240) for _ in range(2)])
    sd = _sd()
    det = yolox.YoloXDetector(ctx, sd, 135, 240, max_frames=2, scale=(96, 160))
    got = det.run(frames)
    model = oyx.YOLOXRef(sd)
    total = exact = 0
    for f in range(2):
        x, sf = oyx.preprocess(np.ascontiguousarray(frames[f][..., ::-1]), (96, 160))     # the wrapper hands mmtrack RGB
        assert x.shape[1:3] == (det.hp, det.wp) and np.array_equal(sf, det.scale_factor)
        # device input buffer: BGR order + zero 4th channel; the Focus weights undo the order
        din, _, _ = det.net.buffer("input")
        buf = np.empty((2, det.hp, det.wp, 4), np.float32)
        ctx.d2h(buf, int(din))
        assert np.array_equal(buf[f, :, :, 2::-1], x[0]) and not buf[f, :, :, 3].any()
        cls, reg, obj = model.forward(x)
        dev = {}
        for name, ref in (("cls", cls), ("reg", reg), ("obj", obj)):
            dev[name] = []
            for l in range(3):
                dptr, _, _ = det.net.buffer(f"{name}{l}")
                a = np.empty((2,) + ref[l].shape[1:], np.float32)
                ctx.d2h(a, int(dptr))
                d = a[f:f + 1]
                scale = max(1.0, float(np.abs(ref[l]).max()))
                assert np.allclose(d, ref[l], rtol=1e-4, atol=1e-5 * scale), (name, l)
                total += d.size
                exact += int((d == ref[l]).sum())
                dev[name].append(d)
        # decode + NMS from the DEVICE head outputs both ways: exact
        ref_dets = oyx.detections(dev["cls"], dev["reg"], dev["obj"], sf)
        assert ref_dets.shape == got[f].shape and np.array_equal(ref_dets, got[f])
        assert 0 < len(ref_dets) < 8192
    assert exact / total > 0.99, exact / total
    det.close()


def test_yolox_x_full_size_800x1440(ctx):
    """the program `bench.py --workload cascade0`-style ByteTrack runs time -- a 1080p frame at the config's test
    scale (800, 1440) (3rdparty/mmtracking/mot/bytetrack/bytetrack_yolox_x_crowdhuman_mot17-private-half.py:6,62-78) through
    YOLOX-X -- against the oracle: resize / pad `==`, network outputs rtol 1e-4 (Swish in double precision on two libms),
    decode + NMS from the device's head outputs `==`."""
    rng = np.random.default_rng(13)
    frames = np.stack([synth_frame(rng, 1080, 1920)])
    # objectness bias: with the small tests' -3.0 ALL 23 625 priors of an 800 x 1440 input pass score_thr 0.01 (seeded-random heads
    # barely vary) and the device's 8 192-candidate NMS capacity would cut them; -4.5 leaves ~1 100 (measured)
    sd = _sd(obj_bias=-4.5)
    det = yolox.YoloXDetector(ctx, sd, 1080, 1920, max_frames=1)               # default scale (800, 1440)
    got = det.run(frames)
    x, sf = oyx.preprocess(np.ascontiguousarray(frames[0][..., ::-1]))
    assert (det.hp, det.wp) == x.shape[1:3] == (800, 1440) or x.shape[1:3] == (det.hp, det.wp)
    assert np.array_equal(sf, det.scale_factor)
    din, _, _ = det.net.buffer("input")
    buf = np.empty((1, det.hp, det.wp, 4), np.float32)
    ctx.d2h(buf, int(din))
    assert np.array_equal(buf[0, :, :, 2::-1], x[0]) and not buf[0, :, :, 3].any()
    cls, reg, obj = oyx.YOLOXRef(sd).forward(x)
    total = exact = 0
    dev = {}
    for name, ref in (("cls", cls), ("reg", reg), ("obj", obj)):
        dev[name] = []
        for l in range(3):
            dptr, _, _ = det.net.buffer(f"{name}{l}")
            a = np.empty(ref[l].shape, np.float32)
            ctx.d2h(a, int(dptr))
            scale = max(1.0, float(np.abs(ref[l]).max()))
            assert np.allclose(a, ref[l], rtol=1e-4, atol=1e-5 * scale), (name, l)
            total += a.size
            exact += int((a == ref[l]).sum())
            dev[name].append(a)
    ref_dets = oyx.detections(dev["cls"], dev["reg"], dev["obj"], sf)
    assert ref_dets.shape == got[0].shape and np.array_equal(ref_dets, got[0])
    print(f"YOLOX-X {det.hp}x{det.wp}: {exact} of {total} head outputs bit-identical, {len(ref_dets)} detections")
    assert exact / total > 0.99 and 0 < len(ref_dets) < 8192
    det.close()


def test_mmtrack_bytetrack_wrapper(ctx, tmp_path, monkeypatch):
    monkeypatch.setenv("POSEPIPE_SYNTHETIC_WEIGHTS", "1")
    from posepipeline_amd import video
    from posepipeline_amd.wrappers import mmtrack as wmt
    rng = np.random.default_rng(13)
    frames = np.stack([synth_frame(rng, 96, 160) for _ in range(3)])
    path = str(tmp_path / "v.ppvid")
    video.write_ppvid(path, frames)
    tracks = wmt.mmtrack_bounding_boxes(path, "bytetrack")
    assert len(tracks) == 3
    sd = yolox.seed_synthetic_head(synth.synth_state_dict(yolox.yolox_param_shapes(), seed=6))
    model, trk = oyx.YOLOXRef(sd), obt.ByteTrackerRef()
    n_rows = 0
    for f in range(3):
        dets = oyx.detect(model, np.ascontiguousarray(frames[f][..., ::-1]))
        if len(dets) > 8192:
            pytest.skip("random head produced more candidates than pp_nms holds")
        rows = trk.step(dets)
        n_rows += len(dets)
        assert len(tracks[f]) == len(rows)
        for d, x in zip(tracks[f], rows):
            assert d["track_id"] == int(x[0]) and np.allclose(d["tlbr"], x[1:5], rtol=1e-4, atol=1e-2) and np.isclose(d["confidence"], x[5], rtol=1e-4)
    # With seeded weights the 200-layer network's activations collapse to near-constant maps, so this end-to-end run
    # usually yields no detections (n_rows == 0): it checks the plumbing (streaming, sizes, dtypes, empty-frame handling).
    # Non-trivial candidates are covered above (decode + NMS from device maps) and the association by
    # tests/test_bytetrack.py on seeded multi-person sequences.
    assert n_rows >= 0
    wmt._cache.clear()
