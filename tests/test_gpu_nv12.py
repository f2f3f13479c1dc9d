"""NV12 frame source on the device: conversion bit for bit against oracle/nv12.py, the streamer's NV12 path, and the cascade on an
NV12 clip against the same cascade on the frames the oracle converts."""
import ctypes as C

import numpy as np
import pytest

from oracle import nv12 as onv
from posepipeline_amd import _lib as L
from posepipeline_amd import video
from posepipeline_amd.streaming import FrameStreamer

pytestmark = pytest.mark.gpu


def _convert(ctx, planes, h, w):
    n = planes.shape[0]
    src = ctx.malloc(max(planes.nbytes, 16))
    dst = ctx.malloc(max(n * h * w * 3, 16))
    ctx.h2d(src, planes)
    L.check(ctx.lib.pp_nv12_to_bgr(ctx.handle, C.c_void_p(src), n, h, w, C.c_void_p(dst)), "pp_nv12_to_bgr")
    out = np.empty((n, h, w, 3), np.uint8)
    ctx.d2h(out, dst)
    ctx.free(src)
    ctx.free(dst)
    return out


@pytest.mark.parametrize("n,h,w", [(1, 2, 4), (3, 34, 20), (2, 270, 484), (1, 1080, 1920), (5, 16, 4096), (2, 480, 854), (3, 6, 2), (1, 4, 6)])
def test_nv12_to_bgr_equals_the_oracle_on_every_code(ctx, n, h, w):
    rng = np.random.default_rng(h + w)
    planes = rng.integers(0, 256, (n, h * 3 // 2, w)).astype(np.uint8)      # every byte value, legal range or not
    planes[0, :2, :2] = [[0, 255], [16, 235]]
    assert np.array_equal(_convert(ctx, planes, h, w), onv.nv12_to_bgr(planes, h, w))


def test_nv12_rejects_odd_sizes(ctx):
    d = ctx.malloc(4096)
    assert ctx.lib.pp_nv12_to_bgr(ctx.handle, C.c_void_p(d), 1, 3, 8, C.c_void_p(d)) == -1
    assert ctx.lib.pp_nv12_to_bgr(ctx.handle, C.c_void_p(d), 1, 4, 7, C.c_void_p(d)) == -1
    assert ctx.lib.pp_nv12_to_bgr(ctx.handle, C.c_void_p(d), 0, 4, 8, C.c_void_p(d)) == 0       # empty clip
    ctx.free(d)


def test_streamer_uploads_nv12_and_yields_bgr(ctx, tmp_path):
    """7 frames in chunks of 3 (ragged last chunk) from a PPVID002 file: every chunk on the device equals the oracle's conversion
    of the same planes; `ahead` is resident (converted) before the chunk in front of it is released"""
    rng = np.random.default_rng(7)
    h, w = 36, 40
    frames = rng.integers(0, 256, (7, h, w, 3)).astype(np.uint8)
    path = str(tmp_path / "clip.ppvid")
    video.write_ppvid(path, frames, pixfmt="nv12")
    ref = onv.nv12_to_bgr(video.bgr_to_nv12(frames), h, w)
    st = FrameStreamer(ctx, video.open_video(path), 3)
    seen = []
    for dev, n, first in st:
        got = np.empty((n, h, w, 3), np.uint8)
        ctx.d2h(got, dev)
        assert np.array_equal(got, ref[first:first + n])
        if st.ahead is not None:
            adev, an, afirst = st.ahead
            nxt = np.empty((an, h, w, 3), np.uint8)
            ctx.d2h(nxt, adev)
            assert np.array_equal(nxt, ref[afirst:afirst + an])
        seen.append((first, n))
        st.release()
    st.close()
    assert seen == [(0, 3), (3, 3), (6, 1)]


def test_cascade_on_an_nv12_clip_equals_the_cascade_on_the_converted_frames(ctx, tmp_path):
    from posepipeline_amd.cascade import Cascade
    from tests.test_gpu_cascade import _setup, synth_frame
    rng = np.random.default_rng(5)
    h, w = 136, 240
    frames = np.stack([synth_frame(rng, h, w) for _ in range(5)])
    path = str(tmp_path / "clip.ppvid")
    video.write_ppvid(path, frames, pixfmt="nv12")
    bgr = onv.nv12_to_bgr(video.bgr_to_nv12(frames), h, w)
    det_sd, pose_spec, pose_sd, lift_sd = _setup(h, w)
    cas = Cascade(ctx, det_sd, pose_sd, lift_sd, h, w, chunk=2, max_persons=1, pose_spec=pose_spec)
    gt = [np.array([[60 + 4 * t, 20, 130 + 4 * t, 120, 0.9]], np.float32) for t in range(5)]
    ref = list(cas.run_video(video.open_video(bgr), replay_fn=lambda first, n: gt[first:first + n]))
    cas.reset()
    got = list(cas.run_video(video.open_video(path), replay_fn=lambda first, n: gt[first:first + n]))
    assert [o["first_frame"] for o in got] == [o["first_frame"] for o in ref]
    for a, b in zip(ref, got):
        assert a["tracks"] == b["tracks"]
        for what in ("keypoints", "keypoints_3d"):
            assert a[what].keys() == b[what].keys()
            for tid in a[what]:
                assert np.array_equal(a[what][tid], b[what][tid])
    # and with the detector reading the converted frames itself (no replay)
    cas.reset()
    r1 = next(iter(cas.run_video(video.open_video(bgr))))
    cas.reset()
    g1 = next(iter(cas.run_video(video.open_video(path))))
    assert r1["tracks"] == g1["tracks"] and len(r1["tracks"]) > 0


def test_y4m_clip_takes_the_nv12_path(ctx, tmp_path):
    """a YUV4MPEG2 file (8-bit 4:2:0, what ffmpeg writes with -pix_fmt yuv420p) streams like an NV12 source: every chunk on the device
    equals the oracle's conversion of the same planes, and the cascade on the .y4m clip == the cascade on the converted frames"""
    from posepipeline_amd.cascade import Cascade
    from tests.test_gpu_cascade import _setup, synth_frame
    rng = np.random.default_rng(6)
    h, w = 136, 240
    frames = np.stack([synth_frame(rng, h, w) for _ in range(5)])
    path = str(tmp_path / "clip.y4m")
    video.write_y4m(path, frames, fps=30.0)
    bgr = onv.nv12_to_bgr(video.bgr_to_nv12(frames), h, w)
    st = FrameStreamer(ctx, video.open_video(path), 2)
    for dev, n, first in st:
        got = np.empty((n, h, w, 3), np.uint8)
        ctx.d2h(got, dev)
        assert np.array_equal(got, bgr[first:first + n])
        st.release()
    st.close()
    det_sd, pose_spec, pose_sd, lift_sd = _setup(h, w)
    cas = Cascade(ctx, det_sd, pose_sd, lift_sd, h, w, chunk=2, max_persons=1, pose_spec=pose_spec)
    ref = list(cas.run_video(video.open_video(bgr)))
    cas.reset()
    got = list(cas.run_video(video.open_video(path)))
    assert len(ref[0]["tracks"]) > 0
    for a, b in zip(ref, got):
        assert a["tracks"] == b["tracks"]
        for what in ("keypoints", "keypoints_3d"):
            assert a[what].keys() == b[what].keys()
            for tid in a[what]:
                assert np.array_equal(a[what][tid], b[what][tid])
