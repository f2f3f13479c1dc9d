"""Video.get_robust_reader's validation + ffmpeg fallback (pose_pipeline/pipeline.py:47-87) on stand-ins for cv2 and ffmpeg
(neither is installed here): control flow, the exact transcode command, one validation per file, raw containers."""
import os
import sys
import types

import numpy as np
import pytest

from posepipeline_amd import video


class FakeCapture:
    """frames_ok[path] readable frames of `announced`"""
    announced = 12
    frames_ok: dict = {}
    opened: list = []

    def __init__(self, path):
        self.path, self.pos = path, 0
        FakeCapture.opened.append(path)

    def get(self, prop):
        return {7: self.announced, 5: 30.0, 3: 64, 4: 48}.get(prop, 0)

    def read(self):
        if self.pos >= self.frames_ok.get(self.path, self.announced):
            return False, None
        self.pos += 1
        return True, np.zeros((48, 64, 3), np.uint8)

    def release(self):
        pass


@pytest.fixture
def fake_cv2(monkeypatch):
    m = types.ModuleType("cv2")
    m.VideoCapture = FakeCapture
    m.CAP_PROP_FRAME_COUNT, m.CAP_PROP_FPS, m.CAP_PROP_FRAME_WIDTH, m.CAP_PROP_FRAME_HEIGHT = 7, 5, 3, 4
    monkeypatch.setitem(sys.modules, "cv2", m)
    FakeCapture.frames_ok, FakeCapture.opened = {}, []
    video._ROBUST.clear()
    return m


def test_good_file_is_validated_once(fake_cv2, tmp_path):
    p = tmp_path / "ok.mp4"
    p.write_bytes(b"x" * 100)
    calls = []
    assert video.robust_path(str(p), run=lambda cmd: calls.append(cmd)) == str(p)
    assert video.robust_path(str(p), run=lambda cmd: calls.append(cmd)) == str(p)
    assert calls == [] and FakeCapture.opened == [str(p)]                    # second call: cached verdict, no decode
    cap = video.open_video(str(p))
    assert cap.num_frames == 12 and cap.read_batch(5).shape == (5, 48, 64, 3)


def test_unreadable_frame_transcodes_whole_file(fake_cv2, tmp_path, capsys):
    p = tmp_path / "bad.mp4"
    p.write_bytes(b"x" * 100)
    FakeCapture.frames_ok[str(p)] = 7                                        # frame 7 of 12 does not decode
    calls = []

    def ffmpeg(cmd):                                                         # stand-in: writes the transcode, exit status 0
        calls.append(cmd)
        with open(cmd[-1], "wb") as f:
            f.write(b"transcoded")
        return types.SimpleNamespace(returncode=0)
    out = video.robust_path(str(p), run=ffmpeg)
    assert out != str(p) and out.endswith(".mp4")
    assert calls == [["ffmpeg", "-y", "-i", str(p), "-c:v", "libx264", "-b:v", "1M", out]]     # the reference's command
    assert "Transcoding" in capsys.readouterr().out
    assert video.robust_path(str(p), run=ffmpeg) == out and len(calls) == 1
    # through the table API: the path handed to the wrappers is the transcode
    import datetime
    from posepipeline_amd import djshim, pipeline as pl
    djshim.reset()
    key = {"video_project": "p", "filename": "bad"}
    pl.Video.insert1({**key, "video": str(p), "start_time": datetime.datetime(2024, 1, 1)})
    assert pl.Video.get_robust_reader(key, return_cap=False) == out
    djshim.reset()


def test_missing_ffmpeg_is_an_error_not_a_silent_short_clip(fake_cv2, tmp_path):
    p = tmp_path / "bad2.mp4"
    p.write_bytes(b"y" * 10)
    FakeCapture.frames_ok[str(p)] = 0

    def no_ffmpeg(cmd):
        raise FileNotFoundError("ffmpeg")
    with pytest.raises(RuntimeError, match="ffmpeg is not installed"):
        video.robust_path(str(p), run=no_ffmpeg)


def test_failed_transcode_is_an_error_and_is_not_cached(fake_cv2, tmp_path):
    """ffmpeg's exit status is checked: a failed (or empty) transcode raises and leaves no cached verdict behind, so the next
    call validates again instead of handing an empty file to every later stage"""
    import os
    p = tmp_path / "bad3.mp4"
    p.write_bytes(b"z" * 10)
    FakeCapture.frames_ok[str(p)] = 3
    outs = []

    def failing(cmd):
        outs.append(cmd[-1])
        return types.SimpleNamespace(returncode=1)
    with pytest.raises(RuntimeError, match="transcode failed"):
        video.robust_path(str(p), run=failing)
    assert not video._ROBUST and not os.path.exists(outs[0])
    with pytest.raises(RuntimeError, match="transcode failed"):               # empty output with status 0 is a failure, too
        video.robust_path(str(p), run=lambda cmd: types.SimpleNamespace(returncode=0))
    assert not video._ROBUST


def test_raw_containers_need_no_decoder(tmp_path):
    frames = np.arange(4 * 6 * 8 * 3, dtype=np.uint8).reshape(4, 6, 8, 3)
    path = str(tmp_path / "c.ppvid")
    video.write_ppvid(path, frames)
    assert video.robust_path(path) == path
    assert np.array_equal(video.open_video(path).read_batch(10), frames)
    with open(path, "r+b") as f:                                             # truncated: the stream ends at the last whole frame
        f.truncate(32 + 2 * 6 * 8 * 3 + 5)
    v = video.open_video(path)
    assert v.num_frames == 2 and np.array_equal(v.read_batch(10), frames[:2])


def test_y4m_container(tmp_path):
    """YUV4MPEG2 (8-bit 4:2:0): what `ffmpeg -i in.mp4 -pix_fmt yuv420p out.y4m` writes -- a standard container readable without any
    decoder.  Header parameters in any order, frame headers with and without parameters, a truncated last frame, colourspaces that
    are not 8-bit 4:2:0 refused; the planes come out as NV12 (U / V interleaved) for the device conversion path."""
    rng = np.random.default_rng(3)
    frames = rng.integers(0, 256, (5, 12, 16, 3)).astype(np.uint8)
    ref = video.bgr_to_nv12(frames)
    for params in (b"", b" Ixyz"):
        path = str(tmp_path / "c.y4m")
        video.write_y4m(path, frames, fps=25.0, frame_params=params)
        assert video.robust_path(path) == path
        v = video.open_video(path)
        assert (v.num_frames, v.height, v.width, v.fps, v.pixfmt) == (5, 12, 16, 25.0, "nv12")
        assert np.array_equal(np.concatenate([v.read_batch(2), v.read_batch(2), v.read_batch(9)]), ref)
        assert v.read_batch(3).shape == (0, 18, 16)
        with pytest.raises(RuntimeError, match="converted on the device"):
            v.read()
        v.release()
    with open(path, "r+b") as f:
        f.truncate(os.path.getsize(path) - 7)
    assert video.open_video(path).num_frames == 4
    # header variants: parameters in another order, no colourspace tag (= 420jpeg), extension tags
    body = open(path, "rb").read().split(b"\n", 1)[1]
    open(path, "wb").write(b"YUV4MPEG2 H12 W16 A1:1 Ip F30000:1001 XYSCSS=420JPEG\n" + body)
    v = video.open_video(path)
    assert (v.height, v.width, v.num_frames) == (12, 16, 4) and abs(v.fps - 29.97) < 0.01
    assert np.array_equal(v.read_batch(9), ref[:4])
    for cs in (b"C444", b"C422", b"C420p10", b"Cmono"):
        open(path, "wb").write(b"YUV4MPEG2 W16 H12 F25:1 " + cs + b"\n" + body)
        with pytest.raises(ValueError, match="not supported"):
            video.open_video(path)
    open(path, "wb").write(b"RIFF....AVI LIST")
    with pytest.raises(ValueError, match="not a YUV4MPEG2"):
        video.open_video(path)
    open(path, "wb").write(b"YUV4MPEG2 W15 H12 F25:1\n")
    with pytest.raises(ValueError, match="must be even"):
        video.open_video(path)
