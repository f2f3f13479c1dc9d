"""NV12 frame source: the oracle's conversion against the colour codes ITU-R BT.601 publishes, the container round trip, and the
host-side reader (no GPU)."""
import numpy as np
import pytest

from oracle import nv12 as onv
from posepipeline_amd import video


def _solid(y, u, v, h=2, w=4):
    a = np.empty((1, h * 3 // 2, w), np.uint8)
    a[:, :h] = y
    a[:, h:, 0::2] = u
    a[:, h:, 1::2] = v
    return a


@pytest.mark.parametrize("yuv,bgr", [
    ((16, 128, 128), (0, 0, 0)),            # black, white, and the 100 % primaries of BT.601 limited range
    ((235, 128, 128), (255, 255, 255)),
    ((81, 90, 240), (0, 0, 255)),
    ((145, 54, 34), (0, 255, 0)),
    ((41, 240, 110), (255, 0, 0)),
])
def test_oracle_reproduces_bt601_colour_codes(yuv, bgr):
    out = onv.nv12_to_bgr(_solid(*yuv), 2, 4)
    assert out.shape == (1, 2, 4, 3)
    assert np.abs(out.astype(int) - np.array(bgr)).max() <= 1      # the 8-bit codes are themselves rounded


def test_oracle_saturates_and_uses_one_chroma_pair_per_2x2_block():
    out = onv.nv12_to_bgr(_solid(255, 255, 255), 2, 4)               # out-of-range codes clamp to 255 / 0, no wrap-around
    assert out.max() == 255 and out.dtype == np.uint8
    assert onv.nv12_to_bgr(_solid(0, 0, 0), 2, 4)[0, 0, 0].tolist() == [0, 154, 0]   # y = 0: G = (2^19 + (852492 + 409993) * 128) >> 20
    a = _solid(128, 128, 128, 4, 8)
    a[0, 4, 0:2] = (240, 16)                                          # one chroma pair: pixels (0..1, 0..1) only
    out = onv.nv12_to_bgr(a, 4, 8)[0]
    assert len({tuple(p) for p in out[:2, :2].reshape(-1, 3)}) == 1
    assert not np.array_equal(out[0, 0], out[0, 2]) and np.array_equal(out[0, 2], out[3, 7])


def test_encoder_round_trip_is_close_on_smooth_frames():
    rng = np.random.default_rng(0)
    yy, xx = np.mgrid[0:64, 0:96]
    frames = np.stack([np.stack([128 + 100 * np.sin(xx / 29 + f), 128 + 90 * np.cos(yy / 23), 60 + xx + yy], -1)
                       for f in range(2)]).clip(0, 255).astype(np.uint8)
    back = onv.nv12_to_bgr(video.bgr_to_nv12(frames), 64, 96)
    err = np.abs(back.astype(int) - frames)
    assert err.mean() < 2.0 and err.max() <= 8, (err.mean(), err.max())      # 8-bit YUV quantisation + 2 x 2 chroma averaging
    noise = rng.integers(0, 256, (1, 8, 8, 3)).astype(np.uint8)              # and the encoder never leaves the legal range
    enc = video.bgr_to_nv12(noise)
    assert enc[:, :8].min() >= 16 and enc[:, :8].max() <= 235 and enc[:, 8:].min() >= 16 and enc[:, 8:].max() <= 240


def test_containers_open_as_nv12_sources(tmp_path):
    rng = np.random.default_rng(1)
    frames = rng.integers(0, 256, (5, 36, 40, 3)).astype(np.uint8)
    p = str(tmp_path / "clip.ppvid")
    video.write_ppvid(p, frames, fps=25.0, pixfmt="nv12")
    v = video.open_video(p)
    assert (v.pixfmt, v.num_frames, v.height, v.width, v.fps) == ("nv12", 5, 36, 40, 25.0)
    planes = video.bgr_to_nv12(frames)
    assert np.array_equal(v.read_batch(3), planes[:3]) and np.array_equal(v.read_batch(9), planes[3:]) and v.read_batch(1).shape[0] == 0
    with pytest.raises(RuntimeError, match="converted on the device"):
        v.read()
    assert video.robust_path(p) == p
    # header-less rawvideo file, size in the name; a truncated last frame is not a frame
    raw = str(tmp_path / "clip_40x36.nv12")
    with open(raw, "wb") as f:
        f.write(planes.tobytes()[:-7])
    v2 = video.open_video(raw)
    assert (v2.num_frames, v2.height, v2.width) == (4, 36, 40) and np.array_equal(v2.read_batch(4), planes[:4])
    with pytest.raises(ValueError, match="size in the name"):
        video.open_video(str(tmp_path / "clip.nv12"))
    # BGR containers are unchanged
    p1 = str(tmp_path / "bgr.ppvid")
    video.write_ppvid(p1, frames)
    assert getattr(video.open_video(p1), "pixfmt", "bgr24") == "bgr24"
    with pytest.raises(ValueError, match="must be even"):
        video.Nv12Video(np.zeros((1, 3, 5), np.uint8), 2, 5)
