"""DataJoint longblob codec (posepipeline_amd/blob.py).  DataJoint itself is not installed, so the byte
layouts below are this repo's reading of the published dj0 / mYm format (PARITY UNPINNED); what IS checked: the
hot-path table payloads survive a round trip value-for-value and dtype-for-dtype, the shim really stores bytes and
returns fresh copies, and malformed blobs are rejected."""
import datetime
import struct

import numpy as np
import pytest

from posepipeline_amd import blob, djshim, pipeline as pl


def _tracks(rng, n_frames=40):
    out = []
    for t in range(n_frames):
        fr = []
        for k in range(int(rng.integers(0, 4))):
            b = rng.uniform(0, 500, 4)
            fr.append({"track_id": int(k + 1), "tlbr": np.r_[b[:2], b[:2] + b[2:]], "tlhw": b.astype(np.float32),
                       "confidence": float(rng.uniform()), "time_since_update": 0})
        out.append(fr)
    return out


def _same(a, b):
    if isinstance(a, np.ndarray):
        return isinstance(b, np.ndarray) and a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b, equal_nan=a.dtype.kind == "f")
    if isinstance(a, dict):
        return isinstance(b, dict) and list(a) == list(b) and all(_same(a[k], b[k]) for k in a)
    if isinstance(a, (list, tuple)):
        return type(a) is type(b) and len(a) == len(b) and all(_same(x, y) for x, y in zip(a, b))
    return type(a) is type(b) and a == b


def test_hot_path_payloads_round_trip():
    rng = np.random.default_rng(0)
    bbox = rng.uniform(0, 1000, (300, 4))
    bbox[[5, 6, 200]] = np.nan                                      # PersonBbox.bbox: NaN rows = absent
    payloads = [
        _tracks(rng),                                               # TrackingBbox.tracks
        bbox, np.isfinite(bbox).all(1),                             # PersonBbox.bbox / present
        rng.standard_normal((300, 17, 3)).astype(np.float32),       # TopDownPerson.keypoints (float32 ...
        np.zeros((300, 136, 3)),                                    #  ... or float64 when a zero row was stacked in)
        rng.standard_normal((300, 17, 3)), [True] * 299 + [False],  # LiftingPerson.keypoints_3d / keypoints_valid
        [datetime.datetime(2024, 1, 1, 12, 30, 15, 250000) + datetime.timedelta(seconds=i / 30) for i in range(5)],  # VideoInfo.timestamps
        {"nested": (1, -2 ** 70, 2.5, None, "text", b"raw", 1 + 2j), "empty": [], "arr0d": np.float32(3.5)},
    ]
    for p in payloads:
        for compress in (True, False):
            b = blob.pack(p, compress=compress)
            assert isinstance(b, bytes)
            q = blob.unpack(b)
            if isinstance(p, dict) and "arr0d" in p:
                assert q["arr0d"].shape == () and q["arr0d"].dtype == np.float32 and q["nested"] == p["nested"] and q["empty"] == []
            else:
                assert _same(p, q)
    big = blob.pack(np.zeros((300, 17, 3)))
    assert big.startswith(b"ZL123\0") and len(big) < 1000            # compressible payloads are compressed ...
    assert blob.pack(rng.integers(0, 256, 4000, dtype=np.uint8))[:4] == b"mYm\0"   # ... incompressible ones are not; pure arrays keep mYm


def test_wire_layout_of_simple_values():
    assert blob.pack(5) == b"dj0\0" + b"\x0a" + struct.pack("<H", 1) + b"\x05"
    assert blob.pack(-1) == b"dj0\0\x0a\x01\x00\xff"
    assert blob.pack(True) == b"dj0\0\x0b\x01" and blob.pack(None) == b"dj0\0\xff"
    assert blob.pack(2.5) == b"dj0\0\x0d" + struct.pack("<d", 2.5)
    assert blob.pack("ab") == b"dj0\0\x05" + struct.pack("<Q", 2) + b"ab"
    a = np.array([[1, 2, 3], [4, 5, 6]], np.float32)
    expect = b"mYm\0A" + struct.pack("<QQQ", 2, 2, 3) + struct.pack("<II", 7, 0) + a.tobytes(order="F")
    assert blob.pack(a) == expect
    one = blob.pack([7])
    assert one == b"dj0\0\x02" + struct.pack("<Q", 1) + struct.pack("<Q", 4) + b"\x0a\x01\x00\x07"
    d = blob.pack({"k": 1.0})
    assert d == (b"dj0\0\x04" + struct.pack("<Q", 1) + struct.pack("<Q", 10) + b"\x05" + struct.pack("<Q", 1) + b"k" +
                 struct.pack("<Q", 9) + b"\x0d" + struct.pack("<d", 1.0))


def test_rejects_malformed_and_unserialisable():
    good = blob.pack([1, 2, 3], compress=False)
    for bad in (b"", b"xyz\0", good[:-1], good + b"\0", b"dj0\0\x99"):
        with pytest.raises(blob.BlobError):
            blob.unpack(bad)
    with pytest.raises(blob.BlobError):
        blob.pack(object())
    with pytest.raises(blob.BlobError):
        blob.pack(np.zeros(3, dtype=[("a", "f4")]))


def test_shim_stores_blobs_and_returns_copies(tmp_path):
    djshim.reset()
    key = {"video_project": "p", "filename": "f", "tracking_method": 5}
    pl.TrackingBboxMethod().insert1(key)
    tracks = _tracks(np.random.default_rng(1), 6)
    pl.TrackingBbox().insert1({**key, "tracks": tracks, "num_tracks": 3})
    stored = pl.TrackingBbox._store[0]["tracks"]
    assert isinstance(stored, bytes) and blob.unpack(stored) is not None and isinstance(pl.TrackingBbox._store[0]["num_tracks"], int)
    a = (pl.TrackingBbox & key).fetch1("tracks")
    b = (pl.TrackingBbox & key).fetch1("tracks")
    assert _same(a, tracks) and _same(b, tracks) and a is not b and a is not tracks
    if a[0]:
        a[0][0]["track_id"] = 999                                   # mutating a fetched value does not touch the table
        assert (pl.TrackingBbox & key).fetch1("tracks")[0][0]["track_id"] != 999
    with pytest.raises(blob.BlobError):                             # what MySQL could not hold is refused at insert time
        pl.TrackingBbox().insert1({**key, "tracking_method": 6, "tracks": [object()], "num_tracks": 0})
    djshim.reset()
