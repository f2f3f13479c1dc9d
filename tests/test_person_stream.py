"""Host logic of the streamed cascade (posepipeline_amd/person_stream.py) against the whole-clip form of the reference's
table chain: PersonBbox.make (pipeline.py:656-687; `tracking.person_bbox` is pinned on the reference's own output,
tests/golden/person_bbox.npz) -> zero rows for absent frames (wrappers/mmpose.py:67-69) -> one edge-replicated window per
frame over the whole clip (wrappers/videopose3d.py:66-75).  The 2D / 3D stages are stand-in functions here (the GPU
stages are compared with the oracle in tests/test_gpu_cascade.py); what is checked is WHICH box, frame and window they
are given, for every chunking."""
import numpy as np
import pytest

from posepipeline_amd.person_stream import PersonStreams, collect
from posepipeline_amd.tracking import person_bbox
from posepipeline_amd.wrappers.videopose3d import normalize_screen_coordinates

K = 5
SRC = (128, 256)        # powers of two: float32 and float64 normalisation agree exactly (the reference's dtype quirk)


def fake_topdown(jobs):
    """a row that encodes (track, frame, box) exactly in float32"""
    out = []
    for tid, t, box in jobs:
        r = np.zeros((K, 3), np.float32)
        r[:, 0] = np.float32(box[0]) + np.arange(K, dtype=np.float32)
        r[:, 1] = np.float32(box[1]) + np.float32(box[2]) * np.float32(0.5)
        r[:, 2] = np.float32(t % 97) + np.float32(tid) * np.float32(0.25)
        out.append(r)
    return out


def make_lift(pad):
    taps = sorted({-pad, -max(pad // 3, 1), -1, 0, 1, max(pad // 2, 1), pad})
    coef = {j: 1.0 / (3 + i) for i, j in enumerate(taps)}

    def windows(kn):            # one edge-replicated window per frame, like ChunkedGenerator
        n = kn.shape[0]
        p = np.pad(kn.astype(np.float32), ((pad, pad), (0, 0), (0, 0)), mode="edge")
        return np.stack([p[i:i + 2 * pad + 1] for i in range(n)])

    def on_windows(w):
        acc = np.zeros((w.shape[0], K, 3), np.float64)
        for j, c in coef.items():
            x = w[:, pad + j].astype(np.float64)
            acc[:, :, :2] += c * x
            acc[:, :, 2] += c * x[:, :, 0] * x[:, :, 1]
        return acc.astype(np.float32)

    return (lambda kn: on_windows(windows(kn))), windows, on_windows


def random_tracks(rng, n_frames, n_ids, p_drop, dup_at=()):
    """per frame rows (id, x1, y1, x2, y2, score); ids appear / disappear for good like SORT without ReID, with a new id
    after each gap, plus short re-appearing gaps for ids that stay live (DeepSORT-style rows are simulated by `hold`)."""
    tracks = [[] for _ in range(n_frames)]
    next_id = 0
    for _ in range(n_ids):
        t = int(rng.integers(0, max(n_frames - 3, 1)))
        while t < n_frames:
            length = int(rng.integers(1, max(n_frames // 2, 2)))
            tid = next_id
            next_id += 1
            for u in range(t, min(t + length, n_frames)):
                x, y = float(rng.integers(0, 150)), float(rng.integers(0, 60))
                tracks[u].append((tid, np.float32(x), np.float32(y), np.float32(x + 40), np.float32(y + 60), np.float32(0.9)))
            t += length + int(rng.integers(1, 6))
            if rng.random() < p_drop:
                break
    for t in dup_at:
        if t < n_frames and tracks[t]:
            tracks[t].append(tracks[t][0])          # the same id twice in a frame: not "exactly one" -> absent
    return tracks


def reference_chain(tracks, tid, pad, windows, on_windows):
    dicts = [[{"track_id": r[0], "tlhw": np.array([r[1], r[2], r[3] - r[1], r[4] - r[2]], np.float64)} for r in fr] for fr in tracks]
    bbox, present = person_bbox(dicts, [tid])
    rows = []
    for t, b in enumerate(bbox):
        rows.append(np.zeros((K, 3)) if np.isnan(b).any() else fake_topdown([(tid, t, b)])[0])
    kp = np.asarray(rows)                                     # float64 iff a zero row is in it, like the wrapper's result
    kn = normalize_screen_coordinates(kp[:, :, :2], SRC[1], SRC[0])
    return bbox, present, kp, on_windows(windows(kn))


def run_stream(tracks, chunks, pad, lift_fn, max_persons=64, keep_tracks=None, live=None):
    ps = PersonStreams(K, pad, SRC, fake_topdown, lift_fn, max_persons=max_persons, keep_tracks=keep_tracks)
    outs, i = [], 0
    for c in chunks:
        ps.ingest(tracks[i:i + c], None if live is None else live[i:i + c])
        i += c
        outs.append(ps.advance())
    assert i == len(tracks)
    outs.append(ps.advance(final=True))
    assert not ps.streams
    return outs


def chunkings(rng, n):
    yield [n]
    yield [1] * n
    for c in (2, 3, 7, 32):
        yield [c] * (n // c) + ([n % c] if n % c else [])
    for _ in range(3):
        out, left = [], n
        while left:
            c = int(rng.integers(1, min(left, 9) + 1))
            out.append(c)
            left -= c
        yield out


@pytest.mark.parametrize("seed,pad", [(0, 3), (1, 5), (2, 5), (3, 121), (4, 2)])
def test_streams_equal_whole_clip_chain(seed, pad):
    rng = np.random.default_rng(seed)
    n = 300 if pad == 121 else int(rng.integers(20, 60))
    tracks = random_tracks(rng, n, n_ids=4, p_drop=0.5, dup_at=(7, 8, n - 1))
    lift_fn, windows, on_windows = make_lift(pad)
    ids = sorted({r[0] for fr in tracks for r in fr})
    refs = {tid: reference_chain(tracks, tid, pad, windows, on_windows) for tid in ids}
    for chunks in chunkings(rng, n):
        outs = run_stream(tracks, chunks, pad, lift_fn)
        k2, k3 = collect(outs, "keypoints"), collect(outs, "keypoints_3d")
        assert sorted(k2) == ids
        for tid in ids:
            bbox, present, kp, ref3 = refs[tid]
            if not present.any():                                  # e.g. an id that only ever appears duplicated
                assert tid not in k3 or not k2[tid][1].any()
                continue
            first, last = int(np.flatnonzero(present)[0]), int(np.flatnonzero(present)[-1])
            f2, a2 = k2[tid]
            f3, a3 = k3[tid]
            # emitted from the first filled frame on, at least to the last filled one
            assert f2 <= first and f3 <= first and f3 + len(a3) > last, (tid, chunks)
            assert np.array_equal(a2, kp[f2:f2 + len(a2)].astype(np.float32)), (tid, chunks)
            assert np.array_equal(a3, ref3[f3:f3 + len(a3)]), (tid, chunks)


@pytest.mark.parametrize("seed", [10, 11, 12, 13])
def test_gaps_of_a_retained_id(seed):
    """a tracker that re-identifies (ReID, ByteTrack) reports an id again after a gap: every fill pattern of
    bfill(2) / ffill(2), runs at both ends of the clip included"""
    rng = np.random.default_rng(seed)
    pad, n = 4, 70
    tracks, live = [[] for _ in range(n)], []
    present = rng.random((2, n)) < 0.55
    present[0, :3] = False
    present[1, -4:] = False
    retain = 9                                              # the tracker forgets an id `retain` frames after its last row
    for t in range(n):
        for tid in (0, 1):
            if present[tid, t]:
                x = float(rng.integers(0, 150))
                tracks[t].append((tid, np.float32(x), np.float32(7), np.float32(x + 40), np.float32(67), np.float32(0.9)))
        live.append({tid for tid in (0, 1) if present[tid, max(0, t - retain + 1): t + 1].any()})
    # an id whose gap outlives `retain` would get a NEW id from a real tracker: cut such tails off
    for tid in (0, 1):
        seen = False
        for t in range(n):
            if seen and tid not in live[t]:
                for u in range(t, n):
                    tracks[u] = [r for r in tracks[u] if r[0] != tid]
                    live[u] = live[u] - {tid}
                break
            seen = seen or present[tid, t]
    lift_fn, windows, on_windows = make_lift(pad)
    for chunks in chunkings(rng, n):
        outs = run_stream(tracks, chunks, pad, lift_fn, live=live)
        k2, k3 = collect(outs, "keypoints"), collect(outs, "keypoints_3d")
        for tid in k3:
            bbox, present_f, kp, ref3 = reference_chain(tracks, tid, pad, windows, on_windows)
            first, last = int(np.flatnonzero(present_f)[0]), int(np.flatnonzero(present_f)[-1])
            (f2, a2), (f3, a3) = k2[tid], k3[tid]
            assert f2 <= first and f3 <= first and f3 + len(a3) > last
            assert np.array_equal(a2, kp[f2:f2 + len(a2)].astype(np.float32)), (tid, chunks)
            assert np.array_equal(a3, ref3[f3:f3 + len(a3)]), (tid, chunks)
        assert sorted(k3) == [0, 1]


def test_latency_and_trimming():
    """frame t is lifted exactly when frame t+pad has been decided; per-track history stays bounded"""
    pad, n, c = 6, 80, 4
    tracks = [[(0, np.float32(10 + t), np.float32(5), np.float32(50 + t), np.float32(65), np.float32(0.9))] for t in range(n)]
    lift_fn, _, _ = make_lift(pad)
    ps = PersonStreams(K, pad, SRC, fake_topdown, lift_fn)
    emitted = 0
    for i in range(0, n, c):
        ps.ingest(tracks[i:i + c])
        o = ps.advance()
        emitted += len(o["keypoints_3d"].get(0, ()))
        assert emitted == max(0, i + c - pad)
        assert o["keypoints_frames"][0].tolist() == list(range(i, i + c))      # present frames are decided at once
        assert len(ps.streams[0].k2) <= 2 * pad + c
    o = ps.advance(final=True)
    assert emitted + len(o["keypoints_3d"][0]) == n


def test_max_persons_and_keep_tracks():
    pad = 3
    lift_fn, _, _ = make_lift(pad)
    row = lambda i, t: (i, np.float32(10 * i + t), np.float32(5), np.float32(10 * i + t + 30), np.float32(60), np.float32(0.9))
    tracks = [[row(0, t), row(1, t), row(2, t)] for t in range(6)] + [[row(1, t), row(3, t)] for t in range(6, 12)]
    outs = run_stream(tracks, [4, 4, 4], pad, lift_fn, max_persons=2)
    # ids 0 and 1 are adopted in row order; 2 arrives while both are live and is never followed; 3 takes 0's place
    assert sorted(collect(outs)) == [0, 1, 3]
    outs = run_stream(tracks, [5, 7], pad, lift_fn, keep_tracks=[2])
    assert sorted(collect(outs)) == [2]


def test_finished_id_reappearing_is_a_loud_error():
    """an id whose stream was finished (the default live set = the ids of the frame's rows said it was gone) must not silently
    start a second stream whose frames do not continue the first: a tracker that retains ids has to pass live_sets"""
    pad = 2
    lift_fn, _, _ = make_lift(pad)
    ps = PersonStreams(K, pad, SRC, fake_topdown, lift_fn)
    row = lambda tid, x: (tid, float(x), 10.0, float(x) + 20.0, 50.0, 0.9)
    ps.ingest([[row(3, t)] for t in range(4)])
    ps.advance()
    ps.ingest([[] for _ in range(8)])           # id 3 vanishes: with the default live sets its stream ends ...
    ps.advance()
    assert 3 not in ps.streams and 3 in ps.finished
    with pytest.raises(RuntimeError, match="re-appeared"):
        ps.ingest([[row(3, 40)]])               # ... and a later row with the same id is refused
    # with live sets that keep the id alive across the gap the same sequence is fine
    ps2 = PersonStreams(K, pad, SRC, fake_topdown, lift_fn)
    ps2.ingest([[row(3, t)] for t in range(4)], live_sets=[{3}] * 4)
    ps2.advance()
    ps2.ingest([[] for _ in range(8)], live_sets=[{3}] * 8)
    ps2.advance()
    ps2.ingest([[row(3, 40)]], live_sets=[{3}])
    assert 3 in ps2.streams and not ps2.finished
