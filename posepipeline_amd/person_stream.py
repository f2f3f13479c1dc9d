"""Streaming form of PersonBbox -> TopDownPerson -> LiftingPerson for every followed track id (host logic only).

The reference runs the three tables one after the other over the whole clip for ONE annotated subject
(`PersonBboxValid.keep_tracks`, pose_pipeline/pipeline.py:637-687, 1017-1095, 1259-1416).  `PersonStreams` produces the
same values for `keep_tracks = [tid]`, for every followed `tid`, while frames arrive in chunks:

  * the person box of frame t is the track's box iff exactly one row of the frame carries `tid`
    (pipeline.py:662-667), missing frames are filled by `bfill(limit=2)` then `ffill(limit=2)` (:678-681);
  * 2D key points of frame t: the top-down stage on that box, `zeros((K, 3))` where the box is still NaN
    (wrappers/mmpose.py:67-69);
  * 3D joints of frame t: VideoPose3D on the window [t-121, t+121] of that 2D track, edge-replicated at the two
    ends of the CLIP only (wrappers/videopose3d.py:66-75), zero rows included.

Both fills and the lifting look into the future, so the streaming form has latency: the box of an absent frame is
decided once the next two frames are known, and frame t is lifted once frame t+121 has its key points; `advance(final=
True)` (end of clip) emits the rest with the reference's edge replication.  Same FLOPs as the eager form.
The reference stores rows for ALL frames of the clip per subject; here (tid, t) pairs are emitted from the track's first
filled frame to its last one (a track the tracker has dropped is finalised at once: its future is known to be absent),
each equal to the reference's value for that frame.  One reference quirk is followed on a best-effort basis:
`process_videopose3d` normalises a float32 key-point array in float32 and a float64 one (any zero row in the clip makes
the stacked array float64) in float64; a stream uses float32 arithmetic while every frame so far had a box and float64
afterwards, which is what the reference does unless the first absent frame comes later in the clip (then earlier frames
differ by <= 1 float32 ulp of the normalised input).

The compute stages are callables, so this module needs no GPU (tests/test_person_stream.py):
    topdown_fn(jobs)  jobs = [(track_id, frame, tlwh float64[4])] -> [len(jobs)][K][3] float32
    lift_fn(kn)       kn = (n, K, 2) normalised key points -> (n, J, 3)
"""
from __future__ import annotations

import numpy as np

from .wrappers.videopose3d import normalize_screen_coordinates

FILL_LIMIT = 2          # PersonBbox.make: fillna(method="bfill"/"ffill", limit=2), pipeline.py:680-681


class _PersonStream:
    """Streaming PersonBbox(keep_tracks=[tid]) -> TopDownPerson -> LiftingPerson state of one track id."""

    def __init__(self, tid: int, born: int, num_joints: int):
        self.tid = tid
        self.first = max(0, born - FILL_LIMIT)   # first frame with a (back-filled) box
        self.raw: dict = {}                      # frame -> tlwh of frames whose neighbours' fills are still open
        self.last_raw = born                     # last frame with a raw (un-filled) box
        self.next_dec = self.first               # next frame whose box has not been decided
        self.k2_base = self.first                # frame index of k2[0]
        self.k2: list = []                       # decided 2D rows (K,3) float32, zeros where the box stayed NaN
        self.next3d = self.first                 # next frame whose 3D has not been emitted
        self.any_absent = self.first > 0         # reference: a zero row makes the clip's key-point array float64
        self.live = True
        self.k = num_joints

    def decide(self, n_frames: int, final: bool):
        """Box decisions for frames [next_dec, n_frames) in order, as far as they can be made.  final: nothing will
        ever be added to `raw` (end of clip, or the tracker dropped the id).  Yields (frame, tlwh or None)."""
        raw = self.raw
        while self.next_dec < n_frames:
            t = self.next_dec
            box = raw.get(t)
            if box is None:
                # bfill(limit=2): the next valid row, if it is at most 2 frames ahead
                for d in range(1, FILL_LIMIT + 1):
                    if t + d in raw:
                        box = raw[t + d]
                        break
                    if t + d >= n_frames and not final:
                        return                       # frame t+d not tracked yet: undecidable for now
                if box is None:
                    # ffill(limit=2) over what is still missing: the previous valid row, at most 2 frames back
                    for d in range(1, FILL_LIMIT + 1):
                        if t - d in raw:
                            box = raw[t - d]
                            break
            self.next_dec = t + 1
            raw.pop(t - FILL_LIMIT, None)            # nothing looks further back than 2 frames
            yield t, box

    def row(self, t: int, n_total):
        """2D row of frame t for the lifting context (n_total: clip length when known, else None)."""
        if n_total is not None and t >= n_total:
            t = n_total - 1                          # np.pad(..., 'edge') at the end of the clip
        if t < 0:
            t = 0                                    # ... and at its start
        i = t - self.k2_base
        if i < 0 or i >= len(self.k2):
            return None                              # before the first box / after the track ended: a zero row
        return self.k2[i]


class PersonStreams:
    """max_persons: tracks followed at the same time (new ids are adopted, in row order, while fewer are live; an id that
    appears while max_persons others are live is never followed); keep_tracks: follow exactly these ids instead."""

    def __init__(self, num_joints: int, pad: int, src_hw, topdown_fn, lift_fn, max_persons: int = 1, keep_tracks=None):
        self.k, self.pad, self.src = int(num_joints), int(pad), tuple(src_hw)
        self.topdown_fn, self.lift_fn = topdown_fn, lift_fn
        self.max_persons = max_persons
        self.keep_tracks = None if keep_tracks is None else set(int(k) for k in keep_tracks)
        self.n_frames = 0             # frames tracked so far
        self.streams: dict = {}       # track_id -> _PersonStream (followed ids that still owe output)
        self.ignored: set = set()
        self.finished: set = set()    # followed ids whose stream was completed and emitted (their tracker said they were gone)

    def ingest(self, chunk_tracks, live_sets=None):
        """chunk_tracks: per frame, rows (track_id, x1, y1, x2, y2, score[, tlwh]) as the tracking stage reports them.  Records the
        raw presence of the followed ids (exactly one row of the frame carries the id, pipeline.py:662-667).
        live_sets: per frame, the ids the tracker can still report later (Tracker.live_ids after that frame); default: the
        ids of the frame's rows, which is the live set of both built trackers (they report every track they keep)."""
        for i, rows in enumerate(chunk_tracks):
            t = self.n_frames
            count: dict = {}
            for r in rows:
                count[r[0]] = count.get(r[0], 0) + 1
            live = count if live_sets is None else live_sets[i]
            for st in self.streams.values():
                if st.live and st.tid not in live:
                    st.live = False
            for r in rows:
                tid, x1, y1, x2, y2 = r[:5]
                if tid in self.ignored:
                    continue
                if tid in self.finished:
                    # the stream of this id was closed because the tracker's live set no longer held it; a second stream would
                    # emit frames that do not continue the first one.  Trackers that retain ids across misses must pass
                    # live_sets (SortReidTracker.live_ids / ByteTracker.live_ids do).
                    raise RuntimeError(f"track id {tid} re-appeared in frame {t} after its stream was finished: pass live_sets= "
                                       "for a tracker that keeps ids alive across missed frames")
                st = self.streams.get(tid)
                if st is None:
                    if self.keep_tracks is not None:
                        follow = tid in self.keep_tracks
                    else:
                        follow = sum(1 for s in self.streams.values() if s.live) < self.max_persons
                    if not follow:
                        self.ignored.add(tid)
                        continue
                    st = self.streams[tid] = _PersonStream(tid, t, self.k)
                if count[tid] == 1:
                    # tlhw as the wrappers store it: [x1, y1, x2 - x1, y2 - y1] on mmtrack's float32 row (wrappers/mmtrack.py:55),
                    # or the tracker's own to_tlwh() when the row carries it as a 7th element (deep_sort_yolov4/parser.py:80)
                    st.raw[t] = (np.array(r[6], np.float64) if len(r) > 6 else np.array([x1, y1, x2 - x1, y2 - y1], np.float64))
                    st.last_raw = t
            self.n_frames += 1

    def _lift_range(self, st: _PersonStream, lo: int, hi: int, n_total):
        """3D of frames [lo, hi) of one stream from the context [lo - pad, hi + pad)"""
        pad = self.pad
        zero = np.zeros((self.k, 3), np.float32)
        rows = [st.row(t, n_total) for t in range(lo - pad, hi + pad)]
        arr = np.stack([zero if r is None else r for r in rows])[:, :, :2]
        arr = arr.astype(np.float64) if st.any_absent else arr          # the reference's dtype-dependent normalisation
        kn = normalize_screen_coordinates(arr, self.src[1], self.src[0])
        out = self.lift_fn(kn)
        return out[pad:pad + hi - lo]

    def advance(self, final: bool = False):
        """Decide boxes, run 2D, emit 3D for everything that has become computable.  final: end of clip.
        Returns dict(keypoints / keypoints_frames = {track_id: (n,K,3) / (n,) frame numbers} decided in this call,
                     keypoints_3d / keypoints_3d_frames = {track_id: (m,J,3) / (m,)} emitted in this call)."""
        n1 = self.n_frames
        jobs, decided = [], {}
        for tid in sorted(self.streams):
            st = self.streams[tid]
            for t, box in st.decide(n1, final or not st.live):
                decided.setdefault(tid, []).append(t)
                if box is not None:
                    jobs.append((tid, t, box))
        rows2d = self.topdown_fn(jobs) if jobs else []
        k2 = {(j[0], j[1]): r for j, r in zip(jobs, rows2d)}
        kp, kp_frames, kp3, kp3_frames = {}, {}, {}, {}
        pad = self.pad
        zero = np.zeros((self.k, 3), np.float32)
        for tid in sorted(self.streams):
            st = self.streams[tid]
            ts = decided.get(tid, [])
            for t in ts:
                row = k2.get((tid, t))
                if row is None:
                    st.any_absent = True
                    row = zero
                st.k2.append(row)
            # a dropped track owes nothing past its last (forward-filled) box: the rows after it are zero rows for good
            cap = None if st.live else st.last_raw + FILL_LIMIT + 1
            te = [t for t in ts if cap is None or t < cap]
            if te:
                kp[tid] = np.stack(st.k2[te[0] - st.k2_base: te[-1] + 1 - st.k2_base])
                kp_frames[tid] = np.array(te, np.int64)
            # ... and is finalised once its last decided row is one of those zero rows
            ended = (not st.live) and st.next_dec > cap
            if final:
                hi, n_total = st.next_dec if cap is None else min(st.next_dec, cap), n1
            elif ended:
                hi, n_total = cap, None
            else:
                hi, n_total = st.next_dec - pad, None
            if hi > st.next3d:
                kp3[tid] = self._lift_range(st, st.next3d, hi, n_total)
                kp3_frames[tid] = np.arange(st.next3d, hi, dtype=np.int64)
                st.next3d = hi
            if final or ended:
                del self.streams[tid]
                self.finished.add(tid)
            else:
                drop = st.next3d - pad - st.k2_base       # rows no window will read again
                if drop > 0:
                    del st.k2[:drop]
                    st.k2_base += drop
        return dict(keypoints=kp, keypoints_frames=kp_frames, keypoints_3d=kp3, keypoints_3d_frames=kp3_frames)


def collect(outs, what="keypoints_3d"):
    """Stitch advance() / Cascade.step() results: {track_id: (first_frame, array over consecutive frames)}."""
    acc: dict = {}
    for o in outs:
        for tid, arr in o[what].items():
            fr_ = o[what + "_frames"][tid]
            a = acc.setdefault(tid, ([], []))
            a[0].append(np.asarray(fr_))
            a[1].append(np.asarray(arr))
    res = {}
    for tid, (f, a) in acc.items():
        f, a = np.concatenate(f), np.concatenate(a)
        assert np.array_equal(f, np.arange(f[0], f[0] + len(f))), (tid, f)       # consecutive, in order, once each
        res[tid] = (int(f[0]), a)
    return res
