"""One long video across N ranks (one process per GPU, torch.distributed; NCCL = RCCL on MI355X, gloo on CPU).

The reference's only parallel mode is job-level: N OS processes take different *video keys* through
`populate(reserve_jobs=True)` (utils/standard_pipelines.py:40,46,100) -- pure replicas, which need nothing
from this module.  This module adds what the reference cannot do: sharding the FRAMES of one video
(SURVEY.md 8e).  Detection and the 2D stage are independent per frame; only association is sequential in
time, and it costs microseconds per frame on the host, so:

  1. rank r owns the contiguous frames [b_r, b_{r+1}), keeps them resident on its GPU (6.2 MB per 1080p frame of
     288 GB) and detects on them chunk by chunk;
  2. all ranks exchange fixed-size detection slabs (100 x 5 floats + a count per frame) with ONE
     all_gather -- ~2 KB per frame;
  3. every rank runs the identical sequential association over all frames (deterministic host code, so
     no broadcast of the result is needed) and the identical person-box decisions (person_stream.PersonStreams:
     PersonBbox's bfill / ffill for every followed track id);
  4. rank r runs the top-down 2D stage on the person-frames of ITS frames; a second all_gather (K x 3 floats per
     person-frame) gives every rank every 2D track;
  5. every rank lifts the tracks to 3D (0.03 GFLOP per frame: replicated instead of gathered).
No all-reduce anywhere.  Weights travel once per process start: `broadcast_blob_device` delivers rank 0's blobs as
device tensors that `pp_net_create_mem` consumes in place (no host round trip on the receivers).

The result equals the single-process streamed `Cascade` bit for bit, ids included: both are the same PersonStreams
decisions over the same tracker rows, and every compute stage is independent of batch composition.

The compute stages are callables so that the orchestration is testable on CPU with gloo and stub stages
(tests/test_distributed_gloo.py); `cascade_stages` binds the GPU stages of a Cascade (bench.py --mode shard,
tests/test_gpu_sharded.py).
"""
from __future__ import annotations

import numpy as np

from .person_stream import PersonStreams, collect

MAX_DET = 100


def shard_bounds(n: int, world: int):
    """contiguous, near-equal frame ranges: [b_0 = 0, ..., b_world = n]"""
    return [(n * r) // world for r in range(world + 1)]


def _to_tensor(a: np.ndarray, device):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def broadcast_blob(blob: np.ndarray, dist, device="cpu", src=0) -> np.ndarray:
    """weights: one broadcast from `src`; every rank returns the same float32 HOST array (CPU / gloo paths)"""
    import torch
    t = _to_tensor(blob.astype(np.float32), device) if dist.get_rank() == src else torch.empty(blob.size, dtype=torch.float32, device=device)
    dist.broadcast(t, src=src)
    return t.cpu().numpy()


def broadcast_blob_device(blob, n_floats: int, dist, device, src=0, backend="nccl"):
    """weights: one broadcast from `src`; returns a float32 tensor ON `device` holding rank `src`'s blob.  With RCCL
    ("nccl") the collective itself runs on device memory; in the gloo rehearsal (ranks sharing a GPU) the broadcast runs
    on host tensors and one upload follows.  `blob` is only read on rank `src` (the others may pass None): the receivers'
    programs are created from the tensor's data_ptr() (program.Net(blob_dev=...))."""
    import torch
    is_src = dist.get_rank() == src
    if backend == "nccl":
        t = torch.from_numpy(np.ascontiguousarray(blob, np.float32)).to(device) if is_src else \
            torch.empty(n_floats, dtype=torch.float32, device=device)
        dist.broadcast(t, src=src)
        torch.cuda.synchronize(device)          # the consumer reads the tensor on another stream (the library's own)
        return t
    h = torch.from_numpy(np.ascontiguousarray(blob, np.float32)) if is_src else torch.empty(n_floats, dtype=torch.float32)
    dist.broadcast(h, src=src)
    t = h.to(device)
    if str(device) != "cpu":
        torch.cuda.synchronize(device)
    return t


def broadcast_blob_fn(dist, device, backend="nccl", src=0, log=None):
    """blob_fn for cascade.Cascade / faster_rcnn.Detector: every program's blob comes from rank `src` and stays on the
    device (the receivers never touch their own prog.blob).  log: optional list that receives (name, n_floats, tensor).
    The returned pointer stays valid until the NEXT call of blob_fn (or until the closure is dropped): the closure itself holds
    the last tensor, so pp_net_create_ex's device-to-device copy (synchronous: it returns after the copy) never reads memory the
    caching allocator has taken back."""
    held = []

    def blob_fn(name, prog):
        n = int(prog.blob.size)
        t = broadcast_blob_device(prog.blob if dist.get_rank() == src else None, n, dist, device, src=src, backend=backend)
        held[:] = [t]                           # alive while the program that consumes it is being created
        if log is not None:
            log.append((name, n, t))
        return int(t.data_ptr()), n
    blob_fn.held = held
    return blob_fn


class _Gather:
    """One in-flight all_gather of a fixed-size float32 slab per rank: ONE collective on one flat tensor that lives on
    `device` (RCCL: device memory, no per-rank tensor list, one D2H copy when the result is read; gloo: host memory).
    start() returns immediately (async_op), result() waits and returns [world][rows][cols] as numpy."""

    # per-process totals since the last reset_stats(): collectives started, bytes every rank RECEIVES per collective summed
    # (world x slab), seconds result() blocked (the exposed part: the rest of a collective's life overlapped compute)
    stats = {"collectives": 0, "bytes": 0, "wait_s": 0.0}

    @classmethod
    def reset_stats(cls):
        cls.stats = {"collectives": 0, "bytes": 0, "wait_s": 0.0}

    def __init__(self, dist, device, local: np.ndarray):
        self.world = dist.get_world_size()
        self.shape = local.shape
        self.host = None
        _Gather.stats["collectives"] += 1
        _Gather.stats["bytes"] += int(self.world * local.size * (local.dtype.itemsize if local.dtype in (np.float32, np.float64, np.int64, np.int32) else 4))
        if self.world == 1 and getattr(dist, "numpy_only", False):      # a one-rank stand-in: nothing to exchange, no torch
            self.host = np.array(local, copy=True)[None]
            self.work = None
            return
        import torch
        # the slab keeps its dtype when torch can carry it (float32 / float64 / int64 / int32); anything else travels as float32
        if local.dtype not in (np.float32, np.float64, np.int64, np.int32):
            local = local.astype(np.float32)
        t = torch.from_numpy(np.ascontiguousarray(local)).to(device, non_blocking=True).reshape(-1)
        self.out = torch.empty(self.world * t.numel(), dtype=t.dtype, device=device)
        self.src = t                       # kept alive until the collective has finished
        fn = getattr(dist, "all_gather_into_tensor", None)
        if fn is not None:
            self.work = fn(self.out, t, async_op=True)
        else:                              # stand-in dist objects of the tests / world size 1
            outs = [self.out[r * t.numel():(r + 1) * t.numel()] for r in range(self.world)]
            self.work = dist.all_gather(outs, t)

    def result(self) -> np.ndarray:
        if self.host is not None:
            return self.host
        import time
        t0 = time.perf_counter()
        if self.work is not None and hasattr(self.work, "wait"):
            self.work.wait()
        got = self.out.cpu().numpy().reshape((self.world,) + tuple(self.shape))
        _Gather.stats["wait_s"] += time.perf_counter() - t0
        return got


def all_gather_ragged(local: np.ndarray, counts, dist, device="cpu") -> np.ndarray:
    """all_gather of per-rank arrays whose leading dim differs (counts[r] rows on rank r): pad to the max,
    one collective, trim.  Returns the concatenation in rank order, in the INPUT's dtype (float32 / float64 / int32 / int64
    travel as they are -- indices stay exact; other dtypes are gathered as float32)."""
    m = max(max(counts), 1)
    local = np.asarray(local)
    pad = np.zeros((m,) + local.shape[1:], dtype=local.dtype if local.dtype in (np.float32, np.float64, np.int64, np.int32) else np.float32)
    pad[: local.shape[0]] = local
    got = _Gather(dist, device, pad).result()
    return np.concatenate([got[r][: counts[r]] for r in range(len(counts))], axis=0)


def _takes_need(fn) -> bool:
    """does chunks_fn accept the keyword `need`?  A **kwargs catch-all only counts when the callable it forwards to (found through
    `__wrapped__` / functools.partial) names `need` itself: a plain forwarding wrapper around a source without the parameter must
    keep being called without it."""
    import functools
    import inspect
    seen = 0
    while fn is not None and seen < 8:
        seen += 1
        try:
            ps = inspect.signature(fn, follow_wrapped=False).parameters
        except (TypeError, ValueError):
            return False
        if "need" in ps:
            return True
        if not any(p.kind is inspect.Parameter.VAR_KEYWORD for p in ps.values()):
            return False
        fn = getattr(fn, "__wrapped__", None) or (fn.func if isinstance(fn, functools.partial) else None)
    return False


def process_video_sharded(dist, n_frames, chunks_fn, detect_fn, associate_fn, topdown_fn, lift_fn, src_hw, device="cpu",
                          num_joints=17, pad=121, max_persons=1, keep_tracks=None, timings=None):
    """Run the cascade on frames [0, n_frames) sharded over the ranks of `dist` (rank r owns the contiguous frames
    [b_r, b_{r+1})).

    chunks_fn(lo, hi[, need=])     -> this rank's frames as an ITERABLE of (first_frame, n, handle) chunks covering [lo, hi) in
                                      order.  It is called twice -- once for the detection pass, once for the 2D pass -- and
                                      may be lazy (a generator over a FrameStreamer): then only the chunk being processed
                                      (and the one being uploaded) is resident.  Resident shards return the same handles twice.
                                      If it takes a keyword `need`, the 2D pass passes a list of booleans, one per chunk of the
                                      first pass: chunk i holds a person-frame of this rank or not.  A lazy source then yields
                                      (first, n, None) for a chunk that is not needed WITHOUT reading or uploading it (without
                                      the keyword a lazy source re-reads and re-uploads such chunks just to have them skipped).
    detect_fn(handle, first, n)    -> list (per frame) of [m][5] float32 (x1, y1, x2, y2, score), m <= 100
    associate_fn(dets_all_frames)  -> per frame, tracker rows (track_id, x1, y1, x2, y2, score[, tlwh]); sequential
    topdown_fn(handle, n, idx, boxes) -> [len(idx)][K][3] key points of the person-frames (chunk-local frame idx, tlwh)
    lift_fn(kn)                    -> (m, J, 3) 3D joints of a normalised 2D context (m, K, 2)
    timings: optional dict that receives per-phase wall seconds of this rank.

    Both device passes advance in per-chunk ROUNDS: in round k every rank processes its k-th chunk and starts an asynchronous
    all_gather of that round's fixed-size result slab, which travels while chunk k + 1 computes; only the last round's
    collective is exposed.  Tracking is sequential in time over the WHOLE clip, so the association (host, microseconds per
    frame, identical on every rank) runs between the two passes.
    Returns on every rank: dict(tracks = per-frame rows, keypoints / keypoints_3d = {track_id: (first_frame, array)})."""
    import time
    _Gather.reset_stats()
    rank, world = dist.get_rank(), dist.get_world_size()
    # the round slabs carry frame indices and job indices next to float32 payload in ONE float32 tensor per collective: exact below 2^24
    assert n_frames < 2 ** 24, "frame indices travel in float32 slabs: shard clips of 16.7 M frames or more into several calls"
    b = shard_bounds(n_frames, world)
    lo, hi = b[rank], b[rank + 1]
    t0 = time.perf_counter()
    # ---- pass 1: detect chunk by chunk; each round's slab [chunk frames][100 x 5 + count + frame + more] is gathered asynchronously --
    width = MAX_DET * 5 + 3
    it = iter(chunks_fn(lo, hi))
    cur = next(it, None)
    assert cur is None or cur[0] == lo
    # slab height = the longest first chunk (later chunks are not longer); the only blocking collective of the pass
    rows = int(_Gather(dist, device, np.array([[cur[1] if cur else 0]], np.float32)).result().max())
    all_dets = [None] * n_frames
    round_of = np.full(n_frames, -1, np.int64)                 # which round brought frame f (every rank learns every rank's chunking)
    t_det = 0.0
    covered = lo

    def absorb(g, k):
        """slab of round k from every rank -> detections per frame; returns whether any rank has a further chunk"""
        more = False
        for r_slab in g.result():
            more |= bool(r_slab[0, -1] > 0)
            for row in r_slab:
                f = int(row[-2])
                if f >= 0:
                    all_dets[f] = row[: int(row[-3]) * 5].reshape(-1, 5).copy()
                    round_of[f] = k
        return more

    my_chunks = []                                             # (first, n) of this rank's chunks, in order (what `need` refers to)
    prev, k = None, 0
    while True:
        slab = np.full((max(rows, 1), width), -1.0, np.float32)
        slab[:, -1] = 0.0
        if cur is not None:
            first, n, handle = cur
            assert first == covered and n <= rows, (first, covered, n, rows)
            my_chunks.append((first, n))
            ta = time.perf_counter()
            dets = detect_fn(handle, first, n)
            t_det += time.perf_counter() - ta
            for i, d in enumerate(dets):
                d = np.asarray(d, np.float32).reshape(-1, 5)[:MAX_DET]
                slab[i, : len(d) * 5] = d.reshape(-1)
                slab[i, -3] = len(d)
                slab[i, -2] = first + i
            covered = first + n
            cur = next(it, None)
            slab[0, -1] = 1.0 if cur is not None else 0.0      # "this rank has a further chunk"
        g = _Gather(dist, device, slab)                        # round k travels while round k + 1 is detected
        if prev is not None:
            absorb(*prev)
        prev = (g, k)
        k += 1
        if cur is None:
            # nothing left locally: the round just issued tells whether another rank still has chunks (then this rank keeps
            # taking part in the collectives with empty slabs)
            if not absorb(*prev):
                prev = None
                break
            prev = None
    rounds = k
    assert covered == hi and all(d is not None for d in all_dets), (covered, hi)
    t1 = t0 + t_det
    t2 = time.perf_counter()
    # ---- identical sequential association + person-box decisions on every rank ----------------------------------------------------
    tracks = associate_fn(all_dets)
    t3 = time.perf_counter()
    t_2d = [0.0, 0.0]

    def sharded_topdown(jobs):
        # ---- pass 2: 2D on the own shard's person-frames, chunk by chunk; every round's rows are gathered asynchronously ---------
        ta = time.perf_counter()
        t_wait = 0.0
        assert len(jobs) < 2 ** 24, "job indices travel in float32 slabs"
        job_frames = np.array([j[1] for j in jobs], np.int64)
        owner = np.searchsorted(np.asarray(b[1:]), job_frames, side="right")
        # slab height: the most person-frames any (rank, round) computes -- known everywhere: every rank knows every job and,
        # from pass 1, which round brought which frame
        cap = 1
        for r in range(world):
            sel = owner == r
            if sel.any():
                cap = max(cap, int(np.bincount(round_of[job_frames[sel]], minlength=rounds).max()))
        out = [None] * len(jobs)

        def absorb2(g):
            for r_slab in g.result():
                for row in r_slab:
                    i = int(row[-1])
                    if i >= 0:
                        out[i] = row[:-1].reshape(num_joints, 3).copy()

        mine = np.sort(job_frames[owner == rank])
        need = [bool(np.searchsorted(mine, f0 + n0) > np.searchsorted(mine, f0)) for f0, n0 in my_chunks]
        it2 = iter(chunks_fn(lo, hi, need=need) if _takes_need(chunks_fn) else chunks_fn(lo, hi))
        prev2 = None
        for k in range(rounds):
            slab = np.full((cap, num_joints * 3 + 1), -1.0, np.float32)
            ks = np.flatnonzero((owner == rank) & (round_of[job_frames] == k))
            if len(ks):
                first, n, handle = next(it2)
                while first + n <= job_frames[ks].min():       # chunks without a person-frame are passed over (see `need`)
                    first, n, handle = next(it2)
                assert handle is not None, ("chunks_fn yielded a None handle for a chunk that holds person-frames of this rank: a source may "
                                            "return (first, n, None) ONLY for chunks whose `need` entry is False (see the docstring)")
                assert ((job_frames[ks] >= first) & (job_frames[ks] < first + n)).all()
                idx = (job_frames[ks] - first).astype(np.int32)
                boxes = np.array([jobs[i][2] for i in ks], np.float64)
                rows_k = np.asarray(topdown_fn(handle, n, idx, boxes), np.float32).reshape(len(ks), -1)
                slab[: len(ks), :-1] = rows_k
                slab[: len(ks), -1] = ks
            g = _Gather(dist, device, slab)                    # round k travels while round k + 1 computes
            if prev2 is not None:
                tw = time.perf_counter()
                absorb2(prev2)
                t_wait += time.perf_counter() - tw
            prev2 = g
        tw = time.perf_counter()
        if prev2 is not None:
            absorb2(prev2)
        t_wait += time.perf_counter() - tw
        assert all(o is not None for o in out)
        t_2d[0] += time.perf_counter() - ta - t_wait
        t_2d[1] += t_wait
        return out

    ps = PersonStreams(num_joints, pad, src_hw, sharded_topdown, lift_fn, max_persons=max_persons, keep_tracks=keep_tracks)
    ps.ingest(tracks)
    out = ps.advance(final=True)        # lifting of every track, replicated (0.03 GFLOP per frame)
    t4 = time.perf_counter()
    if timings is not None:
        timings.update(collectives=_Gather.stats["collectives"], collective_bytes=_Gather.stats["bytes"],
                       collective_wait=_Gather.stats["wait_s"])
        timings.update(detect=t1 - t0, gather_dets=t2 - t1, associate=t3 - t2, topdown=t_2d[0], gather_2d=t_2d[1],
                       lift=t4 - t3 - t_2d[0] - t_2d[1], total=t4 - t0, rounds=rounds)
    return dict(tracks=tracks, keypoints=collect([out], "keypoints"), keypoints_3d=collect([out], "keypoints_3d"))


def cascade_stages(cas, replay_fn=None):
    """The GPU stages of a `cascade.Cascade` as the callables process_video_sharded takes (chunks are device-resident:
    handle = device pointer).  replay_fn(first, n): per-frame boxes that stand in for the detector's output downstream
    (random-weight detector) -- the detector still runs on every frame."""
    from .tracking import Tracker
    from .wrappers.videopose3d import lift
    assert cas.tracking == "MMTrack_deepsort", "the sharded path is built for the mmtrack configuration"
    h, w = cas.src
    cap = max(1, cas.pose_net.max_batch // 2)

    def detect_fn(handle, first, n):
        dets = cas.detector.run(None, frames_dev=(handle, n))
        return dets if replay_fn is None else replay_fn(first, n)

    def associate_fn(all_dets):
        trk = Tracker(mode=1, match_iou_thr=0.5, obj_score_thr=0.5)
        tracks = []
        for rows in all_dets:
            rows = np.asarray(rows, np.float32).reshape(-1, 5)
            ids, _, info = trk.step(rows[:, :4].astype(np.float64), rows[:, 4].astype(np.float64))
            tracks.append([(int(i), *rows[j]) for i, j in zip(ids, info[:, 1])])
        return tracks

    def topdown_fn(handle, n, idx, boxes):
        out = []
        for i0 in range(0, len(idx), cap):
            k2, _ = cas.topdown.run(handle, idx[i0:i0 + cap], boxes[i0:i0 + cap], frames_dev_shape=(n, h, w))
            out.append(k2)
        return np.concatenate(out)

    return detect_fn, associate_fn, topdown_fn, lambda kn: lift(cas.lift_net, cas.lift_spec, kn[:, :17])
