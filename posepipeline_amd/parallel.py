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


def all_gather_ragged(local: np.ndarray, counts, dist, device="cpu") -> np.ndarray:
    """all_gather of per-rank arrays whose leading dim differs (counts[r] rows on rank r): pad to the max,
    one collective, trim.  Returns the concatenation in rank order."""
    import torch
    world = dist.get_world_size()
    m = max(max(counts), 1)
    pad = np.zeros((m,) + local.shape[1:], dtype=local.dtype)
    pad[: local.shape[0]] = local
    t = _to_tensor(pad, device)
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t)
    return np.concatenate([o.cpu().numpy()[: counts[r]] for r, o in enumerate(outs)], axis=0)


def process_video_sharded(dist, n_frames, chunks_fn, detect_fn, associate_fn, topdown_fn, lift_fn, src_hw, device="cpu",
                          num_joints=17, pad=121, max_persons=1, keep_tracks=None, timings=None):
    """Run the cascade on frames [0, n_frames) sharded over the ranks of `dist`.

    chunks_fn(lo, hi)              -> this rank's frames as a list of (first_frame, n, handle) chunks covering [lo, hi)
    detect_fn(handle, first, n)    -> list (per frame) of [m][5] float32 (x1, y1, x2, y2, score), m <= 100
    associate_fn(dets_all_frames)  -> per frame, tracker rows (track_id, x1, y1, x2, y2, score[, tlwh]); sequential
    topdown_fn(handle, n, idx, boxes) -> [len(idx)][K][3] key points of the person-frames (chunk-local frame idx, tlwh)
    lift_fn(kn)                    -> (m, J, 3) 3D joints of a normalised 2D context (m, K, 2)
    timings: optional dict that receives per-phase wall seconds of this rank.
    Returns on every rank: dict(tracks = per-frame rows, keypoints / keypoints_3d = {track_id: (first_frame, array)})."""
    import time
    rank, world = dist.get_rank(), dist.get_world_size()
    b = shard_bounds(n_frames, world)
    lo, hi = b[rank], b[rank + 1]
    counts = [b[r + 1] - b[r] for r in range(world)]
    t0 = time.perf_counter()
    chunks = chunks_fn(lo, hi)
    assert sum(c[1] for c in chunks) == hi - lo and (not chunks or chunks[0][0] == lo)
    # 1-2. detect locally, exchange fixed-size slabs
    slab = np.zeros((hi - lo, MAX_DET, 5), np.float32)
    cnt = np.zeros((hi - lo, 1), np.float32)
    for first, n, handle in chunks:
        for i, d in enumerate(detect_fn(handle, first, n)):
            d = np.asarray(d, np.float32).reshape(-1, 5)[:MAX_DET]
            slab[first - lo + i, : len(d)] = d
            cnt[first - lo + i, 0] = len(d)
    t1 = time.perf_counter()
    packed = np.concatenate([slab.reshape(hi - lo, -1), cnt], axis=1)
    allp = all_gather_ragged(packed, counts, dist, device)
    all_dets = [allp[i, :-1].reshape(MAX_DET, 5)[: int(allp[i, -1])] for i in range(n_frames)]
    t2 = time.perf_counter()
    # 3. identical sequential association + person-box decisions on every rank
    tracks = associate_fn(all_dets)
    t3 = time.perf_counter()
    t_2d = [0.0, 0.0]

    def chunk_of(frame):
        for first, n, handle in chunks:
            if first <= frame < first + n:
                return first, n, handle
        raise IndexError(frame)

    def sharded_topdown(jobs):
        # 4. 2D on the own shard's person-frames (chunk by chunk), then everybody gets every row
        ta = time.perf_counter()
        owner = np.searchsorted(np.asarray(b[1:]), [j[1] for j in jobs], side="right")
        mine = [i for i, r in enumerate(owner) if r == rank]
        rows = np.zeros((len(mine), num_joints, 3), np.float32)
        by_chunk: dict = {}
        for k, i in enumerate(mine):
            by_chunk.setdefault(chunk_of(jobs[i][1])[0], []).append(k)
        for first, n, handle in chunks:
            ks = by_chunk.get(first)
            if not ks:
                continue
            idx = np.array([jobs[mine[k]][1] - first for k in ks], np.int32)
            boxes = np.array([jobs[mine[k]][2] for k in ks], np.float64)
            rows[ks] = np.asarray(topdown_fn(handle, n, idx, boxes), np.float32)
        tb = time.perf_counter()
        per_rank = [int((owner == r).sum()) for r in range(world)]
        got = all_gather_ragged(rows, per_rank, dist, device)
        out = [None] * len(jobs)
        pos = np.concatenate([[0], np.cumsum(per_rank)])
        seen = [0] * world
        for i, r in enumerate(owner):
            out[i] = got[pos[r] + seen[r]]
            seen[r] += 1
        t_2d[0] += tb - ta
        t_2d[1] += time.perf_counter() - tb
        return out

    ps = PersonStreams(num_joints, pad, src_hw, sharded_topdown, lift_fn, max_persons=max_persons, keep_tracks=keep_tracks)
    ps.ingest(tracks)
    out = ps.advance(final=True)        # 5. lifting of every track, replicated
    t4 = time.perf_counter()
    if timings is not None:
        timings.update(detect=t1 - t0, gather_dets=t2 - t1, associate=t3 - t2, topdown=t_2d[0], gather_2d=t_2d[1],
                       lift=t4 - t3 - t_2d[0] - t_2d[1], total=t4 - t0)
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
