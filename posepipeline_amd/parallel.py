"""One long video across N ranks (one process per GPU, torch.distributed; NCCL = RCCL on MI355X, gloo on CPU).

The reference's only parallel mode is job-level: N OS processes take different *video keys* through
`populate(reserve_jobs=True)` (utils/standard_pipelines.py:40,46,100) -- pure replicas, which need nothing
from this module.  This module adds what the reference cannot do: sharding the FRAMES of one video
(SURVEY.md 8e).  Detection and the 2D stage are independent per frame; only association is sequential in
time, and it costs microseconds per frame on the host, so:

  1. rank r owns the contiguous frames [b_r, b_{r+1}) and detects on them;
  2. all ranks exchange fixed-size detection slabs (100 x 5 floats + a count per frame) with ONE
     all_gather -- the only data-path collective, ~2 KB per frame;
  3. every rank runs the identical sequential association over all frames (deterministic host code, so
     no broadcast of the result is needed) and selects the person boxes of its own frames;
  4. rank r runs the top-down 2D stage on its frames; a second all_gather (17 x 3 floats per frame)
     gives every rank the whole 2D track;
  5. rank r lifts its own frames to 3D using the gathered track as temporal context (receptive field 243)
     and rank 0 gathers the 3D joints.
Weights travel once per process start through `broadcast_blob`.

The compute stages are passed in as callables so that the orchestration is testable on CPU with gloo and
stub stages (tests/test_distributed_gloo.py); bench.py / the cascade pass the GPU stages.
"""
from __future__ import annotations

import numpy as np

MAX_DET = 100


def shard_bounds(n: int, world: int):
    """contiguous, near-equal frame ranges: [b_0 = 0, ..., b_world = n]"""
    return [(n * r) // world for r in range(world + 1)]


def _to_tensor(a: np.ndarray, device):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def broadcast_blob(blob: np.ndarray, dist, device="cpu", src=0) -> np.ndarray:
    """weights: one broadcast from `src`; every rank returns the same float32 array"""
    import torch
    t = _to_tensor(blob.astype(np.float32), device) if dist.get_rank() == src else torch.empty(blob.size, dtype=torch.float32, device=device)
    dist.broadcast(t, src=src)
    return t.cpu().numpy()


def all_gather_ragged(local: np.ndarray, counts, dist, device="cpu") -> np.ndarray:
    """all_gather of per-rank arrays whose leading dim differs (counts[r] rows on rank r): pad to the max,
    one collective, trim.  Returns the concatenation in rank order."""
    import torch
    world = dist.get_world_size()
    m = max(counts)
    pad = np.zeros((m,) + local.shape[1:], dtype=local.dtype)
    pad[: local.shape[0]] = local
    t = _to_tensor(pad, device)
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t)
    return np.concatenate([o.cpu().numpy()[: counts[r]] for r, o in enumerate(outs)], axis=0)


def process_video_sharded(dist, n_frames, read_frames, detect_fn, associate_fn, topdown_fn, lift_fn, device="cpu",
                          num_joints=17):
    """Run the cascade on frames [0, n_frames) sharded over the ranks of `dist`.

    read_frames(lo, hi)            -> this rank's frames (any object the stage callables understand)
    detect_fn(frames)              -> list (per frame) of [n][5] float32 (x1, y1, x2, y2, score), n <= 100
    associate_fn(dets_all_frames)  -> (bbox [N][4] float64 TLWH with NaN rows, tracks per frame); sequential
    topdown_fn(frames, bbox_rows)  -> [k][J][3] keypoints for this rank's frames
    lift_fn(kp2d_context, lo, hi)  -> [hi-lo][J][3] 3D joints of frames [lo, hi) given the WHOLE 2D track
    Returns on every rank: dict(tracks, bbox, keypoints [N][J][3], keypoints_3d [N][J][3])."""
    rank, world = dist.get_rank(), dist.get_world_size()
    b = shard_bounds(n_frames, world)
    lo, hi = b[rank], b[rank + 1]
    counts = [b[r + 1] - b[r] for r in range(world)]
    frames = read_frames(lo, hi)
    # 1-2. detect locally, exchange fixed-size slabs
    dets = detect_fn(frames)
    slab = np.zeros((hi - lo, MAX_DET, 5), np.float32)
    cnt = np.zeros((hi - lo, 1), np.float32)
    for i, d in enumerate(dets):
        d = np.asarray(d, np.float32).reshape(-1, 5)[:MAX_DET]
        slab[i, : len(d)] = d
        cnt[i, 0] = len(d)
    packed = np.concatenate([slab.reshape(hi - lo, -1), cnt], axis=1)
    allp = all_gather_ragged(packed, counts, dist, device)
    all_dets = [allp[i, :-1].reshape(MAX_DET, 5)[: int(allp[i, -1])] for i in range(n_frames)]
    # 3. identical sequential association on every rank
    bbox, tracks = associate_fn(all_dets)
    # 4. 2D on the own shard, gather the whole track
    kp_local = np.asarray(topdown_fn(frames, bbox[lo:hi]), np.float64).reshape(hi - lo, num_joints, 3)
    kp = all_gather_ragged(kp_local, counts, dist, device)
    # 5. 3D on the own shard with the whole track as context, gather
    k3_local = np.asarray(lift_fn(kp, lo, hi), np.float64).reshape(hi - lo, -1, 3)
    k3 = all_gather_ragged(k3_local, counts, dist, device)
    return dict(tracks=tracks, bbox=bbox, keypoints=kp, keypoints_3d=k3)
