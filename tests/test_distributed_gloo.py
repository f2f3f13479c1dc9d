"""N > 1 path on CPU: world_size-2 gloo processes run the frame-sharded cascade orchestration
(posepipeline_amd/parallel.py) with the real host stages (C++ SORT tracker, PersonBbox selection) and stub
compute stages; the result must equal the single-process run bit for bit."""
import os
import socket
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from posepipeline_amd import parallel
from posepipeline_amd.tracking import Tracker, person_bbox

N_FRAMES = 37      # odd: shards of 18 and 19 frames


def fake_detect(frame_ids):
    out = []
    for t in frame_ids:
        rng = np.random.default_rng(1000 + int(t))
        rows = []
        for p in range(3):
            if (t + p) % 11 == 0:
                continue                                   # dropout
            x, y = 100 + 300 * p + 4.0 * t, 50 + 10 * p
            rows.append([x, y, x + 90, y + 260, 0.6 + 0.1 * p])
        rows.append([rng.uniform(0, 1500), rng.uniform(0, 700), 0, 0, 0.3])   # below the tracker threshold
        rows[-1][2], rows[-1][3] = rows[-1][0] + 50, rows[-1][1] + 120
        out.append(np.array(rows, np.float32))
    return out


def associate(all_dets):
    trk = Tracker(mode=1, match_iou_thr=0.5, obj_score_thr=0.5)
    tracks = []
    for rows in all_dets:
        ids, _, info = trk.step(rows[:, :4].astype(np.float64), rows[:, 4].astype(np.float64))
        fr = []
        for i, j in zip(ids, info[:, 1]):
            x = rows[j]
            fr.append({"track_id": int(i), "tlbr": x[:4], "tlhw": np.array([x[0], x[1], x[2] - x[0], x[3] - x[1]]),
                       "confidence": x[4]})
        tracks.append(fr)
    bbox, _ = person_bbox(tracks, [0])
    return bbox, tracks


def fake_topdown(frame_ids, bbox_rows):
    kp = np.zeros((len(frame_ids), 17, 3))
    for i, (t, bb) in enumerate(zip(frame_ids, bbox_rows)):
        if np.isnan(bb).any():
            continue
        j = np.arange(17)
        kp[i, :, 0] = bb[0] + bb[2] * (j % 4) / 4.0 + 0.01 * t
        kp[i, :, 1] = bb[1] + bb[3] * (j // 4) / 5.0
        kp[i, :, 2] = 0.5 + 0.01 * j
    return kp


def fake_lift(kp_all, lo, hi):
    n = kp_all.shape[0]
    out = np.zeros((hi - lo, 17, 3))
    for i, t in enumerate(range(lo, hi)):
        idx = np.clip(np.arange(t - 121, t + 122), 0, n - 1)        # 243-frame edge-replicated window
        out[i, :, :2] = kp_all[idx, :, :2].mean(axis=0)
        out[i, :, 2] = t
    return out


def run(d):
    return parallel.process_video_sharded(d, N_FRAMES, lambda lo, hi: np.arange(lo, hi), fake_detect, associate, fake_topdown,
                                          fake_lift)


class LocalDist:
    """world_size 1 stand-in with the torch.distributed calls parallel.py uses"""
    def get_rank(self):
        return 0

    def get_world_size(self):
        return 1

    def all_gather(self, outs, t):
        outs[0].copy_(t)

    def broadcast(self, t, src=0):
        pass


def _worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res = run(dist)
        blob = np.arange(1000, dtype=np.float32) * (1 if rank == 0 else -1)
        got = parallel.broadcast_blob(blob, dist)
        ragged = parallel.all_gather_ragged(np.full((rank + 2, 3), rank, np.float64), [2, 3], dist)
        np.savez(os.path.join(outdir, f"r{rank}.npz"), bbox=res["bbox"], kp=res["keypoints"], k3=res["keypoints_3d"],
                 ids=np.array([[t["track_id"] for t in fr] + [-1] * (8 - len(fr)) for fr in res["tracks"]]), blob=got, ragged=ragged)
    finally:
        dist.destroy_process_group()


def test_sharded_video_equals_single_process():
    ref = run(LocalDist())
    # SORT without ReID re-ids a person after a dropout, so track 0 covers the frames up to its first miss
    assert np.isnan(ref["bbox"]).any() and (~np.isnan(ref["bbox"]).any(axis=1)).sum() >= 10
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, port, d), nprocs=2, join=True)
        ref_ids = np.array([[t["track_id"] for t in fr] + [-1] * (8 - len(fr)) for fr in ref["tracks"]])
        for r in range(2):
            g = np.load(os.path.join(d, f"r{r}.npz"))
            assert np.array_equal(g["ids"], ref_ids)                                     # track ids bit-exact
            assert np.array_equal(np.nan_to_num(g["bbox"]), np.nan_to_num(ref["bbox"]))
            assert np.array_equal(g["kp"], ref["keypoints"])
            assert np.array_equal(g["k3"], ref["keypoints_3d"])
            assert np.array_equal(g["blob"], np.arange(1000, dtype=np.float32))          # rank 0's weights everywhere
            assert np.array_equal(g["ragged"], np.array([[0.0] * 3] * 2 + [[1.0] * 3] * 3))


def test_shard_bounds():
    assert parallel.shard_bounds(37, 2) == [0, 18, 37]
    assert parallel.shard_bounds(300, 8)[-1] == 300 and len(parallel.shard_bounds(300, 8)) == 9
    assert parallel.shard_bounds(3, 8) == [0, 0, 0, 1, 1, 1, 2, 2, 3]


def test_bench_dist_helpers_world2_gloo(tmp_path):
    """bench.py's N > 1 plumbing (weight broadcast, barrier, max-over-ranks timing) under torchrun with 2 gloo ranks: the
    same Dist class the RCCL runs use (POSEPIPE_DIST_BACKEND=gloo keeps the tensors on the CPU)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "dist_probe.py"
    script.write_text(
        "import json, sys\n"
        f"sys.path.insert(0, {root!r})\n"
        "import numpy as np\n"
        "import bench\n"
        "class Ctx:\n"
        "    def synchronize(self):\n"
        "        pass\n"
        "D = bench.Dist()\n"
        "blob = np.arange(1000, dtype=np.float32) * (1.0 if D.rank == 0 else -1.0)\n"
        "got = D.bcast_blob(blob)\n"
        "D.barrier(Ctx())\n"
        "t = D.max_time(1.0 + D.rank)\n"
        "ok = bool(np.array_equal(got, np.arange(1000, dtype=np.float32)))\n"
        f"open({str(tmp_path)!r} + '/rank%d.json' % D.rank, 'w').write(json.dumps(dict(rank=D.rank, world=D.world, ok=ok, t=t)))\n"
        "D.close()\n")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, POSEPIPE_DIST_BACKEND="gloo")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", str(port), str(script)], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    import json
    res = [json.loads((tmp_path / f"rank{r}.json").read_text()) for r in (0, 1)]
    assert all(r["ok"] and r["world"] == 2 for r in res)
    assert res[0]["t"] == res[1]["t"] == 2.0          # every rank sees the slowest rank's time
