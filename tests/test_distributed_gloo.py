"""N > 1 path on CPU: world_size-2 gloo processes run the frame-sharded cascade orchestration
(posepipeline_amd/parallel.py) with the real host stages (C++ SORT tracker, PersonStreams box decisions) and stub
compute stages; the result must equal the single-process run -- sharded over one rank, and streamed chunk by chunk the
way cascade.Cascade does it -- bit for bit.  The same orchestration with the GPU stages: tests/test_gpu_sharded.py."""
import os
import socket
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from posepipeline_amd import parallel
from posepipeline_amd.tracking import Tracker

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
        tracks.append([(int(i), *rows[j]) for i, j in zip(ids, info[:, 1])])
    return tracks


def fake_rows(frame_ids, boxes):
    kp = np.zeros((len(frame_ids), 17, 3), np.float32)
    j = np.arange(17)
    for i, (t, bb) in enumerate(zip(frame_ids, boxes)):
        kp[i, :, 0] = np.float32(bb[0]) + np.float32(bb[2]) * (j % 4).astype(np.float32) / np.float32(4) + np.float32(t)
        kp[i, :, 1] = np.float32(bb[1]) + np.float32(bb[3]) * (j // 4).astype(np.float32) / np.float32(8)
        kp[i, :, 2] = np.float32(0.5) + np.float32(0.01) * j.astype(np.float32)
    return kp


PAD = 6
SRC = (1024, 2048)


def fake_lift(kn):
    """a function of the whole +-PAD window of every frame (edge-clamped inside the passed context, like pp_videopose3d_lift)"""
    n = kn.shape[0]
    p = np.pad(kn.astype(np.float32), ((PAD, PAD), (0, 0), (0, 0)), mode="edge")
    w = np.stack([p[i:i + 2 * PAD + 1] for i in range(n)]).astype(np.float64)
    out = np.zeros((n, 17, 3))
    out[:, :, :2] = w.mean(axis=1)
    out[:, :, 2] = w[:, 0, :, 0] - w[:, -1, :, 1]
    return out.astype(np.float32)


def chunks_fn(lo, hi, size=5):
    return [(f, min(size, hi - f), np.arange(f, min(f + size, hi))) for f in range(lo, hi, size)]


def run(d, timings=None):
    return parallel.process_video_sharded(
        d, N_FRAMES, chunks_fn, lambda handle, first, n: fake_detect(handle), associate,
        lambda handle, n, idx, boxes: fake_rows(handle[idx], boxes), fake_lift, SRC, pad=PAD, max_persons=3, timings=timings)


def run_streamed(chunk):
    """the single-process form: what cascade.Cascade does chunk after chunk with the same stages"""
    from posepipeline_amd.person_stream import PersonStreams, collect
    trk = Tracker(mode=1, match_iou_thr=0.5, obj_score_thr=0.5)
    ps = PersonStreams(17, PAD, SRC, lambda jobs: fake_rows([j[1] for j in jobs], [j[2] for j in jobs]), fake_lift, max_persons=3)
    outs = []
    for f in range(0, N_FRAMES, chunk):
        rows_all = []
        for rows in fake_detect(np.arange(f, min(f + chunk, N_FRAMES))):
            ids, _, info = trk.step(rows[:, :4].astype(np.float64), rows[:, 4].astype(np.float64))
            rows_all.append([(int(i), *rows[j]) for i, j in zip(ids, info[:, 1])])
        ps.ingest(rows_all)
        outs.append(ps.advance())
    outs.append(ps.advance(final=True))
    return collect(outs, "keypoints"), collect(outs, "keypoints_3d")


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


def _pack(res):
    out = {}
    for what in ("keypoints", "keypoints_3d"):
        for tid, (first, arr) in res[what].items():
            out[f"{what}_{tid}_first"] = first
            out[f"{what}_{tid}"] = arr
    out["ids"] = np.array([[r[0] for r in fr] + [-1] * (8 - len(fr)) for fr in res["tracks"]])
    return out


def _worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tm = {}
        res = run(dist, tm)
        assert set(tm) >= {"detect", "gather_dets", "associate", "topdown", "gather_2d", "lift", "total"}
        blob = np.arange(1000, dtype=np.float32) * (1 if rank == 0 else -1)
        got = parallel.broadcast_blob(blob, dist)
        got_dev = parallel.broadcast_blob_device(blob if rank == 0 else None, 1000, dist, "cpu", backend="gloo").numpy()
        ragged = parallel.all_gather_ragged(np.full((rank + 2, 3), rank, np.float64), [2, 3], dist)
        np.savez(os.path.join(outdir, f"r{rank}.npz"), blob=got, blob_dev=got_dev, ragged=ragged, **_pack(res))
    finally:
        dist.destroy_process_group()


def test_sharded_video_equals_single_process():
    ref = _pack(run(LocalDist()))
    # SORT without ReID re-ids a person after a dropout: several ids per person, short tracks with fills at both ends
    tids = sorted(int(k.split("_")[2]) for k in ref if k.startswith("keypoints_3d_") and not k.endswith("first"))
    assert len(tids) >= 4
    # the sharded whole-clip form equals the streamed single-process form (what Cascade does), for any chunking
    for chunk in (1, 4, 37):
        k2, k3 = run_streamed(chunk)
        assert sorted(k3) == tids
        for tid in tids:
            assert k2[tid][0] == ref[f"keypoints_{tid}_first"] and np.array_equal(k2[tid][1], ref[f"keypoints_{tid}"])
            assert k3[tid][0] == ref[f"keypoints_3d_{tid}_first"] and np.array_equal(k3[tid][1], ref[f"keypoints_3d_{tid}"])
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, port, d), nprocs=2, join=True)
        for r in range(2):
            g = np.load(os.path.join(d, f"r{r}.npz"))
            assert np.array_equal(g["ids"], ref["ids"])                                  # track ids bit-exact
            for k, v in ref.items():
                assert np.array_equal(g[k], v), k                                        # 2D and 3D of every track, bit for bit
            assert np.array_equal(g["blob"], np.arange(1000, dtype=np.float32))          # rank 0's weights everywhere
            assert np.array_equal(g["blob_dev"], np.arange(1000, dtype=np.float32))
            assert np.array_equal(g["ragged"], np.array([[0.0] * 3] * 2 + [[1.0] * 3] * 3))


def _worker_rounds(rank, world, port, outdir):
    """chunk size 6 over shards of 18 / 19 frames: 3 rounds on rank 0, 4 on rank 1 (rank 0 joins the last round with an empty
    slab); chunks come from a LAZY generator -- nothing but the chunk in work exists -- which is consumed exactly twice"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        events, calls, needs = [], [], []

        def lazy_chunks(lo, hi, need=None):
            calls.append((lo, hi))
            needs.append(need)

            def gen():
                for i, f in enumerate(range(lo, hi, 6)):
                    if need is not None and not need[i]:      # the 2D pass has no person-frame in this chunk: not read, not uploaded
                        events.append(("skip", f))
                        yield f, min(6, hi - f), None
                        continue
                    events.append(("upload", f))
                    yield f, min(6, hi - f), np.arange(f, min(f + 6, hi))
            return gen()

        def detect(handle, first, n):
            events.append(("detect", first))
            return fake_detect(handle)

        def rows(handle, n, idx, boxes):
            events.append(("2d", int(handle[0])))
            return fake_rows(handle[idx], boxes)
        tm = {}
        res = parallel.process_video_sharded(dist, N_FRAMES, lazy_chunks, detect, associate, rows, fake_lift, SRC, pad=PAD,
                                             max_persons=3, timings=tm)
        b = parallel.shard_bounds(N_FRAMES, world)
        assert calls == [(b[rank], b[rank + 1])] * 2 and tm["rounds"] == 4
        # the second pass told the source which chunks it needs (one flag per chunk of the first pass); a chunk it does not need
        # was neither uploaded again nor handed to the 2D stage
        n_chunks = len(range(b[rank], b[rank + 1], 6))
        assert needs[0] is None and len(needs[1]) == n_chunks
        second = events[[i for i, e in enumerate(events) if e[0] == "detect"][-1] + 1:]
        assert {e[1] for e in second if e[0] == "skip"} == {b[rank] + 6 * i for i in range(n_chunks) if not needs[1][i]}
        assert not {e[1] for e in second if e[0] == "2d"} & {e[1] for e in second if e[0] == "skip"}
        # a chunk is requested only after the previous one was processed (bounded residency): uploads and work alternate
        det = [e for e in events if e[0] in ("upload", "detect")][: 2 * len(range(b[rank], b[rank + 1], 6))]
        assert [e[0] for e in det[:2]] == ["upload", "detect"]
        for i in range(1, len(det) - 1, 2):
            assert det[i][0] == "detect" and det[i + 1][0] == "upload" and det[i + 1][1] == det[i][1] + 6
        np.savez(os.path.join(outdir, f"r{rank}.npz"), **_pack(res))
    finally:
        dist.destroy_process_group()


def test_sharded_rounds_with_lazy_chunks_and_ragged_round_counts():
    ref = _pack(run(LocalDist()))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker_rounds, args=(2, port, d), nprocs=2, join=True)
        for r in range(2):
            g = np.load(os.path.join(d, f"r{r}.npz"))
            assert sorted(g.files) == sorted(ref)
            for k, v in ref.items():
                assert np.array_equal(g[k], v), k


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


def test_bench_gpus_flag_launches_that_many_ranks():
    """`python bench.py --gpus 2 ...` -- the form the driver uses WITHOUT torchrun -- must produce 2 ranks (round 3's parsed the
    flag and ran one): bench.py re-launches itself under torch.distributed.run, every rank joins the process group, and the line
    reports the world size the group has.  Rehearsed with the gloo backend and the kernel-free `plumbing` workload (no GPU here);
    the RCCL runs take the same path with backend "nccl"."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, POSEPIPE_DIST_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--workload",
                          "plumbing"], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout                                  # ONE JSON line, from rank 0
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["distributed"]["world_size"] == 2 and line["distributed"]["backend"] == "gloo"
    assert sorted(r["rank"] for r in line["distributed"]["ranks"]) == [0, 1]
    assert len({r["pid"] for r in line["distributed"]["ranks"]}) == 2  # two processes
    assert line["weights_identical_on_every_rank"] and line["slowest_rank_clock"] == 2.0
    # a launcher that started another number of ranks than --gpus names is an error, not a mislabelled line
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    bad = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "4", "--workload", "plumbing"],
                         capture_output=True, text=True, env=env, timeout=300)
    assert bad.returncode != 0 and "--gpus 4 but the launcher started 2" in bad.stdout + bad.stderr
    # default: one rank, no process group
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "plumbing"], capture_output=True, text=True, env=env, timeout=300)
    line = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][0])
    assert line["n_gpus"] == 1 and line["distributed"]["world_size"] == 1 and line["distributed"]["backend"] is None


def test_single_rank_bench_never_imports_torch():
    """N = 1 is what the driver times on a fresh box, where a cold `import torch` costs 1 - 2 minutes: neither the rank plumbing nor
    the one-rank stand-in of the sharded mode (parallel._Gather with a `numpy_only` dist) may pull it in"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import runpy, sys; sys.argv = ['bench.py', '--workload', 'plumbing']; "
            f"runpy.run_path({os.path.join(root, 'bench.py')!r}, run_name='__main__'); "
            "assert 'torch' not in sys.modules, 'bench.py imported torch at N = 1'\n"
            f"sys.path.insert(0, {root!r})\n"
            "import numpy as np\n"
            "from posepipeline_amd import parallel\n"
            "class D:\n"
            "    numpy_only = True\n"
            "    def get_world_size(self): return 1\n"
            "    def get_rank(self): return 0\n"
            "g = parallel._Gather(D(), 'cpu', np.arange(6, dtype=np.float32).reshape(2, 3))\n"
            "r = g.result(); assert r.shape == (1, 2, 3) and r[0, 1, 2] == 5\n"
            "assert 'torch' not in sys.modules\n")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr


def test_takes_need_follows_wrappers_only_to_a_real_parameter():
    """a **kwargs forwarding wrapper gets `need=` only when what it wraps names the parameter"""
    import functools
    from posepipeline_amd.parallel import _takes_need

    def plain(lo, hi):
        return []

    def lazy(lo, hi, need=None):
        return []

    def forward_plain(*a, **k):
        return plain(*a, **k)

    @functools.wraps(lazy)
    def wrapped_lazy(*a, **k):
        return lazy(*a, **k)

    @functools.wraps(plain)
    def wrapped_plain(*a, **k):
        return plain(*a, **k)

    assert not _takes_need(plain) and _takes_need(lazy)
    assert not _takes_need(forward_plain) and not _takes_need(wrapped_plain)
    assert _takes_need(wrapped_lazy) and _takes_need(functools.partial(lazy, 0)) and not _takes_need(functools.partial(plain, 0))
