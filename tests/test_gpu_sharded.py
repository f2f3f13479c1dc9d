"""The frame-sharded cascade (posepipeline_amd/parallel.py) with the REAL GPU stages: two ranks (gloo rendezvous, both on
GPU 0 -- the rehearsal mode of a box with fewer GPUs than ranks) process one clip; every rank's result must equal the
single-process streamed `Cascade` bit for bit, ids included.  Rank 1 builds its programs from WRONG weights and
receives rank 0's blobs as device tensors (`broadcast_blob_fn` -> `pp_net_create_mem`): the comparison only holds if the
broadcast weights are the ones the kernels read."""
import os
import socket
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

H, W, N, CHUNK = 135, 240, 22, 4


def _clip():
    from tests.test_gpu_detector import synth_frame
    rng = np.random.default_rng(17)
    frames = np.stack([synth_frame(rng, H, W) for _ in range(N)])
    gt = []
    for t in range(N):
        rows = [[30 + 3.0 * t, 15, 100 + 3.0 * t, 118, 0.9], [150 - 2.0 * t, 30, 215 - 2.0 * t, 125, 0.8]]
        if t == 9:
            rows = rows[1:]                    # person 0 missed once: new id, fills on both sides of the gap
        if t in (15, 16):
            rows = rows[:1]                    # person 1 missed twice (spans the rank boundary's neighbourhood)
        gt.append(np.array(rows, np.float32))
    return frames, gt


def _state_dicts(seed_shift):
    from posepipeline_amd.models import faster_rcnn as fr, hrnet, synth
    from posepipeline_amd.models import videopose3d as vp3d
    spec = hrnet.HRNetSpec(32, 17, 128, 96)
    return (synth.synth_state_dict(fr.faster_rcnn_param_shapes(), seed=2 + seed_shift), spec,
            synth.synth_state_dict(hrnet.hrnet_param_shapes(spec), seed=1 + seed_shift),
            synth.synth_state_dict(vp3d.videopose3d_param_shapes(vp3d.VideoPose3DSpec()), seed=3 + seed_shift))


def _pack(tracks, k2, k3):
    out = {"ids": np.array([[r[0] for r in fr] + [-1] * (4 - len(fr)) for fr in tracks])}
    for name, d in (("k2", k2), ("k3", k3)):
        for tid, (first, arr) in d.items():
            out[f"{name}_{tid}_first"] = first
            out[f"{name}_{tid}"] = arr
    return out


def _worker(rank, world, port, outdir):
    import torch
    import torch.distributed as dist
    from posepipeline_amd import _lib, parallel
    from posepipeline_amd.cascade import Cascade
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        frames, gt = _clip()
        det_sd, spec, pose_sd, lift_sd = _state_dicts(0 if rank == 0 else 100)      # rank 1: wrong weights, right shapes
        ctx = _lib.Context(0)
        log = []
        cas = Cascade(ctx, det_sd, pose_sd, lift_sd, H, W, chunk=CHUNK, max_persons=3, pose_spec=spec,
                      blob_fn=parallel.broadcast_blob_fn(dist, torch.device("cuda", 0), backend="gloo", log=log))
        assert [n for n, _, _ in log] == ["det_a", "det_b", "pose", "lift"]
        log.clear()                                                                   # pp_net_create_mem copied the blobs
        b = parallel.shard_bounds(N, world)
        lo, hi = b[rank], b[rank + 1]
        dptr = ctx.malloc(frames[lo:hi].nbytes)
        ctx.h2d(dptr, frames[lo:hi])                                                  # the rank's shard stays resident
        fb = H * W * 3

        def chunks_fn(lo_, hi_):
            assert (lo_, hi_) == (lo, hi)
            return [(f, min(CHUNK, hi - f), dptr + (f - lo) * fb) for f in range(lo, hi, CHUNK)]

        tm = {}
        res = parallel.process_video_sharded(dist, N, chunks_fn, *parallel.cascade_stages(cas, lambda first, n: gt[first:first + n]),
                                             src_hw=(H, W), max_persons=3, timings=tm)
        # the detector really ran on this rank's frames with the BROADCAST weights: compare one frame's own detections
        own = cas.detector.run(frames[lo:lo + 1])[0]
        np.savez(os.path.join(outdir, f"r{rank}.npz"), own_det=own, total=tm["total"], **_pack(res["tracks"], res["keypoints"], res["keypoints_3d"]))
        ctx.free(dptr)
    finally:
        dist.destroy_process_group()


def test_two_ranks_equal_single_process_cascade(ctx):
    import torch.multiprocessing as mp
    from oracle import detector as odet
    from posepipeline_amd import parallel
    from posepipeline_amd.cascade import Cascade, collect
    frames, gt = _clip()
    det_sd, spec, pose_sd, lift_sd = _state_dicts(0)
    cas = Cascade(ctx, det_sd, pose_sd, lift_sd, H, W, chunk=CHUNK, max_persons=3, pose_spec=spec)
    outs = [cas.step(frames[i:i + CHUNK], replay=gt[i:i + CHUNK]) for i in range(0, N, CHUNK)] + [cas.flush()]
    ref = _pack([fr for o in outs for fr in o["tracks"]], collect(outs, "keypoints"), collect(outs, "keypoints_3d"))
    assert sum(k.startswith("k3_") and not k.endswith("first") for k in ref) >= 4        # re-identified persons: >= 4 tracks
    cas.close()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    model = odet.FasterRCNNRef(det_sd)
    b = parallel.shard_bounds(N, 2)
    with tempfile.TemporaryDirectory() as d:
        mp.get_context("spawn")
        mp.spawn(_worker, args=(2, port, d), nprocs=2, join=True)
        for r in range(2):
            g = np.load(os.path.join(d, f"r{r}.npz"))
            assert sorted(k for k in g.files if k not in ("own_det", "total")) == sorted(ref)
            for k, v in ref.items():
                assert np.array_equal(g[k], v), (r, k)                                    # ids, 2D, 3D: bit for bit
            assert np.array_equal(g["own_det"], odet.detect(model, frames[b[r]][:, :, ::-1])), r


# ---- the same at the bench's size and numerics: 1080p, HRNet-W48, DEFAULT (split) kernels, one chunk per rank -------------------
H2, W2, N2, CHUNK2 = 1080, 1920, 8, 4


def _clip_1080p():
    from tests.test_gpu_detector import synth_frame
    rng = np.random.default_rng(29)
    frames = np.stack([synth_frame(rng, H2, W2) for _ in range(N2)])
    gt = []
    for t in range(N2):
        rows = [[300 + 20.0 * t, 150, 480 + 20.0 * t, 700, 0.9], [1200 - 15.0 * t, 260, 1370 - 15.0 * t, 760, 0.8]]
        if t == 4:
            rows = rows[:1]                    # person 1 missed in the first frame of rank 1's shard: fills across the rank boundary
        gt.append(np.array(rows, np.float32))
    return frames, gt


def _state_dicts_1080p(seed_shift):
    from posepipeline_amd.models import faster_rcnn as fr, hrnet, synth
    from posepipeline_amd.models import videopose3d as vp3d
    spec = hrnet.hrnet_w48_384x288()
    return (synth.synth_state_dict(fr.faster_rcnn_param_shapes(), seed=2 + seed_shift), spec,
            synth.synth_state_dict(hrnet.hrnet_param_shapes(spec), seed=1 + seed_shift),
            synth.synth_state_dict(vp3d.videopose3d_param_shapes(vp3d.VideoPose3DSpec()), seed=3 + seed_shift))


def _worker_1080p(rank, world, port, outdir):
    os.environ["POSEPIPE_CONV_EXACT"] = "0"            # the library's default numerics in this process (read when nets are created)
    import torch
    import torch.distributed as dist
    from posepipeline_amd import _lib, parallel
    from posepipeline_amd.cascade import Cascade
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        frames, gt = _clip_1080p()
        det_sd, spec, pose_sd, lift_sd = _state_dicts_1080p(0 if rank == 0 else 100)
        ctx = _lib.Context(0)
        with _lib.default_numerics("split"):
            cas = Cascade(ctx, det_sd, pose_sd, lift_sd, H2, W2, chunk=CHUNK2, max_persons=3, pose_spec=spec,
                          blob_fn=parallel.broadcast_blob_fn(dist, torch.device("cuda", 0), backend="gloo"))
        assert cas.pose_net.numerics == "split" and cas.detector.net_a.numerics == "split"
        b = parallel.shard_bounds(N2, world)
        lo, hi = b[rank], b[rank + 1]
        fb = H2 * W2 * 3
        dptr = ctx.malloc(CHUNK2 * fb)                  # ONE chunk buffer: the shard is streamed through it, once per pass
        uploads = []

        def chunks_fn(lo_, hi_):
            for f in range(lo_, hi_, CHUNK2):
                n = min(CHUNK2, hi_ - f)
                ctx.h2d(dptr, frames[f:f + n])
                uploads.append(f)
                yield f, n, dptr

        tm = {}
        res = parallel.process_video_sharded(dist, N2, chunks_fn, *parallel.cascade_stages(cas, lambda first, n: gt[first:first + n]),
                                             src_hw=(H2, W2), max_persons=3, timings=tm)
        assert uploads == [lo, lo] and tm["rounds"] == 1                                   # one chunk per rank, two passes
        np.savez(os.path.join(outdir, f"r{rank}.npz"), **_pack(res["tracks"], res["keypoints"], res["keypoints_3d"]))
        ctx.free(dptr)
    finally:
        dist.destroy_process_group()


def test_two_ranks_1080p_default_numerics(ctx):
    """configs[3] as the bench shards it, in the numerics the bench times: 8 frames of 1080p, 2 ranks x 1 chunk of 4 frames,
    HRNet-W48 384x288, split kernels; rank 1 starts from wrong weights and receives rank 0's.  Every rank's ids, 2D and 3D
    equal the single-process streamed Cascade (created with the same numerics) bit for bit -- per-element results of the
    split kernels do not depend on batch composition either."""
    import torch.multiprocessing as mp
    from posepipeline_amd import _lib
    from posepipeline_amd.cascade import Cascade, collect
    frames, gt = _clip_1080p()
    det_sd, spec, pose_sd, lift_sd = _state_dicts_1080p(0)
    with _lib.default_numerics("split"):
        cas = Cascade(ctx, det_sd, pose_sd, lift_sd, H2, W2, chunk=CHUNK2, max_persons=3, pose_spec=spec)
    outs = [cas.step(frames[i:i + CHUNK2], replay=gt[i:i + CHUNK2]) for i in range(0, N2, CHUNK2)] + [cas.flush()]
    ref = _pack([fr for o in outs for fr in o["tracks"]], collect(outs, "keypoints"), collect(outs, "keypoints_3d"))
    assert sum(k.startswith("k3_") and not k.endswith("first") for k in ref) >= 3           # the missed person returns under a new id
    cas.close()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker_1080p, args=(2, port, d), nprocs=2, join=True)
        for r in range(2):
            g = np.load(os.path.join(d, f"r{r}.npz"))
            assert sorted(g.files) == sorted(ref)
            for k, v in ref.items():
                assert np.array_equal(g[k], v), (r, k)


def test_bench_gpus2_launches_two_ranks_on_real_kernels():
    """`python bench.py --gpus 2` (no torchrun: the form the driver may use) on the real cascade: two ranks -- sharing this box's one GPU,
    collectives over gloo -- each time their own chunks; rank 0 prints ONE JSON line with n_gpus = 2, the process group's world size and
    both ranks' step times.  (The RCCL runs differ only in the backend string and in one GPU per rank.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, POSEPIPE_DIST_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "POSEPIPE_CONV_EXACT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--chunk", "8",
                          "--cpu-frames", "0"], capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["distributed"]["world_size"] == 2 and len(line["distributed"]["ranks"]) == 2
    assert line["value"] > 0 and len(line["per_rank_ms_per_step"]) == 2 and line["scaling"] == "weak"
    assert line["config"]["frames_per_step_per_gpu"] == 8 and "secondary" not in line      # side legs are N = 1 only
    assert len(line["per_rank_roofline_frac"]) == 2 and all(0 < f < 1 for f in line["per_rank_roofline_frac"])
    assert line["collectives_in_timed_region"]["count"] == 0 and line["distributed"]["backend"]


def test_bench_gpus2_shard_mode_on_real_kernels():
    """`python bench.py --gpus 2 --mode shard`: ONE clip sharded over two ranks through parallel.py on the real
    cascade (gloo here, two ranks on this box's one GPU): one JSON line whose per-rank block carries what makes the first real
    multi-GPU run readable -- phase times, each rank's end-to-end roofline fraction, collective count / bytes / blocked time."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, POSEPIPE_DIST_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "POSEPIPE_CONV_EXACT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--mode", "shard", "--steps", "2", "--warmup", "1",
                          "--chunk", "8", "--cpu-frames", "0"], capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["mode"] == "shard" and line["distributed"]["world_size"] == 2
    assert line["value"] > 0 and line["config"]["frames_per_step_per_gpu"] == 8
    ranks = line["per_rank_phase_ms"]
    assert len(ranks) == 2
    for r in ranks:
        # two passes (detections, 2D rows) x `steps` rounds of one all_gather each, 2 ranks x slab bytes received
        assert r["rounds"] == 2 and r["collectives"] >= 4 and r["collective_bytes"] > 0 and r["collective_wait"] >= 0
        assert 0 < r["roofline_frac"] < 1 and r["total"] > 0
    assert line["distributed"]["backend"] == "gloo"
