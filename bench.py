#!/usr/bin/env python
"""bench.py -- throughput of the MI355X hot path, one JSON line on rank 0.

Workloads (config.workload):
  c2  BASELINE.json configs[1]: HRNet-W32 256x192 top-down 2D keypoints on 64 pre-cropped persons per
      step (flip_test: 128 backbone samples), flip-merge + decode, keypoints back on the host.
      Inputs (normalised crops) are resident in HBM when the timed region starts.
One process per GPU (torchrun); ranks work on independent frame batches (weak scaling, no data-path
collective); weights are generated on rank 0 and broadcast over RCCL.  `value` = frames of all ranks /
max-over-ranks wall time of exactly K steps bracketed by barrier + synchronize.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--batch", type=int, default=64, help="person-frames per step per GPU")
    ap.add_argument("--cpu-frames", type=int, default=6, help="frames of the CPU-baseline sample (0 = skip)")
    return ap.parse_args()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from posepipeline_amd import _lib, ops
    from posepipeline_amd.models import hrnet, synth
    from posepipeline_amd.program import Net

    ctx = _lib.Context(local_rank)
    spec = hrnet.hrnet_w32_256x192()
    shapes = hrnet.hrnet_param_shapes(spec)
    sd = synth.synth_state_dict(shapes, seed=1)
    prog = hrnet.build_hrnet_program(spec, sd)
    if world > 1:
        # weights: one RCCL broadcast from rank 0 (every rank then owns the same resident blob)
        import torch
        blob = torch.from_numpy(prog.blob).cuda() if rank == 0 else torch.empty(prog.blob.size, dtype=torch.float32, device="cuda")
        dist.broadcast(blob, src=0)
        prog.blob = blob.cpu().numpy()
        del blob
    n = args.batch
    net = Net(ctx, prog, max_batch=2 * n)
    td = ops.TopDown(net, num_joints=17, flip_perm=hrnet.flip_perm(17), post="default")
    # BASELINE configs[1] names W32 256x192 -> mmpose's W32 configs decode with post_process='default' 

    rng = np.random.default_rng(1000 + rank)                      # config index 1, per-rank shard
    x = np.zeros((n, spec.in_h, spec.in_w, 4), np.float32)
    x[..., :3] = rng.standard_normal((n, spec.in_h, spec.in_w, 3)).astype(np.float32)
    cs = np.tile(np.array([[96.0, 128.0, 192 / 200 * 1.25, 256 / 200 * 1.25]], np.float32), (n, 1))
    dptr, nbytes, _ = net.buffer("input")
    ctx.h2d(dptr, x)                                              # inputs resident in HBM before timing

    def step():
        return td.run_precropped(dptr, cs, n=n)

    def barrier():
        if dist is not None:
            dist.barrier()
        ctx.synchronize()

    for _ in range(args.warmup):
        kp = step()
    barrier()
    t0 = time.perf_counter()
    t_net = t_pre = t_dec = 0.0
    for _ in range(args.steps):
        kp = step()
        a, b, c = td.timing()
        t_pre += a
        t_net += b
        t_dec += c
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        import torch
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        n_launch = len(prog.ops)
        flops_step = prog.flops * 2 * n                          # algorithmic conv FLOPs: flip doubles the samples
        net_ms = t_net / args.steps
        achieved = flops_step / (net_ms * 1e-3) / 1e12
        out = {
            "metric": "frames/sec (whole node), detect->2D->3D cascade on 1080p; MPJPE vs reference",
            "value": world * n * args.steps / dt,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "configs[1]: HRNet-W32 256x192 top-down 2D, pre-cropped persons, flip_test, decode 'default'",
                       "frames_per_step_per_gpu": n, "backbone_samples_per_step": 2 * n,
                       "stages_timed": "mirror copy + backbone (fp32 MFMA) + flip-merge/decode + keypoints D2H",
                       "not_in_this_line": "detector, tracker, 3D lifting (cascade bench lands when those stages do)"},
            "roofline": {"bound": "mfma", "kernel": "conv_igemm_kernel (all %d conv launches of the backbone program)" % n_launch,
                         "achieved": achieved, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP32_MFMA_PEAK_TFLOPS, "traffic": None,
                         "flops_per_launch": flops_step / n_launch, "avg_launch_ms": net_ms / n_launch,
                         "stage_ms": {"pre": t_pre / args.steps, "backbone": net_ms, "decode": t_dec / args.steps}},
        }
        if args.cpu_frames > 0:
            out["cpu_baseline"] = cpu_baseline(sd, x[: args.cpu_frames], cs[: args.cpu_frames], kp[: args.cpu_frames])
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(sd, x, cs, kp_gpu):
    """CPU restatement of the reference wrapper path (oracle/), per-frame batch-1 loop like
    pose_pipeline/wrappers/mmpose.py:60-76, OpenMP over all host cores."""
    from oracle import clib
    from oracle import decode as odec
    from oracle import nets as onets
    from posepipeline_amd.models import hrnet
    model = onets.HRNetRef(sd, 32)
    clib.lib()
    t0 = time.perf_counter()
    kps = []
    for i in range(x.shape[0]):
        img = np.ascontiguousarray(np.transpose(x[i:i + 1, :, :, :3], (0, 3, 1, 2)))
        hm = model.forward(img)
        hmf = model.forward(np.ascontiguousarray(img[:, :, :, ::-1]))
        k, _ = odec.decode_topdown(hm, hmf, hrnet.COCO_FLIP_PAIRS, cs[i:i + 1, :2], cs[i:i + 1, 2:], post_process="default")
        kps.append(k[0])
        if time.perf_counter() - t0 > 20.0:       # bounded sample: stop after ~20 s of CPU work
            break
    dt = time.perf_counter() - t0
    m = len(kps)
    err = float(np.abs(np.array(kps) - kp_gpu[:m]).max())
    return {"value": m / dt, "unit": "frames/s", "cores": clib.N_THREADS, "kind": "port",
            "sample": "%d of the same pre-cropped frames, batch-1 loop, C/OpenMP fmaf-chain convs + numpy decode (%.1f s)" % (m, dt),
            "max_abs_diff_px_vs_gpu": err}


if __name__ == "__main__":
    main()
