#!/usr/bin/env python
"""Generate golden vectors for the THIRD-PARTY stages of the path by calling the real packages.

The path's image / network arithmetic lives in packages that are neither vendored under the reference tree nor installed in
the build container: OpenCV, mmcv-full 1.x, mmpose 0.x, mmdet 2.x, mmtrack 0.x, VideoPose3D.  oracle/
restates them and is therefore "parity unpinned" for those stages.  This script is how that changes: on ANY machine where
some of the packages import (a PosePipe deployment, the reference's Docker image) it runs THEIR functions on the seeded
inputs of tests/golden/spec_3p.py and writes tests/golden/3p_<section>.npz (inputs + third-party outputs, a few hundred
KB in total).  Commit those files; tests/test_oracle_3p.py then holds oracle/ to them (and xfails, with the reason, for
every section whose fixture is absent).

  python tests/golden/make_goldens_3p.py [--out tests/golden] [--only cv2_affine,mmpose]
  VIDEOPOSE3D_PATH=/path/to/VideoPose3D  (same variable as pose_pipeline/wrappers/videopose3d.py:40) for the lifting net

Sections are independent: a missing package or a failing call skips that section (or that key) and is reported; nothing
here imports oracle/ except through spec_3p's input builders (box -> centre / scale / triangles are inputs of the cv2
section and are themselves pinned by the mmpose section).  NOT runnable in the build container: none of the packages is
installed there (the script says so and exits 0 with nothing written).

Third-party entry points exercised (the call sites that select them: pose_pipeline/wrappers/mmpose.py:28,57,75;
wrappers/mmtrack.py:30,45; wrappers/videopose3d.py:43-50,66-82; 3rdparty/mmpose/config/.../hrnet_w48_coco_384x288_dark.py;
3rdparty/mmtracking/_base_/models/faster_rcnn_r50_fpn.py, mot/deepsort/sort_faster-rcnn_fpn_4e_mot17-private-half.py):
  cv2.getAffineTransform, cv2.warpAffine(INTER_LINEAR), cv2.GaussianBlur, cv2.getGaussianKernel, cv2.resize(INTER_LINEAR),
  cv2.cvtColor(COLOR_YUV2BGR_NV12)
  mmpose: apis.inference._box2cs, core.post_processing.{get_affine_transform, flip_back, transform_preds},
          core.evaluation.top_down_eval.keypoints_from_heatmaps, models HRNet + TopdownHeatmapSimpleHead
  mmcv:   ops.nms, ops.roi_align, imnormalize
  mmtrack: models.trackers.SortTracker (reid=None) + models.motion.KalmanFilter
  VideoPose3D: common.model.TemporalModelOptimized1f, common.generators.UnchunkedGenerator-style edge padding
"""
import argparse
import importlib
import os
import sys
import traceback
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

import spec_3p as S  # noqa: E402


def have(mod):
    try:
        importlib.import_module(mod)
        return True
    except Exception:
        return False


# ---- sections: third-party side -------------------------------------------------------------------------------------------
def ref_cv2_affine(d):
    import cv2
    out = {}
    for k in range(int(d["n_cases"])):
        trans = cv2.getAffineTransform(np.float32(d[f"c{k}_src"]), np.float32(d[f"c{k}_dst"]))
        size = tuple(int(v) for v in d[f"c{k}_size"])
        out[f"c{k}_trans"] = np.asarray(trans, np.float64)
        out[f"c{k}_crop"] = cv2.warpAffine(d["frames"][int(d[f"c{k}_frame"])], trans, size, flags=cv2.INTER_LINEAR)
    return out


def ref_cv2_blur(d):
    import cv2
    out = {}
    for k in (17, 11):
        border = (k - 1) // 2
        res = []
        for m in d["hm"]:                                   # mmpose _gaussian_blur: zero border, blur, crop
            dr = np.zeros((m.shape[0] + 2 * border, m.shape[1] + 2 * border), np.float32)
            dr[border:-border, border:-border] = m
            dr = cv2.GaussianBlur(dr, (k, k), 0)
            res.append(dr[border:-border, border:-border].copy())
        out[f"blur{k}"] = np.stack(res)
        out[f"kernel{k}"] = cv2.getGaussianKernel(k, 0, cv2.CV_32F).reshape(-1)
    return out


def ref_cv2_resize(d):
    import cv2
    return {"down": cv2.resize(d["img"], tuple(int(v) for v in d["dsize"]), interpolation=cv2.INTER_LINEAR),
            "up": cv2.resize(d["img"], tuple(int(v) for v in d["dsize_up"]), interpolation=cv2.INTER_LINEAR)}


def ref_cv2_nv12(d):
    import cv2
    return {"bgr": np.stack([cv2.cvtColor(np.ascontiguousarray(f), cv2.COLOR_YUV2BGR_NV12) for f in d["planes"]])}


def _coco_flip_pairs():
    return [[1, 2], [3, 4], [5, 6], [7, 8], [9, 10], [11, 12], [13, 14], [15, 16]]


def ref_mmpose(d):
    out = {}
    from mmpose.core.post_processing import flip_back, get_affine_transform, transform_preds
    from mmpose.core.evaluation.top_down_eval import keypoints_from_heatmaps
    try:
        from mmpose.apis.inference import _box2cs
        cfg = types.SimpleNamespace(data_cfg={"image_size": [288, 384]})
        for i, bb in enumerate(d["boxes"]):
            c, s = _box2cs(cfg, list(bb))
            out[f"box2cs{i}"] = np.concatenate([np.asarray(c, np.float32), np.asarray(s, np.float32)])
    except Exception:                                        # newer 0.x: TopDownGetBboxCenterScale(padding=1.25)
        from mmpose.datasets.pipelines import TopDownGetBboxCenterScale
        t = TopDownGetBboxCenterScale(padding=1.25)
        for i, bb in enumerate(d["boxes"]):
            r = t({"bbox": np.asarray(bb, np.float32), "ann_info": {"image_size": np.array([288, 384])}})
            out[f"box2cs{i}"] = np.concatenate([np.asarray(r["center"], np.float32), np.asarray(r["scale"], np.float32)])
    for i in range(len(d["boxes"])):
        cs = out[f"box2cs{i}"]
        out[f"affine{i}"] = np.asarray(get_affine_transform(cs[:2], cs[2:], 0, [288, 384]), np.float64)
    back = flip_back(d["hm_flipped"].copy(), _coco_flip_pairs(), target_type="GaussianHeatmap")
    back[:, :, :, 1:] = back[:, :, :, :-1]                   # test_cfg shift_heatmap=True (TopdownHeatmapSimpleHead.inference_model)
    out["merged"] = ((d["hm"] + back) * 0.5).astype(np.float32)
    for post, kernel in (("unbiased", 17), ("default", 11)):
        p, m = keypoints_from_heatmaps(d["hm"].copy(), d["center"], d["scale"], post_process=post, kernel=kernel)
        out[f"preds_{post}"], out[f"maxvals_{post}"] = np.asarray(p, np.float32), np.asarray(m, np.float32)
    out["transform_preds"] = np.asarray(transform_preds(np.array([[0.0, 0.0], [47.0, 63.0], [10.25, 20.75]], np.float32),
                                                        d["center"][0], d["scale"][0], [48, 64]), np.float32)
    return out


def ref_mmcv(d):
    import mmcv
    import torch
    from mmcv.ops import nms, roi_align
    out = {}
    for t in (0.5, 0.7):
        _, inds = nms(torch.from_numpy(d["boxes"]), torch.from_numpy(d["scores"]), t)
        out[f"keep_{int(t * 10)}"] = inds.cpu().numpy().astype(np.int64)
    y = roi_align(torch.from_numpy(d["feat"]), torch.from_numpy(d["rois"]), (7, 7), 1.0 / 8.0, 0, "avg", True)
    out["roi_align_s8"] = y.cpu().numpy().astype(np.float32)
    out["imnormalize"] = mmcv.imnormalize(d["img"], np.array([123.675, 116.28, 103.53]), np.array([58.395, 57.12, 57.375]), to_rgb=True)
    return out


def ref_mmtrack(d):
    import torch
    from mmtrack.models.motion import KalmanFilter
    from mmtrack.models.trackers import SortTracker
    trk = SortTracker(obj_score_thr=0.5, reid=None, match_iou_thr=0.5, num_tentatives=2, num_frames_retain=100)
    model = types.SimpleNamespace(motion=KalmanFilter(center_only=False))
    out = {}
    for t in range(len(d["counts"])):
        rows = torch.from_numpy(d["dets"][t, : int(d["counts"][t])].copy())
        labels = torch.zeros(len(rows), dtype=torch.long)
        bboxes, labels, ids = trk.track(img=None, img_metas=[{}], model=model, feats=None, bboxes=rows, labels=labels, frame_id=t)
        out[f"rows{t}"] = np.concatenate([ids.cpu().numpy()[:, None].astype(np.float32), bboxes.cpu().numpy().astype(np.float32)], axis=1)
    return out


def ref_nets(d):
    import torch
    out = {}
    spec, hsd, vsd = S.nets_state_dicts()
    try:
        from mmpose.models import build_posenet
        ch = list(spec.channels)
        cfg = dict(type="TopDown", pretrained=None,
                   backbone=dict(type="HRNet", in_channels=3, extra=dict(
                       stage1=dict(num_modules=1, num_branches=1, block="BOTTLENECK", num_blocks=(4,), num_channels=(64,)),
                       stage2=dict(num_modules=1, num_branches=2, block="BASIC", num_blocks=(4, 4), num_channels=tuple(ch[:2])),
                       stage3=dict(num_modules=4, num_branches=3, block="BASIC", num_blocks=(4, 4, 4), num_channels=tuple(ch[:3])),
                       stage4=dict(num_modules=3, num_branches=4, block="BASIC", num_blocks=(4, 4, 4, 4), num_channels=tuple(ch)))),
                   keypoint_head=dict(type="TopdownHeatmapSimpleHead", in_channels=ch[0], out_channels=17, num_deconv_layers=0,
                                      extra=dict(final_conv_kernel=1), loss_keypoint=dict(type="JointsMSELoss", use_target_weight=True)),
                   train_cfg=dict(), test_cfg=dict(flip_test=False, post_process="unbiased", shift_heatmap=True, modulate_kernel=17))
        model = build_posenet(cfg).eval()
        missing = model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in hsd.items()}, strict=False)
        assert not [k for k in missing.missing_keys if "num_batches_tracked" not in k], missing.missing_keys
        with torch.no_grad():
            out["hrnet_heatmaps"] = model.forward_dummy(torch.from_numpy(d["x_hrnet"])).cpu().numpy().astype(np.float32)
    except Exception:
        traceback.print_exc()
    try:
        sys.path.append(os.environ["VIDEOPOSE3D_PATH"])
        from common.model import TemporalModelOptimized1f
        model = TemporalModelOptimized1f(17, 2, 17, filter_widths=[3, 3, 3, 3, 3], causal=False, dropout=0.25, channels=1024)
        model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in vsd.items()}, strict=False)
        model.eval()
        kp = d["kp_vp3d"]
        pad = 121
        win = np.stack([np.pad(kp, ((pad, pad), (0, 0), (0, 0)), "edge")[i:i + 2 * pad + 1] for i in range(len(kp))])   # ChunkedGenerator windows
        with torch.no_grad():
            out["vp3d"] = model(torch.from_numpy(win.astype("float32"))).cpu().numpy()[:, 0].astype(np.float32)
    except Exception:
        traceback.print_exc()
    return out


REFERENCE = {"cv2_affine": ref_cv2_affine, "cv2_blur": ref_cv2_blur, "cv2_resize": ref_cv2_resize, "cv2_nv12": ref_cv2_nv12, "mmpose": ref_mmpose,
             "mmcv": ref_mmcv, "mmtrack": ref_mmtrack, "nets": ref_nets}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=HERE)
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    written = []
    for name, (inputs, _oracle, checks, needs) in S.SECTIONS.items():
        if args.only and name not in args.only.split(","):
            continue
        absent = [m for m in needs if not have(m)]
        if absent:
            print(f"[{name}] skipped: {', '.join(absent)} not importable here")
            continue
        d = inputs(S.section_rng(name))
        try:
            ref = REFERENCE[name](d)
        except Exception:
            traceback.print_exc()
            print(f"[{name}] FAILED (see traceback) -- nothing written")
            continue
        missing = [k for k in checks(d) if k not in ref]
        versions = {m: getattr(importlib.import_module(m), "__version__", "?") for m in needs}
        path = os.path.join(args.out, f"3p_{name}.npz")
        np.savez_compressed(path, **d, **ref, versions=np.array(repr(versions)))
        written.append(path)
        print(f"[{name}] wrote {path} ({os.path.getsize(path) // 1024} KB), packages {versions}" +
              (f"; keys not produced: {missing}" if missing else ""))
    if not written:
        print("no fixture written: none of the third-party packages of the path is installed on this machine")


if __name__ == "__main__":
    main()
