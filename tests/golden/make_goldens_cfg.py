#!/usr/bin/env python
"""Architecture / hyper-parameter pins from the config files the reference VENDORS (3rdparty/mmpose/config, 3rdparty/mmtracking):
the plain `dict(...)` Python files its wrappers hand to mmpose / mmtrack (wrappers/mmpose.py:33-52, wrappers/mmtrack.py:12-27).

Runs in the build container only (reads /root/reference); writes tests/golden/arch_configs.json -- DATA only (numbers, names and
index pairs picked out of the evaluated configs), no source text.  tests/test_arch_configs.py compares the product's tables
(models/hrnet.py, wrappers/mmpose.py:_METHODS, models/faster_rcnn.py, the detector's C constants, tracking.py, models/reid_r50.py,
models/yolox.py) and the oracle's constants with it on the CPU.

The loader is a minimal restatement of mmcv.Config's file semantics: a config is the module namespace of the file; `_base_`
names files (relative to the file) whose dicts are merged first, the file's own values override them key by key (dicts recurse).

usage: python tests/golden/make_goldens_cfg.py [/root/reference]"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def _merge(base, over):
    out = dict(base)
    for k, v in over.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict) and not v.get("_delete_", False):
            out[k] = _merge(out[k], v)
        else:
            out[k] = {kk: vv for kk, vv in v.items() if kk != "_delete_"} if isinstance(v, dict) else v
    return out


def load_config(path):
    ns = {"__file__": path}
    with open(path) as f:
        exec(compile(f.read(), path, "exec"), ns)
    cfg = {k: v for k, v in ns.items() if not k.startswith("__") and isinstance(v, (dict, list, tuple, int, float, str, bool, type(None)))}
    bases = cfg.pop("_base_", [])
    if isinstance(bases, str):
        bases = [bases]
    merged = {}
    for b in bases:
        merged = _merge(merged, load_config(os.path.normpath(os.path.join(os.path.dirname(path), b))))
    return _merge(merged, cfg)


def flip_pairs(keypoint_info):
    """index pairs (i < j) of the `swap=` fields of an mmpose dataset_info"""
    by_name = {v["name"]: k for k, v in keypoint_info.items()}
    pairs = set()
    for k, v in keypoint_info.items():
        if v["swap"]:
            j = by_name[v["swap"]]
            pairs.add((min(k, j), max(k, j)))
    return sorted(pairs)


def step(pipeline, type_):
    return next(s for s in pipeline if s["type"] == type_)


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    pose = os.path.join(ref, "3rdparty/mmpose/config")
    trk = os.path.join(ref, "3rdparty/mmtracking")
    out = {}

    # ---- wrappers/mmpose.py:33-35 "HRNet_W48_COCO" -----------------------------------------------------------------------------
    c = load_config(os.path.join(pose, "top_down/darkpose/coco/hrnet_w48_coco_384x288_dark.py"))
    m = c["model"]
    norm = step(c["val_pipeline"], "NormalizeTensor")
    out["hrnet_w48_coco_384x288_dark"] = {
        "backbone_type": m["backbone"]["type"],
        "in_channels": m["backbone"]["in_channels"],
        "extra": {k: {kk: (list(vv) if isinstance(vv, tuple) else vv) for kk, vv in s.items()} for k, s in m["backbone"]["extra"].items()},
        "head": {"type": m["keypoint_head"]["type"], "in_channels": m["keypoint_head"]["in_channels"],
                 "out_channels": m["keypoint_head"]["out_channels"], "num_deconv_layers": m["keypoint_head"]["num_deconv_layers"],
                 "final_conv_kernel": m["keypoint_head"]["extra"]["final_conv_kernel"]},
        "test_cfg": m["test_cfg"],
        "image_size": c["data_cfg"]["image_size"], "heatmap_size": c["data_cfg"]["heatmap_size"],
        "num_joints": c["data_cfg"]["num_joints"],
        "val_pipeline_types": [s["type"] for s in c["val_pipeline"]],
        "normalize_mean": norm["mean"], "normalize_std": norm["std"],
    }
    for name in ("coco", "halpe"):
        info = load_config(os.path.join(pose, f"_base_/{name}.py"))["dataset_info"]
        out[f"dataset_{name}"] = {"num_keypoints": len(info["keypoint_info"]),
                                  "names": [info["keypoint_info"][i]["name"] for i in range(len(info["keypoint_info"]))],
                                  "flip_pairs": flip_pairs(info["keypoint_info"])}

    # ---- wrappers/mmtrack.py:16-19 "deepsort" (the Faster-RCNN + SortTracker configuration) ------------------------------------------
    c = load_config(os.path.join(trk, "mot/deepsort/deepsort_faster-rcnn_fpn_4e_mot17-private-half.py"))
    det = c["model"]["detector"]
    msfa = step(c["test_pipeline"], "MultiScaleFlipAug")
    out["deepsort_faster_rcnn"] = {
        "model_type": c["model"]["type"],
        "backbone": {k: (list(v) if isinstance(v, tuple) else v) for k, v in det["backbone"].items() if k in ("type", "depth", "num_stages", "out_indices", "style")},
        "neck": {k: det["neck"][k] for k in ("type", "in_channels", "out_channels", "num_outs")},
        "rpn_head": {"in_channels": det["rpn_head"]["in_channels"], "feat_channels": det["rpn_head"]["feat_channels"],
                     "anchor_scales": det["rpn_head"]["anchor_generator"]["scales"], "anchor_ratios": det["rpn_head"]["anchor_generator"]["ratios"],
                     "anchor_strides": det["rpn_head"]["anchor_generator"]["strides"],
                     "target_means": det["rpn_head"]["bbox_coder"]["target_means"], "target_stds": det["rpn_head"]["bbox_coder"]["target_stds"],
                     "clip_border": det["rpn_head"]["bbox_coder"]["clip_border"], "use_sigmoid": det["rpn_head"]["loss_cls"]["use_sigmoid"]},
        "roi_extractor": {"output_size": det["roi_head"]["bbox_roi_extractor"]["roi_layer"]["output_size"],
                          "sampling_ratio": det["roi_head"]["bbox_roi_extractor"]["roi_layer"]["sampling_ratio"],
                          "roi_layer": det["roi_head"]["bbox_roi_extractor"]["roi_layer"]["type"],
                          "out_channels": det["roi_head"]["bbox_roi_extractor"]["out_channels"],
                          "featmap_strides": det["roi_head"]["bbox_roi_extractor"]["featmap_strides"]},
        "bbox_head": {k: det["roi_head"]["bbox_head"][k] for k in ("type", "in_channels", "fc_out_channels", "roi_feat_size", "num_classes", "reg_class_agnostic")}
        | {"target_means": det["roi_head"]["bbox_head"]["bbox_coder"]["target_means"], "target_stds": det["roi_head"]["bbox_head"]["bbox_coder"]["target_stds"],
           "clip_border": det["roi_head"]["bbox_head"]["bbox_coder"]["clip_border"], "use_sigmoid": det["roi_head"]["bbox_head"]["loss_cls"]["use_sigmoid"]},
        "test_cfg": det["test_cfg"],
        "img_norm_cfg": c["img_norm_cfg"],
        "test_img_scale": list(msfa["img_scale"]), "test_flip": msfa["flip"],
        "test_transforms": [{k: (list(v) if isinstance(v, tuple) else v) for k, v in t.items() if k in ("type", "keep_ratio", "size_divisor")} for t in msfa["transforms"]],
        "motion": c["model"]["motion"],
        "reid": {"backbone": {k: (list(v) if isinstance(v, tuple) else v) for k, v in c["model"]["reid"]["backbone"].items()},
                 "neck": {k: (list(v) if isinstance(v, tuple) else v) for k, v in c["model"]["reid"]["neck"].items()},
                 "head": {k: c["model"]["reid"]["head"][k] for k in ("type", "num_fcs", "in_channels", "fc_channels", "out_channels", "num_classes")}},
        "tracker": {k: ({kk: (list(vv) if isinstance(vv, tuple) else vv) for kk, vv in v.items()} if isinstance(v, dict) else v)
                    for k, v in c["model"]["tracker"].items()},
    }

    # ---- wrappers/mmtrack.py:20-23 "bytetrack" ----------------------------------------------------------------------------------
    c = load_config(os.path.join(trk, "mot/bytetrack/bytetrack_yolox_x_crowdhuman_mot17-private.py"))
    det = c["model"]["detector"]
    msfa = step(c["test_pipeline"], "MultiScaleFlipAug")
    out["bytetrack_yolox_x"] = {
        "model_type": c["model"]["type"],
        "input_size": list(det["input_size"]),
        "backbone": {k: det["backbone"][k] for k in ("type", "deepen_factor", "widen_factor")},
        "neck": {k: det["neck"][k] for k in ("type", "in_channels", "out_channels", "num_csp_blocks")},
        "bbox_head": {k: det["bbox_head"][k] for k in ("type", "num_classes", "in_channels", "feat_channels")},
        "test_cfg": det["test_cfg"],
        "motion": c["model"]["motion"],
        "tracker": c["model"]["tracker"],
        "test_img_scale": list(msfa["img_scale"]), "test_flip": msfa["flip"],
        "test_transforms": [{k: (list(v) if isinstance(v, tuple) else ({kk: list(vv) for kk, vv in v.items()} if isinstance(v, dict) else v))
                             for k, v in t.items() if k in ("type", "keep_ratio", "size_divisor", "mean", "std", "to_rgb", "pad_val")}
                            for t in msfa["transforms"]],
    }
    path = os.path.join(HERE, "arch_configs.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
