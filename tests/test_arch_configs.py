"""Architecture and hyper-parameters pinned on the config files the reference VENDORS (CPU).

tests/golden/arch_configs.json holds the values evaluated from
  3rdparty/mmpose/config/top_down/darkpose/coco/hrnet_w48_coco_384x288_dark.py, _base_/coco.py, _base_/halpe.py,
  3rdparty/mmtracking/mot/deepsort/deepsort_faster-rcnn_fpn_4e_mot17-private-half.py (+ its _base_ files),
  3rdparty/mmtracking/mot/bytetrack/bytetrack_yolox_x_crowdhuman_mot17-private.py (+ its _base_ files)
by tests/golden/make_goldens_cfg.py in the build container.  The product's tables are typed in by hand with citations; here every
one of them -- and the oracle's constants -- is compared with the evaluated files, so a typo on either side cannot hide behind
"device == oracle".  Where the product holds the structure implicitly (the HRNet builder, the ResNet / FPN / RoI-head builders) it is
read off the PARAMETER SHAPES the builders ask a checkpoint for, i.e. off the architecture that actually runs."""
import ctypes as C
import inspect
import json
import os
import re

import numpy as np
import pytest

from posepipeline_amd import _lib, ops, tracking
from posepipeline_amd.models import faster_rcnn as fr
from posepipeline_amd.models import hrnet, reid_r50, yolox
from posepipeline_amd.wrappers import mmpose as wmm

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def cfg():
    with open(os.path.join(HERE, "golden", "arch_configs.json")) as f:
        return json.load(f)


def defaults(fn):
    return {k: p.default for k, p in inspect.signature(fn).parameters.items() if p.default is not inspect.Parameter.empty}


def hrnet_extra_from_shapes(shapes):
    """mmpose's `extra` dict, re-derived from the state-dict keys / shapes the HRNet builder consumes"""
    extra = {}
    blocks = sorted({int(m.group(1)) for k in shapes for m in [re.match(r"backbone\.layer1\.(\d+)\.conv1\.weight", k)] if m})
    extra["stage1"] = dict(num_modules=1, num_branches=1,
                           block="BOTTLENECK" if "backbone.layer1.0.conv3.weight" in shapes else "BASIC",
                           num_blocks=[len(blocks)], num_channels=[shapes["backbone.layer1.0.conv1.weight"][0]])
    for s in (2, 3, 4):
        keys = [re.match(rf"backbone\.stage{s}\.(\d+)\.branches\.(\d+)\.(\d+)\.conv1\.weight", k) for k in shapes]
        keys = [m for m in keys if m]
        n_mod = 1 + max(int(m.group(1)) for m in keys)
        n_br = 1 + max(int(m.group(2)) for m in keys)
        n_blk = [1 + max(int(m.group(3)) for m in keys if int(m.group(1)) == 0 and int(m.group(2)) == b) for b in range(n_br)]
        # every module of the stage has the same branches / blocks
        for mod in range(n_mod):
            for b in range(n_br):
                assert 1 + max(int(m.group(3)) for m in keys if int(m.group(1)) == mod and int(m.group(2)) == b) == n_blk[b]
        ch = [shapes[f"backbone.stage{s}.0.branches.{b}.0.conv1.weight"][0] for b in range(n_br)]
        basic = f"backbone.stage{s}.0.branches.0.0.conv3.weight" not in shapes
        extra[f"stage{s}"] = dict(num_modules=n_mod, num_branches=n_br, block="BASIC" if basic else "BOTTLENECK", num_blocks=n_blk,
                                  num_channels=ch)
    return extra


def test_hrnet_w48_architecture_and_test_cfg(cfg):
    g = cfg["hrnet_w48_coco_384x288_dark"]
    spec = hrnet.hrnet_w48_384x288()
    shapes = hrnet.hrnet_param_shapes(spec)
    assert g["backbone_type"] == "HRNet" and g["in_channels"] == 3
    assert hrnet_extra_from_shapes(shapes) == g["extra"]
    # stem: two 3x3 stride-2 convolutions to 64 channels (mmpose HRNet; not a config entry) feed stage1's 64-channel bottlenecks
    assert shapes["backbone.conv1.weight"] == (64, 3, 3, 3) and shapes["backbone.conv2.weight"] == (64, 64, 3, 3)
    # head: TopDownSimpleHead without deconvolutions = one final convolution
    fw = shapes["keypoint_head.final_layer.weight"]
    assert g["head"]["type"] == "TopDownSimpleHead" and g["head"]["num_deconv_layers"] == 0
    assert fw == (g["head"]["out_channels"], g["head"]["in_channels"], g["head"]["final_conv_kernel"], g["head"]["final_conv_kernel"])
    assert not any(k.startswith("keypoint_head.deconv") for k in shapes)
    # input / heat-map sizes ([w, h] in the config)
    assert [spec.in_w, spec.in_h] == g["image_size"] and list(spec.heatmap_hw[::-1]) == g["heatmap_size"]
    assert spec.num_joints == g["num_joints"] == cfg["dataset_coco"]["num_keypoints"]
    # test_cfg -> what the wrapper hands the fused stage
    s = wmm.topdown_settings("HRNet_W48_COCO")
    t = g["test_cfg"]
    assert (s["flip_perm"] is not None) == t["flip_test"]
    assert s["post"] == t["post_process"] and s["shift_heatmap"] == t["shift_heatmap"] and s["blur_kernel"] == t["modulate_kernel"]
    assert s["num_joints"] == g["num_joints"]
    # val pipeline: TopDownAffine -> ToTensor -> NormalizeTensor(mean, std)
    assert g["val_pipeline_types"] == ["LoadImageFromFile", "TopDownAffine", "ToTensor", "NormalizeTensor", "Collect"]
    assert list(ops.TOPDOWN_MEAN) == g["normalize_mean"] and list(ops.TOPDOWN_STD) == g["normalize_std"]
    from oracle import preprocess as opre
    assert np.array_equal(opre.MEAN, np.array(g["normalize_mean"], np.float32)) and np.array_equal(opre.STD, np.array(g["normalize_std"], np.float32))
    # the Halpe / WholeBody methods share the backbone and the test_cfg (wrappers/mmpose.py:41-52); W32 is mmpose's plain config
    for m in ("HRNet_W48_HALPE", "HRNet_W48_COCOWholeBody"):
        sm = wmm.topdown_settings(m)
        assert (sm["post"], sm["shift_heatmap"], sm["blur_kernel"]) == (s["post"], s["shift_heatmap"], s["blur_kernel"])
        assert hrnet_extra_from_shapes(hrnet.hrnet_param_shapes(wmm._METHODS[m][0](wmm._METHODS[m][2]))) == g["extra"]


def test_flip_pairs_and_joint_names(cfg):
    pairs = lambda p: sorted([min(a, b), max(a, b)] for a, b in p)
    assert pairs(hrnet.COCO_FLIP_PAIRS) == cfg["dataset_coco"]["flip_pairs"]
    assert pairs(hrnet.HALPE_FLIP_PAIRS) == cfg["dataset_halpe"]["flip_pairs"]
    assert cfg["dataset_halpe"]["num_keypoints"] == 136 == wmm._METHODS["HRNet_W48_HALPE"][2]
    # flip_perm is the involution the pairs define
    for k, p in ((17, hrnet.COCO_FLIP_PAIRS), (136, hrnet.HALPE_FLIP_PAIRS)):
        perm = hrnet.flip_perm(k, p)
        assert np.array_equal(perm[perm], np.arange(k)) and int((perm != np.arange(k)).sum()) == 2 * len(p)
    # the joint-name tables of the wrapper follow the dataset files' order (names: "left_eye" -> "Left Eye")
    nice = lambda n: " ".join(w.capitalize() for w in n.split("_"))
    assert wmm.mmpose_joint_dictionary["MMPose"] == [nice(n) for n in cfg["dataset_coco"]["names"]]
    assert wmm.mmpose_joint_dictionary["MMPoseHalpe"][:17] == [nice(n) for n in cfg["dataset_halpe"]["names"][:17]]
    # (joint 19, the dataset file's 'hip', is "Pelvis" in the reference's own table, pose_pipeline/wrappers/mmpose.py:8-24)
    assert [n.lower() for n in wmm.mmpose_joint_dictionary["MMPoseHalpe"][17:19]] == cfg["dataset_halpe"]["names"][17:19]
    assert cfg["dataset_halpe"]["names"][19] == "hip" and wmm.mmpose_joint_dictionary["MMPoseHalpe"][19] == "Pelvis"


def detector_constants():
    lib = _lib.load_library()
    out = (C.c_double * 15)()
    assert lib.pp_detector_constants(out, 15) == 15
    keys = ("rpn_nms_pre", "rpn_nms_iou", "rpn_max_per_img", "rcnn_score_thr", "rcnn_nms_iou", "rcnn_max_per_img", "scale_long",
            "scale_short", "size_divisor", "roi_size", "finest_scale")
    d = dict(zip(keys, list(out)[:11]))
    d["rcnn_stds"] = list(out)[11:15]
    return d


def test_faster_rcnn_architecture_and_test_cfg(cfg):
    g = cfg["deepsort_faster_rcnn"]
    sh = fr.faster_rcnn_param_shapes()
    P = "detector."
    # backbone: ResNet-50 (3, 4, 6, 3 bottlenecks), 'pytorch' style = the stride sits on the 3x3 convolution
    assert g["backbone"] == dict(type="ResNet", depth=50, num_stages=4, out_indices=[0, 1, 2, 3], style="pytorch")
    assert 2 + 3 * sum(b for b, _, _ in fr.R50_LAYERS) == g["backbone"]["depth"] and len(fr.R50_LAYERS) == g["backbone"]["num_stages"]
    for li, (blocks, planes, _) in enumerate(fr.R50_LAYERS):
        assert sh[f"{P}backbone.layer{li + 1}.{blocks - 1}.conv3.weight"] == (4 * planes, planes, 1, 1)
        assert f"{P}backbone.layer{li + 1}.{blocks}.conv1.weight" not in sh
    # neck
    lat = [sh[f"{P}neck.lateral_convs.{i}.conv.weight"] for i in range(4)]
    assert [s[1] for s in lat] == g["neck"]["in_channels"] and {s[0] for s in lat} == {g["neck"]["out_channels"]}
    assert len(fr.STRIDES) == g["neck"]["num_outs"]
    # RPN head
    r = g["rpn_head"]
    d = defaults(fr.base_anchors)
    assert list(d["scales"]) == r["anchor_scales"] and list(d["ratios"]) == r["anchor_ratios"] and list(d["strides"]) == r["anchor_strides"]
    na = len(r["anchor_scales"]) * len(r["anchor_ratios"])
    assert sh[f"{P}rpn_head.rpn_conv.weight"] == (r["feat_channels"], r["in_channels"], 3, 3)
    assert sh[f"{P}rpn_head.rpn_cls.weight"][0] == na and sh[f"{P}rpn_head.rpn_reg.weight"][0] == 4 * na and r["use_sigmoid"]
    assert r["target_means"] == [0.0] * 4 and r["target_stds"] == [1.0] * 4 and r["clip_border"] is False
    # RoI extractor + bbox head
    k = detector_constants()
    e, b = g["roi_extractor"], g["bbox_head"]
    assert e["roi_layer"] == "RoIAlign" and e["sampling_ratio"] == 0 and e["output_size"] == k["roi_size"] == b["roi_feat_size"]
    assert e["featmap_strides"] == list(fr.STRIDES[:4]) and e["out_channels"] == b["in_channels"] == 256
    assert sh[f"{P}roi_head.bbox_head.shared_fcs.0.weight"] == (b["fc_out_channels"], b["in_channels"] * b["roi_feat_size"] ** 2)
    assert sh[f"{P}roi_head.bbox_head.shared_fcs.1.weight"] == (b["fc_out_channels"], b["fc_out_channels"])
    assert sh[f"{P}roi_head.bbox_head.fc_cls.weight"][0] == b["num_classes"] + 1 and not b["use_sigmoid"]
    assert sh[f"{P}roi_head.bbox_head.fc_reg.weight"][0] == 4 * b["num_classes"] and b["reg_class_agnostic"] is False
    assert np.allclose(k["rcnn_stds"], b["target_stds"], rtol=1e-7) and b["target_means"] == [0.0] * 4 and b["clip_border"] is False
    assert np.array_equal(np.array(k["rcnn_stds"], np.float32), np.array(b["target_stds"], np.float32))
    # test_cfg, device side (C constants) and oracle side (function defaults)
    t = g["test_cfg"]
    f32 = lambda v: float(np.float32(v))
    assert (k["rpn_nms_pre"], k["rpn_max_per_img"], k["rpn_nms_iou"]) == (t["rpn"]["nms_pre"], t["rpn"]["max_per_img"], f32(t["rpn"]["nms"]["iou_threshold"]))
    assert t["rpn"]["min_bbox_size"] == 0 and t["rpn"]["nms"]["type"] == "nms" == t["rcnn"]["nms"]["type"]
    assert (k["rcnn_score_thr"], k["rcnn_nms_iou"], k["rcnn_max_per_img"]) == (f32(t["rcnn"]["score_thr"]), f32(t["rcnn"]["nms"]["iou_threshold"]), t["rcnn"]["max_per_img"])
    from oracle import detector as odet
    dr, df = defaults(odet.rpn_proposals), defaults(odet.final_detections)
    assert (dr["nms_pre"], dr["max_per_img"], dr["iou_thr"]) == (t["rpn"]["nms_pre"], t["rpn"]["max_per_img"], t["rpn"]["nms"]["iou_threshold"])
    assert (df["score_thr"], df["iou_thr"], df["max_per_img"]) == (t["rcnn"]["score_thr"], t["rcnn"]["nms"]["iou_threshold"], t["rcnn"]["max_per_img"])
    assert fr.Detector.MAX_ROIS == t["rpn"]["max_per_img"] and fr.Detector.MAX_DET == t["rcnn"]["max_per_img"]
    # test pipeline: Resize(keep_ratio) to (1088, 1088) -> Normalize(mean, std, to_rgb) -> Pad(32)
    assert g["test_img_scale"] == [k["scale_long"], k["scale_short"]] and g["test_flip"] is False
    tt = {t_["type"]: t_ for t_ in g["test_transforms"]}
    assert tt["Resize"]["keep_ratio"] is True and tt["Pad"]["size_divisor"] == k["size_divisor"]
    n = g["img_norm_cfg"]
    assert list(fr.DET_MEAN) == n["mean"] and list(fr.DET_STD) == n["std"] and n["to_rgb"] is True
    lib = _lib.load_library()
    dims = [C.c_int32() for _ in range(4)]
    assert lib.pp_detector_input_size(1080, 1920, *[C.byref(v) for v in dims]) == 0
    assert [v.value for v in dims] == [612, 1088, 640, 1088]          # mmcv.rescale_size + Pad(32) of a 1080p frame


def test_sort_tracker_and_reid(cfg):
    g = cfg["deepsort_faster_rcnn"]
    t = g["tracker"]
    assert g["model_type"] == "DeepSORT" and t["type"] == "SortTracker" and g["motion"] == dict(type="KalmanFilter", center_only=False)
    want = dict(obj_score_thr=t["obj_score_thr"], match_iou_thr=t["match_iou_thr"], match_score_thr=t["reid"]["match_score_thr"],
                num_samples=t["reid"]["num_samples"], num_tentatives=t["num_tentatives"], num_frames_retain=t["num_frames_retain"])
    assert defaults(tracking.SortReidTracker.__init__) == want
    from oracle import reid_mm
    from oracle import tracking as otrk
    assert defaults(reid_mm.SortReidTrackerRef.__init__ if hasattr(reid_mm, "SortReidTrackerRef") else
                    next(v for v in vars(reid_mm).values() if inspect.isclass(v) and "match_score_thr" in defaults(v.__init__)).__init__) == want
    assert defaults(next(v for v in vars(otrk).values() if inspect.isclass(v) and "obj_score_thr" in defaults(v.__init__)).__init__) == \
        dict(obj_score_thr=t["obj_score_thr"], match_iou_thr=t["match_iou_thr"])
    d = defaults(tracking.Tracker.__init__)
    assert d["match_iou_thr"] == t["match_iou_thr"] and d["obj_score_thr"] == t["obj_score_thr"]
    assert t["momentums"] is None and t["reid"]["img_norm_cfg"] is None
    # ReID model: ResNet-50 (last stage only) -> GlobalAveragePooling((8, 4), 1) -> Linear 2048 -> 1024 (+ BN + ReLU) -> Linear -> 128
    r = g["reid"]
    assert r["backbone"] == dict(type="ResNet", depth=50, num_stages=4, out_indices=[3], style="pytorch")
    assert list(reid_r50.CROP_HW) == t["reid"]["img_scale"] == list(defaults(reid_mm.crop_imgs)["out_hw"])
    sh = reid_r50.reid_param_shapes() if hasattr(reid_r50, "reid_param_shapes") else reid_r50.param_shapes()
    h = r["head"]
    assert h["num_fcs"] == 1 and sh["head.fcs.0.fc.weight"] == (h["fc_channels"], h["in_channels"])
    assert sh["head.fc_out.weight"] == (h["out_channels"], h["fc_channels"])
    from posepipeline_amd.models import synth
    prog = reid_r50.build_reid_program(synth.synth_state_dict(sh, seed=3))
    pool = [op for op in prog.ops if op.type == _lib.PP_OP_AVGPOOL]
    assert len(pool) == 1 and [pool[0].kh, pool[0].kw, pool[0].stride] == r["neck"]["kernel_size"] + [r["neck"]["stride"]]


def test_bytetrack_yolox(cfg):
    g = cfg["bytetrack_yolox_x"]
    assert g["model_type"] == "ByteTrack" and g["motion"] == dict(type="KalmanFilter")
    assert (yolox.DEEPEN, yolox.WIDEN) == (g["backbone"]["deepen_factor"], g["backbone"]["widen_factor"])
    sh = yolox.yolox_param_shapes()
    P = "detector."
    n = g["neck"]
    assert [sh[f"{P}neck.out_convs.{i}.conv.weight"][1] for i in range(3)] == n["in_channels"]
    assert {sh[f"{P}neck.out_convs.{i}.conv.weight"][0] for i in range(3)} == {n["out_channels"]}
    # CSP blocks of the neck: round(3 * deepen_factor) bottlenecks each
    nb = 1 + max(int(m.group(1)) for k in sh for m in [re.match(rf"{re.escape(P)}neck\.top_down_blocks\.0\.blocks\.(\d+)\.conv1\.conv\.weight", k)] if m)
    assert nb == n["num_csp_blocks"] == round(3 * g["backbone"]["deepen_factor"])
    b = g["bbox_head"]
    assert sh[f"{P}bbox_head.multi_level_conv_cls.0.weight"] == (b["num_classes"], b["feat_channels"], 1, 1)
    assert sh[f"{P}bbox_head.multi_level_cls_convs.0.0.conv.weight"][:2] == (b["feat_channels"], b["in_channels"])
    d = defaults(yolox.YoloXDetector.__init__)
    assert list(d["scale"]) == g["test_img_scale"] == g["input_size"] and g["test_flip"] is False
    assert (d["score_thr"], d["iou_thr"]) == (g["test_cfg"]["score_thr"], g["test_cfg"]["nms"]["iou_threshold"])
    tt = {t_["type"]: t_ for t_ in g["test_transforms"]}
    assert tt["Resize"]["keep_ratio"] is True and tt["Pad"]["size_divisor"] == yolox.SIZE_DIVISOR
    assert tt["Pad"]["pad_val"]["img"] == [yolox.PAD_VALUE] * 3
    assert tt["Normalize"]["mean"] == [0.0] * 3 and tt["Normalize"]["std"] == [1.0] * 3 and tt["Normalize"]["to_rgb"] is False
    from oracle import yolox as oyx
    do, dd = defaults(oyx.preprocess), defaults(oyx.detections)
    assert (list(do["scale"]), do["divisor"], do["pad_val"]) == (g["test_img_scale"], tt["Pad"]["size_divisor"], tt["Pad"]["pad_val"]["img"][0])
    assert (dd["score_thr"], dd["iou_thr"]) == (g["test_cfg"]["score_thr"], g["test_cfg"]["nms"]["iou_threshold"])
    # tracker
    t = g["tracker"]
    want = dict(high=t["obj_score_thrs"]["high"], low=t["obj_score_thrs"]["low"], init_track_thr=t["init_track_thr"],
                weight_iou_with_det_scores=t["weight_iou_with_det_scores"], match_iou_high=t["match_iou_thrs"]["high"],
                match_iou_low=t["match_iou_thrs"]["low"], match_iou_tentative=t["match_iou_thrs"]["tentative"],
                num_frames_retain=t["num_frames_retain"])
    d = defaults(tracking.ByteTracker.__init__)
    assert {k: d[k] for k in want} == want
    from oracle import bytetrack as obt
    do = defaults(next(v for v in vars(obt).values() if inspect.isclass(v) and "thr_tentative" in defaults(v.__init__)).__init__)
    assert (do["high"], do["low"], do["init_thr"], do["weight_iou"], do["thr_high"], do["thr_low"], do["thr_tentative"]) == \
        (want["high"], want["low"], want["init_track_thr"], want["weight_iou_with_det_scores"], want["match_iou_high"], want["match_iou_low"],
         want["match_iou_tentative"])
