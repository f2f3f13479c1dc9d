"""DeepSortYOLOv4 path on the CPU: oracle vs the golden vectors generated from the reference (real PIL letterbox,
extract_image_patch), and the product's host-side tables / program builders vs the oracle."""
import os

import numpy as np

from oracle import reid as oreid
from oracle import yolo as oyolo
from posepipeline_amd.models import mars, yolov4

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_letterbox_matches_pil_golden():
    g = np.load(os.path.join(GOLD, "letterbox.npz"))
    for i in range(int(g["n"])):
        out, _ = oyolo.letterbox(g[f"in{i}"], tuple(int(v) for v in g[f"size{i}"]))
        assert np.array_equal(out, g[f"out{i}"]), i          # Pillow's 8-bit bicubic, bit for bit


def test_patch_rect_matches_reference_golden():
    g = np.load(os.path.join(GOLD, "reid_patch.npz"))
    hw = tuple(int(v) for v in g["image_hw"])
    for b, is_int, rect in zip(g["boxes"], g["is_int"], g["rects"]):
        bb = b.astype(np.int64) if is_int else b
        for fn in (oreid.patch_rect, mars.patch_rect):         # oracle and product host code
            got = fn(bb, hw)
            assert ([-1] * 4 if got is None else list(got)) == list(rect)


def test_product_bicubic_tables_equal_oracle():
    for src, dst in ((1920, 416), (1080, 234), (240, 96), (40, 96), (77, 77)):
        tab, ksize = yolov4.pil_bicubic_table(src, dst)
        ref = oyolo.pil_coeffs(src, dst)
        assert tab.shape == (dst, 2 + ksize)
        for xx, (xmin, cnt, k) in enumerate(ref):
            assert tab[xx, 0] == xmin and tab[xx, 1] == cnt and cnt <= ksize
            assert np.array_equal(tab[xx, 2:2 + cnt], k) and not tab[xx, 2 + cnt:].any()
    assert yolov4.letterbox_geometry(1080, 1920) == (416, 234)


def test_yolov4_program_shape():
    layers = yolov4._layers()
    assert len(layers) == 110                                  # 72 backbone + 38 head convolutions
    shapes = yolov4.yolov4_param_shapes()
    n_params = sum(int(np.prod(s)) for s in shapes.values())
    assert 64.0e6 < n_params < 64.8e6                          # YOLOv4: 64.4 M parameters (incl. BN statistics)
    sd = yolov4.synth_params(yolov4.yolov4_param_shapes(num_classes=2), seed=0)
    prog = yolov4.build_yolov4_program(sd, size=64, num_classes=2)
    assert {"input", "y19", "y38", "y76"} <= set(prog.named)
    assert [prog.bufs[prog.named[k]] for k in ("y19", "y38", "y76")] == [(2, 2, 21), (4, 4, 21), (8, 8, 21)]
    full = yolov4.build_yolov4_program(yolov4.synth_params(shapes, seed=0), size=416)
    assert 29.0e9 < full.flops / 2 < 31.0e9                    # ~30 GMAC at 416x416 (60 BFLOPs)


def test_mars_program_shape():
    sd = yolov4.synth_params(mars.mars_param_shapes(), seed=1)
    prog = mars.build_mars_program(sd)
    assert prog.bufs[prog.named["input"]] == (128, 64, 4) and prog.bufs[prog.named["features"]] == (1, 1, 128)
    assert mars.tf_same(63, 3, 2) == (1, 0, 32) and mars.tf_same(32, 3, 2) == (0, 1, 16) and mars.tf_same(32, 1, 2) == (0, 0, 16)
    n_params = sum(int(np.prod(s)) for s in mars.mars_param_shapes().values())
    assert 2.7e6 < n_params < 2.9e6                            # the deep_sort paper's 2.8 M parameter network


def test_oracle_yolo_decode_and_nms_known_answers():
    # one confident cell, everything else suppressed: the decoded box must be the anchor box centred on the cell
    nc, g = 2, 2
    out = np.full((1, g, g, 3 * (5 + nc)), -20.0, np.float32)
    f = out.reshape(g, g, 3, 5 + nc)
    f[1, 0, 1, :] = [0.0, 0.0, 0.0, 0.0, 20.0, 20.0, -20.0]    # cell (y=1, x=0), anchor 1, class 0
    anchors = np.array([[10, 12], [16, 24], [30, 20]], np.float32)
    boxes, scores = oyolo.boxes_and_scores(out, anchors, nc, (64, 64), (64, 64))
    i = (1 * g + 0) * 3 + 1
    assert scores[i, 0] > 0.99 and (np.delete(scores[:, 0], i) < 1e-6).all()
    assert np.allclose(boxes[i], [48 - 12, 16 - 8, 48 + 12, 16 + 8], atol=1e-4)      # y1, x1, y2, x2
    b = np.array([[0, 0, 10, 10], [0, 1, 10, 11], [20, 20, 30, 30], [0, 0, 10, 10]], np.float32)
    s = np.array([0.9, 0.8, 0.7, 0.9], np.float32)
    assert list(oyolo.tf_nms(b, s, 200, 0.5)) == [0, 2]       # tie 0/3 -> lower index; 1 overlaps 0 (IoU .82)
    assert list(oyolo.tf_nms(b, s, 1, 0.5)) == [0]
