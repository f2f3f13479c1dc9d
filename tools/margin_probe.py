"""What do the detector's decision margins look like next to the ACTUAL differences between the default (fp16-form) and the exact
kernels?  Per frame: the eight margins of the split run, and the measured deviations split vs exact of every quantity a decision
is taken on (RPN scores / boxes, RoI scores, final boxes), plus whether the integer outputs agree.  python tools/margin_probe.py [frames]"""
import sys

import numpy as np

sys.path.insert(0, ".")
from posepipeline_amd import _lib as L                      # noqa: E402
from posepipeline_amd.models import faster_rcnn as fr       # noqa: E402
from tests.test_gpu_parity_modes import _clip_1080p_four_persons, _det_sd   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ctx = L.Context(0)
frames = _clip_1080p_four_persons(n, np.random.default_rng(41))
sd = _det_sd()
out = {}
for mode in ("exact", "split"):
    det = fr.Detector(ctx, sd, 1080, 1920, max_frames=n, numerics=mode)
    det.enable_margins(True, 8.0)
    dets, props = det.run(frames, want_proposals=True)
    m = det.margins(n)
    rpn = [det.net_a.read(f"rpn{l}", n) for l in range(5)]
    cls = det.net_b.read("cls", n * det.MAX_ROIS).reshape(n, det.MAX_ROIS, 2)
    reg = det.net_b.read("reg", n * det.MAX_ROIS).reshape(n, det.MAX_ROIS, 4)
    out[mode] = dict(dets=dets, props=props, m=m, rpn=rpn, cls=cls, reg=reg)
    det.close()
e, s = out["exact"], out["split"]
sig = lambda x: 1.0 / (1.0 + np.exp(-x.astype(np.float64)))
print("names:", fr.Detector.MARGIN_NAMES)
for f in range(n):
    d_rpn_score = max(np.abs(sig(e["rpn"][l][f][..., :3]) - sig(s["rpn"][l][f][..., :3])).max() for l in range(5))
    d_rpn_delta = max(np.abs(e["rpn"][l][f][..., 3:15] - s["rpn"][l][f][..., 3:15]).max() for l in range(5))
    same_props = e["props"][f].shape == s["props"][f].shape and np.abs(e["props"][f] - s["props"][f]).max() < 0.5
    d_prop = np.abs(e["props"][f] - s["props"][f]).max() if same_props else np.nan
    rel_prop = (np.abs(e["props"][f] - s["props"][f]).max(1) / np.maximum(1.0, np.maximum(e["props"][f][:, 2] - e["props"][f][:, 0], e["props"][f][:, 3] - e["props"][f][:, 1]))).max() if same_props else np.nan
    same_dets = e["dets"][f].shape == s["dets"][f].shape and np.abs(e["dets"][f][:, :4] - s["dets"][f][:, :4]).max() < 0.5
    if same_props:
        sm = lambda c: np.exp(c[:, 0] - c.max(1)) / np.exp(c - c.max(1, keepdims=True)).sum(1)
        d_roi_score = np.abs(sm(e["cls"][f].astype(np.float64)) - sm(s["cls"][f].astype(np.float64))).max()
        d_reg = np.abs(e["reg"][f] - s["reg"][f]).max()
    else:
        d_roi_score = d_reg = np.nan
    if same_dets:
        bw = np.maximum(e["dets"][f][:, 2] - e["dets"][f][:, 0], e["dets"][f][:, 3] - e["dets"][f][:, 1])
        d_det = np.abs(e["dets"][f][:, :4] - s["dets"][f][:, :4]).max()
        d_det_rel = (np.abs(e["dets"][f][:, :4] - s["dets"][f][:, :4]).max(1) / np.maximum(bw, 1)).max()
        d_det_score = np.abs(e["dets"][f][:, 4] - s["dets"][f][:, 4]).max()
    else:
        d_det = d_det_rel = d_det_score = np.nan
    print(f"frame {f:2d} same_props {int(same_props)} same_dets {int(same_dets)} | margins split: " + " ".join(f"{v:9.2e}" for v in s["m"][f]) +
          f" | margins exact: " + " ".join(f"{v:9.2e}" for v in e["m"][f]))
    print(f"          d_rpn_score {d_rpn_score:.2e} d_rpn_delta {d_rpn_delta:.2e} d_prop_px {d_prop:.2e} rel {rel_prop:.2e} d_roi_score {d_roi_score:.2e} d_reg {d_reg:.2e} "
          f"d_det_px {d_det:.2e} rel {d_det_rel:.2e} d_det_score {d_det_score:.2e}")
