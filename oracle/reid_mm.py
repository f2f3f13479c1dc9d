"""CPU oracle for the ReID branch of mmtrack's DeepSORT configuration.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates, from the published mmtrack 0.x / mmcls / PyTorch sources (none is vendored or installed: PARITY UNPINNED), what
pose_pipeline/wrappers/mmtrack.py:45 reaches with method "deepsort", i.e. the `reid=` / `tracker=` sections of
3rdparty/mmtracking/mot/deepsort/deepsort_faster-rcnn_fpn_4e_mot17-private-half.py:17-54:
  crop_imgs          SortTracker.crop_imgs: box * scale_factor (float32), clamp, int(), empty side + 1, crop of the detector's
                     normalised input tensor, F.interpolate(size (256, 128), 'bilinear', align_corners=False)
  ReidNetRef         mmcls ResNet-50 -> AvgPool2d((8, 4)) -> Linear + BN1d + ReLU -> Linear (fc_out features)
  SortReidTrackerRef SortTracker.track with ReID: Kalman gating (chi2inv95[4]), appearance assignment among confirmed tracks
                     (mean of the last 10 embeddings, torch.cdist, accept <= 2.0), IoU assignment among tracks of the
                     previous frame (accept 1 - IoU < 0.5), tentative / retain bookkeeping.
Where mmtrack feeds NaN (gated pairs) to scipy's Hungarian solver -- rejected by current scipy, unspecified in old ones --
this oracle and the product use the same explicit reading: a gated pair costs 1e6 and is never accepted.
"""
from __future__ import annotations

import numpy as np
from scipy.optimize import linear_sum_assignment

from . import clib
from .bytetrack import KalmanRef, xyxy_to_cxcyah
from .detector import FasterRCNNRef
from .nets import relu
from .tracking import bbox_overlaps

f32 = np.float32
CHI2INV95_4 = 9.4877
GATED = 1e6


def crop_rects(boxes, scale_factor, img_hw):
    out = []
    h, w = img_hw
    for b in np.asarray(boxes, f32).reshape(-1, 4):
        v = [f32(b[k] * f32(scale_factor[k])) for k in range(4)]
        v[0], v[2] = min(max(v[0], f32(0)), f32(w)), min(max(v[2], f32(0)), f32(w))
        v[1], v[3] = min(max(v[1], f32(0)), f32(h)), min(max(v[3], f32(0)), f32(h))
        x1, y1, x2, y2 = (int(t) for t in v)
        if x2 == x1:
            x2 = x1 + 1
        if y2 == y1:
            y2 = y1 + 1
        # entirely beyond the right / bottom edge: mmtrack's slice would be empty (and F.interpolate would raise); take the
        # last pixel column / row -- the product does the same
        x1, x2, y1, y2 = min(x1, w - 1), min(x2, w), min(y1, h - 1), min(y2, h)
        out.append((x1, y1, x2, y2))
    return np.array(out, np.int32).reshape(-1, 4)


def interpolate_bilinear(img_hwc, out_hw):
    """F.interpolate(mode='bilinear', align_corners=False) of one [h][w][c] float32 image (upsample_bilinear2d, float32)"""
    h, w, _ = img_hwc.shape
    oh, ow = out_hw

    def axis(n_in, n_out):
        scale = f32(f32(n_in) / f32(n_out))
        src = np.maximum((scale * (np.arange(n_out, dtype=f32) + f32(0.5))).astype(f32) - f32(0.5), f32(0)).astype(f32)
        i0 = src.astype(np.int64)
        step = (i0 < n_in - 1).astype(np.int64)
        l1 = (src - i0.astype(f32)).astype(f32)
        return i0, step, (f32(1) - l1).astype(f32), l1

    hi, hp, h0, h1 = axis(h, oh)
    wi, wp, w0, w1 = axis(w, ow)
    x = img_hwc.astype(f32)
    top = ((w0[None, :, None] * x[hi][:, wi]).astype(f32) + (w1[None, :, None] * x[hi][:, wi + wp]).astype(f32)).astype(f32)
    bot = ((w0[None, :, None] * x[hi + hp][:, wi]).astype(f32) + (w1[None, :, None] * x[hi + hp][:, wi + wp]).astype(f32)).astype(f32)
    return ((h0[:, None, None] * top).astype(f32) + (h1[:, None, None] * bot).astype(f32)).astype(f32)


def crop_imgs(det_input_hwc, boxes, scale_factor, img_hw, out_hw=(256, 128)):
    """det_input_hwc: the detector's normalised, padded input of ONE frame [Hp][Wp][3 or 4] -> [n][256][128][c]"""
    r = crop_rects(boxes, scale_factor, img_hw)
    return np.stack([interpolate_bilinear(det_input_hwc[y1:y2, x1:x2], out_hw) for x1, y1, x2, y2 in r]) if len(r) else \
        np.zeros((0, *out_hw, det_input_hwc.shape[2]), f32)


class ReidNetRef:
    def __init__(self, sd):
        self.sd = sd
        self.backbone = FasterRCNNRef(sd, prefix="")          # same ResNet-50 wiring / names ("backbone.*")

    def forward(self, crops_nhwc3):
        c5 = self.backbone.backbone(np.ascontiguousarray(crops_nhwc3[..., :3], f32))[3]      # [n][8][4][2048]
        n = c5.shape[0]
        acc = np.zeros((n, c5.shape[3]), f32)
        for a in range(8):                                                                 # AvgPool2d((8, 4)): (kh, kw) order
            for b in range(4):
                acc = (acc + c5[:, a, b]).astype(f32)
        g = (acc / f32(32)).astype(f32)
        sd = self.sd
        w, b = _fold_fc(sd)
        f = relu(clib.conv2d_nhwc(g[:, None, None, :], w, b))
        out = clib.conv2d_nhwc(f, sd["head.fc_out.weight"][:, :, None, None], sd["head.fc_out.bias"])
        return out.reshape(n, 128)


def _fold_fc(sd):
    """Linear (with bias) + BatchNorm1d folded like the product folds conv + BN: float64, one rounding"""
    w = sd["head.fcs.0.fc.weight"].astype(np.float64)
    s = sd["head.fcs.0.bn.weight"].astype(np.float64) / np.sqrt(sd["head.fcs.0.bn.running_var"].astype(np.float64) + 1e-5)
    b = (sd["head.fcs.0.fc.bias"].astype(np.float64) - sd["head.fcs.0.bn.running_mean"].astype(np.float64)) * s + \
        sd["head.fcs.0.bn.bias"].astype(np.float64)
    return (w * s[:, None]).astype(f32)[:, :, None, None], b.astype(f32)


def cdist(a, b):
    d = a.astype(f32)[:, None, :].astype(np.float64) - b.astype(f32)[None, :, :].astype(np.float64)
    return np.sqrt((d * d).sum(-1)).astype(f32)


class SortReidTrackerRef:
    def __init__(self, obj_score_thr=0.5, match_iou_thr=0.5, match_score_thr=2.0, num_samples=10, num_tentatives=2,
                 num_frames_retain=100):
        self.obj_score_thr, self.match_iou_thr, self.match_score_thr = obj_score_thr, match_iou_thr, match_score_thr
        self.num_samples, self.num_tentatives, self.retain = num_samples, num_tentatives, num_frames_retain
        self.kf = KalmanRef()
        self.tracks = {}
        self.num_tracks = 0

    def keep(self, dets):
        return np.asarray(dets, f32).reshape(-1, 5)[:, 4] > f32(self.obj_score_thr)

    def step(self, dets, embeds, frame_id):
        dets = np.asarray(dets, f32).reshape(-1, 5)
        n = len(dets)
        ids = np.full(n, -1, np.int64)
        zs = [xyxy_to_cxcyah(d[:4]) for d in dets]
        if self.tracks and n:
            costs = []
            for t in self.tracks.values():
                t["mean"], t["cov"] = self.kf.predict(t["mean"], t["cov"])
                pm = self.kf.H @ t["mean"]
                h = t["mean"][3]
                pc = np.linalg.multi_dot((self.kf.H, t["cov"], self.kf.H.T)) + np.diag(np.square([h / 20, h / 20, 1e-1, h / 20]))
                chol = np.linalg.cholesky(pc)
                d = np.array(zs) - pm
                z = np.linalg.solve(chol, d.T)          # triangular system; same values as scipy.linalg.solve_triangular
                costs.append(np.sum(z * z, axis=0))
            costs = np.stack(costs)
            gated = costs > CHI2INV95_4
            all_ids = list(self.tracks)
            active = [i for i in all_ids if not self.tracks[i]["tentative"]]
            if active:
                te = []
                for i in active:
                    e = self.tracks[i]["embeds"][-self.num_samples:]
                    acc = np.zeros_like(e[0])
                    for v in e:
                        acc = (acc + v).astype(f32)
                    te.append((acc / f32(len(e))).astype(f32))
                dist = cdist(np.stack(te), embeds)
                g = gated[[all_ids.index(i) for i in active]]
                row, col = linear_sum_assignment(np.where(g, GATED, dist.astype(np.float64)))
                for r, c in zip(row, col):
                    if g[r, c]:
                        continue
                    if dist[r, c] <= self.match_score_thr:
                        ids[c] = active[r]
            active = [i for i in all_ids if i not in ids and self.tracks[i]["frame"] == frame_id - 1]
            if active:
                free = np.flatnonzero(ids == -1)
                if len(free):
                    dists = (f32(1) - bbox_overlaps(np.stack([self.tracks[i]["box"] for i in active]), dets[free, :4])).astype(np.float64)
                    row, col = linear_sum_assignment(dists)
                    for r, c in zip(row, col):
                        if dists[r, c] < 1 - self.match_iou_thr:
                            ids[free[c]] = active[r]
        for k in range(n):
            if ids[k] < 0:
                ids[k] = self.num_tracks
                self.num_tracks += 1
        for i, d, z, e in zip(ids.tolist(), dets, zs, np.asarray(embeds, f32).reshape(n, -1) if n else []):
            if i in self.tracks:
                t = self.tracks[i]
                t["mean"], t["cov"] = self.kf.update(t["mean"], t["cov"], z)
                t["box"], t["frame"], t["n"] = d[:4].copy(), frame_id, t["n"] + 1
                t["embeds"].append(e.copy())
                if t["tentative"] and t["n"] >= self.num_tentatives:
                    t["tentative"] = False
            else:
                mean, cov = self.kf.initiate(z)
                self.tracks[i] = dict(mean=mean, cov=cov, box=d[:4].copy(), frame=frame_id, n=1, tentative=True, embeds=[e.copy()])
        for i in [i for i, t in self.tracks.items() if frame_id - t["frame"] >= self.retain or (t["tentative"] and t["frame"] != frame_id)]:
            del self.tracks[i]
        return np.concatenate([ids[:, None].astype(f32), dets], axis=1).astype(f32) if n else np.zeros((0, 6), f32)
