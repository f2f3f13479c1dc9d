"""CPU oracle for flip-merge + heatmap decode.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates the numpy tail of mmpose 0.x `TopDown.forward_test` that the wrapper reaches at
pose_pipeline/wrappers/mmpose.py:75 with test_cfg = dict(flip_test=True, post_process='unbiased',
shift_heatmap=True, modulate_kernel=17) (3rdparty/mmpose/config/top_down/darkpose/coco/
hrnet_w48_coco_384x288_dark.py:81-85):
    head.inference_model: flip_back, shift_heatmap;  (hm + hm_flipped) * 0.5
    keypoints_from_heatmaps: _get_max_preds, _gaussian_blur, log, _taylor, transform_preds
The reference tree holds the same maths once, in float64 and with a bbox back-map, as
pose_pipeline/utils/inference.py:27-114 (copied from DarkPose); `get_max_preds`, `taylor` and
`transform_preds` of that file are importable here and pin the corresponding functions below through
tests/golden/dark_decode.npz (tests/test_oracle_golden.py).  `gaussian_blur` needs cv2.GaussianBlur,
which is not installed: the separable float32 blur below restates OpenCV's published algorithm
(getGaussianKernel sigma rule, generic row filter summed k = 0..ksize-1, symmetric column filter
summed centre-out in pairs) -- PARITY UNPINNED for that one step.

dtype notes (mmpose 0.x ran on NumPy 1.x): scalar float32 (op) python-float promotes to float64,
float32-array (op) scalar stays float32.  Casts are explicit here so the result does not depend on
the NumPy version.
"""
from __future__ import annotations

import numpy as np

f32 = np.float32
f64 = np.float64


def flip_merge(hm, hm_flipped, flip_pairs, shift_heatmap=True):
    """(output_heatmap + flip_back(output_flipped_heatmap)) * 0.5, float32."""
    hm = np.asarray(hm, f32)
    back = np.asarray(hm_flipped, f32).copy()
    for left, right in flip_pairs:
        back[:, left], back[:, right] = hm_flipped[:, right], hm_flipped[:, left]
    back = back[..., ::-1].copy()
    if shift_heatmap:
        back[:, :, :, 1:] = back[:, :, :, :-1].copy()
    return ((hm + back) * f32(0.5)).astype(f32)


def get_max_preds(heatmaps):
    """mmpose `_get_max_preds`: float32 preds, -1 where maxval <= 0 (in-tree variant zeroes instead)."""
    n, k, _, w = heatmaps.shape
    flat = heatmaps.reshape(n, k, -1)
    idx = np.argmax(flat, 2).reshape(n, k, 1)
    maxvals = np.amax(flat, 2).reshape(n, k, 1)
    preds = np.tile(idx, (1, 1, 2)).astype(f32)
    preds[:, :, 0] = preds[:, :, 0] % w
    preds[:, :, 1] = preds[:, :, 1] // w
    preds = np.where(np.tile(maxvals, (1, 1, 2)) > 0.0, preds, f32(-1)).astype(f32)
    return preds, maxvals


def gaussian_kernel1d(ksize):
    """cv::getGaussianKernel(ksize, sigma=0, CV_32F)."""
    sigma = 0.3 * ((ksize - 1) * 0.5 - 1) + 0.8
    scale2x = -0.5 / (sigma * sigma)
    x = np.arange(ksize, dtype=f64) - (ksize - 1) * 0.5
    t = np.exp(scale2x * x * x)
    t = t * (1.0 / t.sum())
    return t.astype(f32)


def gaussian_blur_f32(hm2d, ksize):
    """cv2.GaussianBlur(zero-padded hm, (ksize, ksize), 0) cropped back, float32, no FMA."""
    h, w = hm2d.shape
    r = ksize // 2
    k = gaussian_kernel1d(ksize)
    pad = np.zeros((h, w + 2 * r), dtype=f32)
    pad[:, r:r + w] = hm2d
    # row filter: s = k[0]*S[0]; s += k[j]*S[j] ...
    row = (k[0] * pad[:, 0:w]).astype(f32)
    for j in range(1, ksize):
        row = (row + (k[j] * pad[:, j:j + w]).astype(f32)).astype(f32)
    # symmetric column filter: s = k[c]*S[c]; s += k[c+j]*(S[c+j] + S[c-j])
    padv = np.zeros((h + 2 * r, w), dtype=f32)
    padv[r:r + h] = row
    out = (k[r] * padv[r:r + h]).astype(f32)
    for j in range(1, r + 1):
        pair = (padv[r + j:r + j + h] + padv[r - j:r - j + h]).astype(f32)
        out = (out + (k[r + j] * pair).astype(f32)).astype(f32)
    return out


def gaussian_blur(heatmaps, kernel=11):
    """mmpose `_gaussian_blur` (in-tree: utils/inference.py:78-92): blur, then rescale to the old max."""
    heatmaps = heatmaps.copy()
    n, k = heatmaps.shape[:2]
    for i in range(n):
        for j in range(k):
            origin_max = np.max(heatmaps[i, j])
            dr = gaussian_blur_f32(heatmaps[i, j], kernel)
            heatmaps[i, j] = dr
            # (an all-zero map blurs to all zeros: 0 / 0 -- numpy warns and yields NaN, as the reference's own line does; the
            # product path's maps are never all zero after the flip average, the oracle is only spared the warning)
            with np.errstate(divide="ignore", invalid="ignore"):
                heatmaps[i, j] *= f32(origin_max / np.max(heatmaps[i, j]))
    return heatmaps


def taylor(heatmap, coord):
    """mmpose `_taylor` (in-tree: utils/inference.py:57-75) with NumPy-1.x scalar promotion spelled out."""
    h, w = heatmap.shape[:2]
    px, py = int(coord[0]), int(coord[1])
    if 1 < px < w - 2 and 1 < py < h - 2:
        hm = heatmap
        dx = 0.5 * f64(f32(hm[py][px + 1] - hm[py][px - 1]))
        dy = 0.5 * f64(f32(hm[py + 1][px] - hm[py - 1][px]))
        dxx = 0.25 * ((f64(hm[py][px + 2]) - 2.0 * f64(hm[py][px])) + f64(hm[py][px - 2]))
        dxy = 0.25 * f64(f32(f32(f32(hm[py + 1][px + 1] - hm[py - 1][px + 1]) - hm[py + 1][px - 1]) + hm[py - 1][px - 1]))
        dyy = 0.25 * ((f64(hm[py + 2][px]) - 2.0 * f64(hm[py][px])) + f64(hm[py - 2][px]))
        derivative = np.array([[dx], [dy]], dtype=f64)
        hessian = np.array([[dxx, dxy], [dxy, dyy]], dtype=f64)
        if dxx * dyy - dxy ** 2 != 0:
            hessianinv = np.linalg.inv(hessian)
            offset = -hessianinv @ derivative
            offset = np.squeeze(np.array(offset.T), axis=0)
            coord = (coord.astype(f64) + offset).astype(f32)
    return coord


def transform_preds(coords, center, scale, output_size):
    """mmpose `transform_preds` (use_udp=False); coords float32 [K][2]."""
    scale = (np.asarray(scale, f32) * f32(200.0)).astype(f32)
    scale_x = f32(f64(scale[0]) / output_size[0])       # float64 scalar, cast when it meets the f32 array
    scale_y = f32(f64(scale[1]) / output_size[1])
    out = np.ones_like(coords, dtype=f32)
    out[:, 0] = ((coords[:, 0] * scale_x).astype(f32) + f32(center[0])).astype(f32) - f32(f64(scale[0]) * 0.5)
    out[:, 1] = ((coords[:, 1] * scale_y).astype(f32) + f32(center[1])).astype(f32) - f32(f64(scale[1]) * 0.5)
    return out.astype(f32)


def keypoints_from_heatmaps(heatmaps, center, scale, post_process="unbiased", kernel=17):
    """-> (preds [N][K][2] float32 image px, maxvals [N][K][1] float32)."""
    heatmaps = np.asarray(heatmaps, f32).copy()
    n, k, h, w = heatmaps.shape
    preds, maxvals = get_max_preds(heatmaps)
    if post_process == "unbiased":
        # float32 log, evaluated in double and rounded once so that the value does not depend on a libm
        # (numpy's own float32 log is within 1 ulp of this; DARK's Hessian can amplify that ulp on flat maps)
        heatmaps = np.log(np.maximum(gaussian_blur(heatmaps, kernel), f32(1e-10)).astype(f64)).astype(f32)
        for i in range(n):
            for j in range(k):
                preds[i][j] = taylor(heatmaps[i][j], preds[i][j])
    elif post_process is not None:
        for i in range(n):
            for j in range(k):
                hm = heatmaps[i][j]
                px, py = int(preds[i][j][0]), int(preds[i][j][1])
                if 1 < px < w - 1 and 1 < py < h - 1:
                    diff = np.array([hm[py][px + 1] - hm[py][px - 1], hm[py + 1][px] - hm[py - 1][px]], dtype=f32)
                    preds[i][j] += (np.sign(diff) * f32(0.25)).astype(f32)
    for i in range(n):
        preds[i] = transform_preds(preds[i], center[i], scale[i], [w, h])
    return preds, maxvals


def decode_topdown(hm, hm_flipped, flip_pairs, center, scale, post_process="unbiased", kernel=17, shift_heatmap=True):
    """Full tail: returns (keypoints [N][K][3] float32 = (x, y, score), merged heatmap)."""
    merged = flip_merge(hm, hm_flipped, flip_pairs, shift_heatmap) if hm_flipped is not None else np.asarray(hm, f32)
    preds, maxvals = keypoints_from_heatmaps(merged, center, scale, post_process, kernel)
    out = np.zeros((merged.shape[0], merged.shape[1], 3), dtype=f32)
    out[:, :, 0:2] = preds
    out[:, :, 2:3] = maxvals
    return out, merged


# ---- UDP variant (ViTPose configs: test_cfg use_udp=True, post_process='default', modulate_kernel=11, shift_heatmap=False)
# NOT in the reference tree; restated from mmpose 0.x `post_dark_udp` / `transform_preds(use_udp=True)`: parity unpinned.
def gaussian_blur_reflect_f32(hm2d, ksize):
    """cv2.GaussianBlur(hm, (ksize, ksize), 0) in place on the map itself: BORDER_REFLECT_101, float32, no FMA;
    same summation orders as gaussian_blur_f32."""
    h, w = hm2d.shape
    r = ksize // 2
    k = gaussian_kernel1d(ksize)
    pad = np.pad(hm2d.astype(f32), ((0, 0), (r, r)), mode="reflect")
    row = (k[0] * pad[:, 0:w]).astype(f32)
    for j in range(1, ksize):
        row = (row + (k[j] * pad[:, j:j + w]).astype(f32)).astype(f32)
    padv = np.pad(row, ((r, r), (0, 0)), mode="reflect")
    out = (k[r] * padv[r:r + h]).astype(f32)
    for j in range(1, r + 1):
        pair = (padv[r + j:r + j + h] + padv[r - j:r - j + h]).astype(f32)
        out = (out + (k[r + j] * pair).astype(f32)).astype(f32)
    return out


def post_dark_udp(coords, heatmaps, kernel=11):
    """coords [N][K][2] float32 (argmax), heatmaps [N][K][H][W] float32 -> refined coords (float32).
    Deviation, stated: joints without a peak (maxval <= 0, coords -1) are left alone -- mmpose indexes the flattened,
    padded batch with -1 there and reads an unrelated map."""
    n, kk, h, w = heatmaps.shape
    out = coords.astype(f32).copy()
    for i in range(n):
        for j in range(kk):
            px, py = int(coords[i, j, 0]), int(coords[i, j, 1])
            if px < 0:
                continue
            hm = gaussian_blur_reflect_f32(heatmaps[i, j], kernel)
            hm = np.log(np.clip(hm, f32(0.001), f32(50.0)).astype(f64)).astype(f32)
            p = np.pad(hm, 1, mode="edge")
            x, y = px + 1, py + 1
            i_, ix1, iy1, ix1y1 = p[y, x], p[y, x + 1], p[y + 1, x], p[y + 1, x + 1]
            ix1_y1_, ix1_, iy1_ = p[y - 1, x - 1], p[y, x - 1], p[y - 1, x]
            dx = f32(0.5) * f32(ix1 - ix1_)
            dy = f32(0.5) * f32(iy1 - iy1_)
            dxx = f32(f32(ix1 - f32(f32(2.0) * i_)) + ix1_)
            dyy = f32(f32(iy1 - f32(f32(2.0) * i_)) + iy1_)
            t = f32(ix1y1 - ix1)
            for v, sgn in ((iy1, -1), (i_, 1), (i_, 1), (ix1_, -1), (iy1_, -1), (ix1_y1_, 1)):
                t = f32(t + v) if sgn > 0 else f32(t - v)
            dxy = f32(0.5) * t
            hess = np.array([[dxx, dxy], [dxy, dyy]], f64) + f64(np.finfo(f32).eps) * np.eye(2)
            off = np.linalg.inv(hess) @ np.array([dx, dy], f64)
            out[i, j, 0] = f32(f64(out[i, j, 0]) - off[0])
            out[i, j, 1] = f32(f64(out[i, j, 1]) - off[1])
    return out


def transform_preds_udp(coords, center, scale, output_size):
    scale = (np.asarray(scale, f32) * f32(200.0)).astype(f32)
    scale_x = f32(f64(scale[0]) / (output_size[0] - 1.0))
    scale_y = f32(f64(scale[1]) / (output_size[1] - 1.0))
    out = np.ones_like(coords, dtype=f32)
    out[:, 0] = ((coords[:, 0] * scale_x).astype(f32) + f32(center[0])).astype(f32) - f32(f64(scale[0]) * 0.5)
    out[:, 1] = ((coords[:, 1] * scale_y).astype(f32) + f32(center[1])).astype(f32) - f32(f64(scale[1]) * 0.5)
    return out.astype(f32)


def decode_topdown_udp(hm, hm_flipped, flip_pairs, center, scale, kernel=11):
    """flip-merge (no shift) + DARK-UDP + UDP back-mapping -> (keypoints [N][K][3], merged heatmap)"""
    merged = flip_merge(hm, hm_flipped, flip_pairs, False) if hm_flipped is not None else np.asarray(hm, f32)
    n, k, h, w = merged.shape
    preds, maxvals = get_max_preds(merged)
    preds = post_dark_udp(preds, merged, kernel)
    for i in range(n):
        preds[i] = transform_preds_udp(preds[i], center[i], scale[i], [w, h])
    out = np.zeros((n, k, 3), dtype=f32)
    out[:, :, 0:2] = preds
    out[:, :, 2:3] = maxvals
    return out, merged
