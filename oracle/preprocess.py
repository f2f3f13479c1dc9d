"""CPU oracle for top-down pre-processing.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates what `inference_top_down_pose_model(model, frame, [{"bbox": bbox}], format='xywh')`
(pose_pipeline/wrappers/mmpose.py:75) does before the network, i.e. the test pipeline of
3rdparty/mmpose/config/top_down/darkpose/coco/hrnet_w48_coco_384x288_dark.py:129-144:
    _box2cs -> LoadImage (BGR->RGB swap of an ndarray) -> TopDownAffine -> ToTensor -> NormalizeTensor
mmpose 0.x and OpenCV are un-vendored third-party dependencies (requirements.txt:1-2 pins nothing
usable; mmpose is a comment at :12) and are not installed here: PARITY UNPINNED against the real
packages.  The published algorithms restated here:
  * mmpose 0.x `_box2cs`, `get_affine_transform`, `_get_3rd_point` (float32 points, float64 sums);
  * cv2.getAffineTransform (6x6 system, LU with partial pivoting, double);
  * cv2.warpAffine(INTER_LINEAR, BORDER_CONSTANT=0) for 8-bit images in OpenCV <= 4.10: inverse
    map in double, AB_BITS=10 fixed-point coordinates, INTER_BITS=5 sub-pixel steps, bilinear
    weights as exact multiples of 1/1024 scaled to 2^15, result (sum + 2^14) >> 15;
  * torchvision to_tensor (/255 in float32) and normalize ((x - mean) / std in float32).
The in-tree twin of the crop idiom (different bbox convention) is
pose_pipeline/utils/bounding_box.py:32-53; `fix_bb_aspect_ratio` there is pinned by a golden fixture.
"""
from __future__ import annotations

import numpy as np

MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)
STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)


def box2cs(bbox, image_size=(288, 384)):
    """mmpose 0.x `_box2cs`: bbox (x, y, w, h) float64 -> (center float32[2], scale float32[2])."""
    x, y, w, h = (float(v) for v in bbox[:4])
    aspect_ratio = image_size[0] / image_size[1]
    center = np.array([x + w * 0.5, y + h * 0.5], dtype=np.float32)
    if w > aspect_ratio * h:
        h = w * 1.0 / aspect_ratio
    elif w < aspect_ratio * h:
        w = h * aspect_ratio
    scale = np.array([w / 200.0, h / 200.0], dtype=np.float32)
    scale = (scale * np.float32(1.25)).astype(np.float32)
    return center, scale


def _third_point(a, b):
    direction = (a - b).astype(np.float32)
    return (b + np.array([-direction[1], direction[0]], dtype=np.float32)).astype(np.float32)


def affine_points(center, scale, output_size):
    """mmpose `get_affine_transform(center, scale, rot=0, output_size)`: the two float32 triangles."""
    scale_tmp = (scale * np.float32(200.0)).astype(np.float32)
    src_w = np.float64(scale_tmp[0])
    dst_w, dst_h = float(output_size[0]), float(output_size[1])
    src_dir = np.array([0.0, src_w * -0.5], dtype=np.float64)     # rotate_point(..., 0)
    dst_dir = np.array([0.0, dst_w * -0.5], dtype=np.float64)
    src = np.zeros((3, 2), dtype=np.float32)
    src[0, :] = center
    src[1, :] = (center.astype(np.float64) + src_dir).astype(np.float32)
    src[2, :] = _third_point(src[0, :], src[1, :])
    dst = np.zeros((3, 2), dtype=np.float32)
    dst[0, :] = [dst_w * 0.5, dst_h * 0.5]
    dst[1, :] = (np.array([dst_w * 0.5, dst_h * 0.5]) + dst_dir).astype(np.float32)
    dst[2, :] = _third_point(dst[0, :], dst[1, :])
    return src, dst


def get_affine_transform_cv(src, dst):
    """cv2.getAffineTransform(src, dst) -> 2x3 float64 (zeros when the system is singular)."""
    a = np.zeros((6, 6), dtype=np.float64)
    b = np.zeros(6, dtype=np.float64)
    for i in range(3):
        a[2 * i, 0:3] = [src[i, 0], src[i, 1], 1.0]
        a[2 * i + 1, 3:6] = [src[i, 0], src[i, 1], 1.0]
        b[2 * i], b[2 * i + 1] = dst[i, 0], dst[i, 1]
    try:
        m = np.linalg.solve(a, b)
    except np.linalg.LinAlgError:
        m = np.zeros(6)
    return m.reshape(2, 3)


def _cv_round(v):
    """cvRound / saturate_cast<int>(double): round half to even, saturating to int32."""
    return np.clip(np.rint(v), -2147483648.0, 2147483647.0).astype(np.int64)


def warp_affine_u8(img, m_fwd, dsize):
    """cv2.warpAffine(img, M, dsize, flags=INTER_LINEAR) for HxWxC uint8, border constant 0."""
    img = np.asarray(img)
    assert img.dtype == np.uint8 and img.ndim == 3
    h, w, c = img.shape
    dw, dh = int(dsize[0]), int(dsize[1])
    m = np.array(m_fwd, dtype=np.float64).reshape(6).copy()
    d = m[0] * m[4] - m[1] * m[3]
    d = 1.0 / d if d != 0 else 0.0
    a11, a22 = m[4] * d, m[0] * d
    m[0] = a11
    m[1] *= -d
    m[3] *= -d
    m[4] = a22
    b1 = -m[0] * m[2] - m[1] * m[5]
    b2 = -m[3] * m[2] - m[4] * m[5]
    m[2], m[5] = b1, b2
    xs = np.arange(dw, dtype=np.float64)
    ys = np.arange(dh, dtype=np.float64)
    adelta = _cv_round(m[0] * xs * 1024.0)
    bdelta = _cv_round(m[3] * xs * 1024.0)
    x0 = _cv_round((m[1] * ys + m[2]) * 1024.0) + 16
    y0 = _cv_round((m[4] * ys + m[5]) * 1024.0) + 16
    X = (x0[:, None] + adelta[None, :]) >> 5
    Y = (y0[:, None] + bdelta[None, :]) >> 5
    sx = np.clip(X >> 5, -32768, 32767)
    sy = np.clip(Y >> 5, -32768, 32767)
    fx, fy = X & 31, Y & 31
    w00, w01, w10, w11 = (32 - fx) * (32 - fy), fx * (32 - fy), (32 - fx) * fy, fx * fy

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
        v = img[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)].astype(np.int64)
        return v * ok[..., None]

    acc = (tap(sy, sx) * w00[..., None] + tap(sy, sx + 1) * w01[..., None] + tap(sy + 1, sx) * w10[..., None] +
           tap(sy + 1, sx + 1) * w11[..., None])
    return ((acc * 32 + 16384) >> 15).astype(np.uint8)


def normalize_lut(mean=MEAN, std=STD):
    """[3][256] float32: ((v / 255) - mean[c]) / std[c], every step rounded to float32 (to_tensor + normalize)."""
    v = (np.arange(256, dtype=np.float32) / np.float32(255.0)).astype(np.float32)
    return ((v[None, :] - mean[:, None]).astype(np.float32) / std[:, None]).astype(np.float32)


def top_down_input(frame_wrapper_rgb, bbox, image_size=(288, 384)):
    """frame_wrapper_rgb: the HxWx3 uint8 array the wrapper passes to mmpose, i.e. cv2's BGR frame
    after wrappers/mmpose.py:73 `cvtColor(BGR2RGB)`.  mmpose's LoadImage treats any ndarray as BGR and
    swaps again, so the tensor's channel 0 is the ORIGINAL B plane.
    Returns (tensor [3][H][W] float32, center, scale, crop_u8 [H][W][3] in tensor channel order)."""
    center, scale = box2cs(bbox, image_size)
    src, dst = affine_points(center, scale, image_size)
    trans = get_affine_transform_cv(src, dst)
    img = frame_wrapper_rgb[:, :, ::-1]                      # LoadImage: "BGR"->RGB swap
    crop = warp_affine_u8(np.ascontiguousarray(img), trans, image_size)
    lut = normalize_lut()
    t = np.stack([lut[c][crop[:, :, c]] for c in range(3)], axis=0)
    return t.astype(np.float32), center, scale, crop


# ---- UDP variant (ViTPose configs, `use_udp=True`; NOT in the reference tree: parity unpinned) -------------------
def get_warp_matrix_udp(center, scale, image_size):
    """mmpose `get_warp_matrix(theta=0, size_input=c * 2.0, size_dst=image_size - 1.0, size_target=s * 200.0)`:
    scalars in float64, the returned 2x3 matrix float32 (as mmpose builds it)."""
    size_input = (np.asarray(center, np.float32) * np.float32(2.0)).astype(np.float32)
    size_target = (np.asarray(scale, np.float32) * np.float32(200.0)).astype(np.float32)
    size_dst = np.asarray(image_size, np.float64) - 1.0
    m = np.zeros((2, 3), dtype=np.float32)
    sx = size_dst[0] / float(size_target[0])
    sy = size_dst[1] / float(size_target[1])
    m[0, 0] = 1.0 * sx
    m[0, 1] = -0.0 * sx
    m[0, 2] = sx * (-0.5 * float(size_input[0]) * 1.0 + 0.5 * float(size_input[1]) * 0.0 + 0.5 * float(size_target[0]))
    m[1, 0] = 0.0 * sy
    m[1, 1] = 1.0 * sy
    m[1, 2] = sy * (-0.5 * float(size_input[0]) * 0.0 - 0.5 * float(size_input[1]) * 1.0 + 0.5 * float(size_target[1]))
    return m


def top_down_input_udp(frame_wrapper_rgb, bbox, image_size=(192, 256)):
    """`top_down_input` with TopDownAffine(use_udp=True).  Same channel quirk as the non-UDP path."""
    center, scale = box2cs(bbox, image_size)
    m = get_warp_matrix_udp(center, scale, image_size)
    img_bgr_view = frame_wrapper_rgb[:, :, ::-1]
    crop = warp_affine_u8(np.ascontiguousarray(img_bgr_view), m.astype(np.float64), image_size)
    lut = normalize_lut()
    t = np.stack([lut[c][crop[:, :, c]] for c in range(3)], axis=0).astype(np.float32)
    return t, center, scale, crop
