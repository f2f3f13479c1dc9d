"""CPU oracle for the NV12 frame source.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference hands BGR frames from `cv2.VideoCapture.read()` to its models (pose_pipeline/pipeline.py:47-87
get_robust_reader, wrappers/mmtrack.py:38-45, wrappers/mmpose.py:55-75).  A decoder's native output is NV12 (Y plane, then an
interleaved half-resolution UV plane); the drop-in uploads NV12 and converts on the device (csrc/nv12.hip).  This file restates
the conversion OpenCV publishes for `cv2.cvtColor(yuv, cv2.COLOR_YUV2BGR_NV12)` on 8-bit images (opencv/modules/imgproc/src/
color_yuv.simd.hpp, OpenCV 4.x: `uvToRGBuv` / `yRGBuvToRGBA`): ITU-R BT.601 limited range, 20-bit fixed point,

    y   = max(0, Y - 16) * 1220542
    B   = sat8((y + (1 << 19) + 2116026 (U - 128)) >> 20)
    G   = sat8((y + (1 << 19) -  852492 (V - 128) - 409993 (U - 128)) >> 20)
    R   = sat8((y + (1 << 19) + 1673527 (V - 128)) >> 20)

OpenCV is an un-vendored third-party dependency (requirements.txt) and is not installed in this image: PARITY UNPINNED
against the real package (the generator tests/golden/make_goldens_3p.py records cv2's own output when it is run where cv2
exists).
"""
from __future__ import annotations

import numpy as np

CY, CUB, CUG, CVG, CVR, SHIFT = 1220542, 2116026, -409993, -852492, 1673527, 20


def nv12_to_bgr(nv12: np.ndarray, h: int, w: int) -> np.ndarray:
    """nv12 [n][h * 3 / 2][w] (or [n][h * w * 3 / 2]) u8 -> [n][h][w][3] u8 BGR"""
    a = np.ascontiguousarray(nv12, np.uint8).reshape(-1, h * 3 // 2, w)
    n = a.shape[0]
    y = np.maximum(a[:, :h].astype(np.int64) - 16, 0) * CY
    uv = a[:, h:].reshape(n, h // 2, w // 2, 2).astype(np.int64) - 128
    u = np.repeat(np.repeat(uv[..., 0], 2, axis=1), 2, axis=2)
    v = np.repeat(np.repeat(uv[..., 1], 2, axis=1), 2, axis=2)
    half = 1 << (SHIFT - 1)
    b = (y + half + CUB * u) >> SHIFT
    g = (y + half + CVG * v + CUG * u) >> SHIFT
    r = (y + half + CVR * v) >> SHIFT
    return np.clip(np.stack([b, g, r], axis=-1), 0, 255).astype(np.uint8)
