"""fp16-form vs exact on the DeepSortYOLOv4 / YOLOX detectors (concat slices, SPP max-pools, Mish / Swish epilogues): finite, close."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from posepipeline_amd import _lib as L
from posepipeline_amd.models import synth, yolov4, yolox
from tests.test_gpu_detector import synth_frame
ctx = L.Context(0)
rng = np.random.default_rng(5)
frames = np.stack([synth_frame(rng, 1080, 1920) for _ in range(2)])
sd = yolov4.synth_params(yolov4.yolov4_param_shapes(80), seed=4, head_bias=0.7)
outs = {}
for mode in ("exact", "split_f16", "split_bf16"):
    det = yolov4.YoloV4Detector(ctx, sd, 1080, 1920, max_frames=2, numerics=mode)
    det.run(frames)
    o = []
    for name in ("y19", "y38", "y76"):
        dptr, _, _ = det.net.buffer(name)
        h = {"y19": 13, "y38": 26, "y76": 52}[name]
        buf = np.empty((2, h, h, 255), np.float32)
        ctx.d2h(buf, int(dptr))
        o.append(buf)
    outs[mode] = o
    print(mode, "kinds", np.bincount(det.net.conv_kinds()), [bool(np.isfinite(a).all()) for a in o])
    det.close()
for mode in ("split_f16", "split_bf16"):
    print(mode, [float(np.abs(a - b).max() / np.abs(b).max()) for a, b in zip(outs[mode], outs["exact"])])
sdx = yolox.seed_synthetic_head(synth.synth_state_dict(yolox.yolox_param_shapes(), seed=6), -4.5)
res = {}
for mode in ("exact", "split_f16"):
    det = yolox.YoloXDetector(ctx, sdx, 1080, 1920, max_frames=1, numerics=mode)
    res[mode] = det.run(frames[:1])[0]
    print("yolox", mode, res[mode].shape, bool(np.isfinite(res[mode]).all()))
    det.close()
# near-equal scores of a seeded-random head come out in another order and a few candidates sit on the 0.01 threshold: match by box
a, b = res["exact"], res["split_f16"]
d = np.abs(a[:, None, :4] - b[None, :, :4]).max(axis=2)
j = d.argmin(axis=1)
ok = d[np.arange(len(a)), j] <= 1e-2
print(f"yolox: {int(ok.sum())} of {len(a)} exact detections re-found within 0.01 px; score difference of those {float(np.abs(a[ok, 4] - b[j[ok], 4]).max()):.2e}")
