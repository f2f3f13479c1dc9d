#!/usr/bin/env python
"""Equivalent of the reference's scripts/process_h36m.py:1-17 on this package: every video of the "h36m" project goes
through the standard recipe with the DeepSortYOLOv4 tracker and the Halpe top-down method -- the same call as line 15,
    top_down_pipeline(k, top_down_method_name="MMPoseHalpe", tracking_method_name='DeepSortYOLOv4')
`--top-down-method ViTPoseH` runs BASELINE.json configs[4] instead (ViTPose-H, bf16 MFMA encoder; a lookup row this
package adds, not one of the reference's).
Tables: posepipeline_amd.pipeline (in-memory shim, or real DataJoint with POSEPIPE_USE_DATAJOINT=1).
Set POSEPIPE_SYNTHETIC_WEIGHTS=1 when no checkpoints are installed."""
import argparse

from posepipeline_amd.pipeline import Video
from posepipeline_amd.utils.standard_pipelines import top_down_pipeline

VIDEO_PROJECT = "h36m"


def main(top_down_method_name="MMPoseHalpe", tracking_method_name="DeepSortYOLOv4"):
    keys = (Video & f'video_project="{VIDEO_PROJECT}"').fetch("KEY")
    for k in keys:
        top_down_pipeline(k, top_down_method_name=top_down_method_name, tracking_method_name=tracking_method_name)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--top-down-method", default="MMPoseHalpe")
    ap.add_argument("--tracking-method", default="DeepSortYOLOv4")
    a = ap.parse_args()
    main(a.top_down_method, a.tracking_method)
