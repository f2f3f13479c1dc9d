#!/usr/bin/env python
"""Equivalent of the reference's scripts/process_h36m.py:1-17 on this package: every video of the "h36m" project goes
through the standard recipe with the DeepSortYOLOv4 tracker and the Halpe top-down method -- the same call as line 15,
    top_down_pipeline(k, top_down_method_name="MMPoseHalpe", tracking_method_name='DeepSortYOLOv4')
Tables: posepipeline_amd.pipeline (in-memory shim, or real DataJoint with POSEPIPE_USE_DATAJOINT=1).
Set POSEPIPE_SYNTHETIC_WEIGHTS=1 when no checkpoints are installed."""
from posepipeline_amd.pipeline import Video
from posepipeline_amd.utils.standard_pipelines import top_down_pipeline

VIDEO_PROJECT = "h36m"


def main():
    keys = (Video & f'video_project="{VIDEO_PROJECT}"').fetch("KEY")
    for k in keys:
        top_down_pipeline(k, top_down_method_name="MMPoseHalpe", tracking_method_name="DeepSortYOLOv4")


if __name__ == "__main__":
    main()
