"""Module-path shim: `import pose_pipeline...` resolves to the MI355X package (posepipeline_amd).

The reference's make() methods import the wrappers by these paths (pose_pipeline/pipeline.py:525-547 `from
pose_pipeline.wrappers.mmtrack import mmtrack_bounding_boxes`, :1020-1039 `...wrappers.mmpose import
mmpose_top_down_person`, :1270-1273 `...wrappers.videopose3d import process_videopose3d`, :519-523
`...wrappers.deep_sort_yolov4.parser import tracking_bounding_boxes`), and scripts use `from pose_pipeline import *` and
`pose_pipeline.utils.standard_pipelines`.  Putting this directory (the repository root) on sys.path BEFORE a reference
checkout makes every one of those imports land on the drop-in, with no edit to the caller.
Each sub-module below replaces itself in sys.modules with the posepipeline_amd module of the same role, so
`pose_pipeline.wrappers.mmpose is posepipeline_amd.wrappers.mmpose` (one module object, one model cache).
Only the hot-path modules exist here; everything else of the reference (SMPL, OpenPose, hand / face wrappers ...) is out
of scope and raises ImportError as an absent module should.
"""
import os

from posepipeline_amd.pipeline import (BestDetectedFrames, DetectedFrames, LiftingMethod, LiftingMethodLookup,  # noqa: F401
                                       LiftingPerson, PersonBbox, PersonBboxValid, TopDownMethod, TopDownMethodLookup,
                                       TopDownPerson, TrackingBbox, TrackingBboxMethod, TrackingBboxMethodLookup, Video,
                                       VideoInfo)
from posepipeline_amd.weights import model_data_dir as _model_data_dir

# pose_pipeline/__init__.py:19 of the reference; `pose_pipeline.env` is an attribute scripts use (scripts/process_h36m.py:7-8)
from . import env  # noqa: E402,F401  (the sub-module replaces itself with posepipeline_amd.env)
from posepipeline_amd.env import (add_path, pytorch_memory_limit, set_environmental_variables,  # noqa: E402,F401
                                  tensorflow_memory_limit)

# pose_pipeline/__init__.py:24-27 of the reference: $PIPELINE_3RDPARTY, else <checkout>/3rdparty
MODEL_DATA_DIR = _model_data_dir()

__all__ = ["Video", "VideoInfo", "TrackingBboxMethodLookup", "TrackingBboxMethod", "TrackingBbox", "PersonBboxValid",
           "PersonBbox", "DetectedFrames", "BestDetectedFrames", "TopDownMethodLookup", "TopDownMethod", "TopDownPerson", "LiftingMethodLookup",
           "LiftingMethod", "LiftingPerson", "MODEL_DATA_DIR", "add_path", "set_environmental_variables",
           "pytorch_memory_limit", "tensorflow_memory_limit"]
