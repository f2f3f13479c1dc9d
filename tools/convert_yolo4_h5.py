#!/usr/bin/env python
"""yolo4.h5 (Keras model file of tracking_method 0, pose_pipeline/wrappers/deep_sort_yolov4/yolo.py:51-54) -> yolo4.npz with
the parameter names of posepipeline_amd/models/yolov4.py.  Run ONCE on any machine that has h5py; the wrapper
(posepipeline_amd/wrappers/deep_sort_yolov4/parser.py) then loads the .npz next to the .h5 without h5py / TensorFlow.

  python tools/convert_yolo4_h5.py $PIPELINE_3RDPARTY/deep_sort_yolov4/yolo4.h5 [out.npz]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from posepipeline_amd import checkpoints_tf as ck  # noqa: E402
from posepipeline_amd.models import yolov4  # noqa: E402


def main():
    src = sys.argv[1]
    dst = sys.argv[2] if len(sys.argv) > 2 else os.path.splitext(src)[0] + ".npz"
    sd = ck.yolo_params_from_keras(ck.read_keras_h5(src), yolov4.yolov4_param_shapes())
    np.savez(dst, **sd)
    print(f"{dst}: {len(sd)} arrays, {sum(v.size for v in sd.values()) / 1e6:.1f} M parameters")


if __name__ == "__main__":
    main()
