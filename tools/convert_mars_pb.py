#!/usr/bin/env python
"""mars-small128.pb (frozen TensorFlow GraphDef, pose_pipeline/wrappers/deep_sort_yolov4/parser.py:41-42) -> .npz with the
parameter names of posepipeline_amd/models/mars.py.  Optional: the wrapper reads the .pb directly (pure-Python protobuf wire
decoder, posepipeline_amd/checkpoints_tf.py); this tool exists to inspect / cache the result.  No TensorFlow needed.

  python tools/convert_mars_pb.py $PIPELINE_3RDPARTY/deep_sort_yolov4/mars-small128.pb [out.npz]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from posepipeline_amd import checkpoints_tf as ck  # noqa: E402
from posepipeline_amd.models import mars  # noqa: E402


def main():
    src = sys.argv[1]
    dst = sys.argv[2] if len(sys.argv) > 2 else os.path.splitext(src)[0] + ".npz"
    consts = ck.read_graphdef_consts(src)
    print(f"{src}: {len(consts)} float constants")
    sd = ck.mars_params_from_consts(consts, mars.mars_param_shapes())
    np.savez(dst, **sd)
    print(f"{dst}: {len(sd)} arrays, {sum(v.size for v in sd.values()) / 1e6:.2f} M parameters")


if __name__ == "__main__":
    main()
