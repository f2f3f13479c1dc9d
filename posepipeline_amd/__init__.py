"""posepipeline_amd -- MI355X-native hot path of PosePipe's detect/track -> 2D -> 3D cascade.

The compute lives in libposepipe_hip.so (posepipeline_amd/csrc, C ABI in include/posepipe_hip.h);
this package is the thin host side: ctypes binding, layer-program builders for the backbones and the
drop-in wrappers that mirror pose_pipeline/wrappers/{mmtrack,mmpose,videopose3d}.py.
"""
__version__ = "0.1.0"
