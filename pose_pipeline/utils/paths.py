"""Module-path shim (see pose_pipeline/__init__.py): pose_pipeline/utils/paths.py's find_full_path lives in posepipeline_amd.paths."""
import importlib
import sys

sys.modules[__name__] = importlib.import_module("posepipeline_amd.paths")
