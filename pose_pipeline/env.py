"""Module-path shim (see pose_pipeline/__init__.py): this name IS posepipeline_amd.env."""
import importlib
import sys

sys.modules[__name__] = importlib.import_module("posepipeline_amd.env")
