"""`annotate_single_person` under the reference's module name (pose_pipeline/utils/tracking.py:5); the implementation
lives with the recipes that call it."""
from .standard_pipelines import annotate_single_person  # noqa: F401
