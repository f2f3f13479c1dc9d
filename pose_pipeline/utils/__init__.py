"""Module-path shim, see pose_pipeline/__init__.py."""
