"""Where the non-package dependencies and checkpoints live (pose_pipeline/paths.py:5-9, utils/paths.py:9-33)."""
from __future__ import annotations

import os
import pathlib


def _dj_config():
    # the same configuration object the tables use (real DataJoint when selected, else the in-memory shim)
    if os.environ.get("POSEPIPE_USE_DATAJOINT") == "1":
        import datajoint as dj
    else:
        from . import djshim as dj
    return dj.config


def get_pose_project_dir() -> str:
    """`dj.config["custom"]["pose_project_dir"]`: directory of the local installs, a string ending in "/".  The reference
    falls back to its author's home directory; here the fallback is $POSE_PROJECT_DIR, and with neither set the assertion
    below fires with the same message the reference gives on a machine without that directory."""
    pose_project_dir = _dj_config().get("custom", {}).get("pose_project_dir", os.environ.get("POSE_PROJECT_DIR", ""))
    assert pose_project_dir and pathlib.Path(pose_project_dir).is_dir(), \
        f"Could not find pose project directory: {pose_project_dir!r}"
    return pose_project_dir


def _to_path(path) -> pathlib.Path:
    return pathlib.Path(str(path).replace("\\", "/"))


def find_full_path(root_directories, relative_path) -> pathlib.Path:
    """First existing `<root>/<relative_path>` over the given roots (one root or a list, in order); a relative_path that
    exists as given wins.  FileNotFoundError when none does."""
    rel = _to_path(relative_path)
    if rel.exists():
        return rel
    roots = [root_directories] if isinstance(root_directories, (str, pathlib.Path)) else list(root_directories)
    for root in roots:
        cand = _to_path(root) / rel
        if cand.exists():
            return cand
    raise FileNotFoundError(f"No valid full-path found (from {roots}) for {rel}")
