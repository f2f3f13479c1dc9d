"""Recipes over the hot-path tables: video key -> tracking -> person box -> top-down 2D -> 3D lifting.

Same entry points, arguments and return values as the reference's recipes
(pose_pipeline/utils/standard_pipelines.py:10 `tracking_pipeline`, :56 `top_down_pipeline`, :110 `lifting_pipeline`;
pose_pipeline/utils/tracking.py:5 `annotate_single_person`), so `scripts/process_h36m.py`-style callers run unchanged.
The bodies are this package's own: every recipe is a walk over the declarative `STAGES` table below (lookup table ->
method table -> computed tables), so adding a stage or a method is a table entry, not another copy of the sequence.

The DEFAULTS are the reference's own (utils/standard_pipelines.py:12,58-59,112-114), with the reference's behaviour:
  * tracking_method_name "DeepSortYOLOv4" -- tracking_method 0, built here (wrappers/deep_sort_yolov4/);
  * top_down_method_name "MMpose" -- NOT a row of TopDownMethodLookup (the row is spelled "MMPose"), so a call that relies on
    the default fails in the lookup's fetch1, exactly like the reference; callers pass "MMPose" / "MMPoseHalpe" / ...
    (scripts/process_h36m.py does);
  * lifting_method_name "GastNet" -- a lookup row whose wrapper is outside the hot path:
    LiftingPerson.make raises for it; callers of the hot path pass "VideoPose3D".
A method name that is not in its lookup table raises from fetch1 (no silent substitution).  `BestDetectedFrames` is populated
where the reference populates it (:100, :162); the OpenPose branch (:95-97, SURVEY.md section 2, out of scope) is not part of
the walk.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Union

import numpy as np

from .. import pipeline as P


@dataclass(frozen=True)
class Stage:
    """A method-selectable stage: `<column>_name` in `lookup` names the method, `method` holds (parent key, method id),
    `computed` are the tables that a populate() on that key fills."""
    column: str
    lookup: type
    method: type
    computed: tuple

    def method_id(self, name: str):
        return (self.lookup & f'{self.column}_name="{name}"').fetch1(self.column)

    def enter(self, parent_key: dict, name: str) -> dict:
        """parent key + this stage's method id, registered in the method table (idempotent)"""
        key = {**parent_key, self.column: self.method_id(name)}
        self.method.insert1(key, skip_duplicates=True)
        return key

    def run(self, key: dict, reserve_jobs: bool):
        for table in self.computed:
            table.populate(key, reserve_jobs=reserve_jobs)


STAGES = {
    "tracking": Stage("tracking_method", P.TrackingBboxMethodLookup, P.TrackingBboxMethod, (P.TrackingBbox,)),
    "top_down": Stage("top_down_method", P.TopDownMethodLookup, P.TopDownMethod, (P.TopDownPerson,)),
    "lifting": Stage("lifting_method", P.LiftingMethodLookup, P.LiftingMethod, (P.LiftingPerson,)),
}


def annotate_single_person(filt, subject_id=0, confirm=False):
    """Videos in which the tracker found exactly one identity get that identity as their subject of interest
    (PersonBboxValid row), unless they already have one.  confirm=True asks on stdin first."""
    todo = ((P.TrackingBbox & filt & "num_tracks=1") - P.PersonBboxValid).fetch("KEY")
    if confirm:
        answer = input(f"{len(todo)} single-person video(s) can be annotated automatically -- proceed? [y/N] ")
        if not answer.strip().lower().startswith("y"):
            print("nothing annotated")
            return
    for key in todo:
        per_frame = (P.TrackingBbox & key).fetch1("tracks")
        ids = sorted({row["track_id"] for frame in per_frame for row in frame})
        assert len(ids) == 1, f"num_tracks = 1 but the track list holds ids {ids}"
        P.PersonBboxValid.insert1({**key, "video_subject_id": subject_id, "keep_tracks": np.asarray(ids)})   # stored as an array, like np.unique's


def _as_list(keys):
    return [keys] if isinstance(keys, dict) else list(keys)


def _person_key(tracking_key: dict):
    """the PersonBbox key under a tracking key, or None while no (single) subject has been annotated / computed"""
    P.PersonBbox.populate(tracking_key, reserve_jobs=True)
    rows = P.PersonBbox & tracking_key
    return rows.fetch1("KEY") if len(rows) == 1 else None


def tracking_pipeline(keys: Union[Dict, List[Dict]], tracking_method_name: str = "DeepSortYOLOv4", reserve_jobs: bool = False):
    """Video key(s) -> tracked boxes -> (auto-annotated) subject -> smoothed person box.
    Returns the PersonBbox keys of the videos that have exactly one."""
    done = []
    for video_key in _as_list(keys):
        P.VideoInfo.populate(video_key, reserve_jobs=reserve_jobs)
        tracking_key = STAGES["tracking"].enter(video_key, tracking_method_name)
        STAGES["tracking"].run(tracking_key, reserve_jobs)
        annotate_single_person(video_key)
        person = _person_key(tracking_key)
        P.DetectedFrames.populate(tracking_key, reserve_jobs=reserve_jobs)
        if person is not None:
            done.append(person)
    return done


def top_down_pipeline(key: Union[Dict, List[Dict]], tracking_method_name: str = "DeepSortYOLOv4",
                      top_down_method_name: str = "MMpose", reserve_jobs: bool = False):
    """... -> 2D key points of the subject.  Returns the TopDownPerson keys; False as soon as a video has no person box
    (its subject is not annotated yet, or was marked invalid with a negative video_subject_id)."""
    out = []
    for person in tracking_pipeline(key, tracking_method_name, reserve_jobs=reserve_jobs):
        if _person_key(person) is None:
            marked = P.PersonBboxValid & person
            if len(marked) == 1 and marked.fetch1("video_subject_id") < 0:
                print(f"{key}: marked invalid, skipped")
            else:
                print(f"{person}: no subject of interest annotated yet")
            return False
        top_down_key = STAGES["top_down"].enter(person, top_down_method_name)
        STAGES["top_down"].run(top_down_key, reserve_jobs)
        P.BestDetectedFrames.populate(key, reserve_jobs=reserve_jobs)
        out.append(top_down_key)
    return out


def lifting_pipeline(key, tracking_method_name: str = "DeepSortYOLOv4", top_down_method_name: str = "MMpose",
                     lifting_method_name: str = "GastNet", reserve_jobs: bool = False):
    """... -> 3D joints.  Returns True when the video has a LiftingPerson row afterwards; the falsy result of
    top_down_pipeline, or False, when an upstream row is missing (another worker holds the job)."""
    upstream = top_down_pipeline(key, tracking_method_name, top_down_method_name, reserve_jobs=reserve_jobs)
    if not upstream:
        return upstream
    person = (P.PersonBbox & {**key, "tracking_method": STAGES["tracking"].method_id(tracking_method_name)}).fetch1("KEY")
    top_down_key = {**person, "top_down_method": STAGES["top_down"].method_id(top_down_method_name)}
    if len(P.TopDownPerson & top_down_key) == 0:
        print(f"{top_down_key}: 2D key points not available (job reserved elsewhere?)")
        return False
    lifting_key = STAGES["lifting"].enter(top_down_key, lifting_method_name)
    STAGES["lifting"].run(key, reserve_jobs)
    if len(P.LiftingPerson & lifting_key) == 0:
        print(f"{lifting_key}: 3D joints not available (job reserved elsewhere?)")
        return False
    for table in (P.VideoInfo, P.DetectedFrames, P.BestDetectedFrames):
        table.populate(key, reserve_jobs=reserve_jobs)
    return len(P.LiftingPerson & key) > 0
