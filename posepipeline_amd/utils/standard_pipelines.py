"""Recipes that chain the hot-path tables, with the reference's names and arguments
(pose_pipeline/utils/standard_pipelines.py:10-164 `tracking_pipeline`, `top_down_pipeline`,
`lifting_pipeline`; pose_pipeline/utils/tracking.py:5-21 `annotate_single_person`).

Differences kept deliberately small: the default tracking method stays "MMTrack_deepsort" (the reference's default
"DeepSortYOLOv4" is built too -- pass tracking_method_name="DeepSortYOLOv4"), the default lifter is "VideoPose3D"
(instead of "GastNet"), the reference's "MMpose" default for the 2D method -- a name that is not in its own
lookup table -- is "MMPose", and `BestDetectedFrames` / OpenPose branches (out of scope) are not called.
"""
from __future__ import annotations

from typing import Dict, List, Union

import numpy as np

from ..pipeline import (DetectedFrames, LiftingMethod, LiftingMethodLookup, LiftingPerson, PersonBbox, PersonBboxValid,
                        TopDownMethod, TopDownMethodLookup, TopDownPerson, TrackingBbox, TrackingBboxMethod,
                        TrackingBboxMethodLookup, VideoInfo)


def annotate_single_person(filt, subject_id=0, confirm=False):
    """utils/tracking.py:5-21: videos whose tracker found exactly one identity get that identity as the subject"""
    keys = ((TrackingBbox & filt & "num_tracks=1") - PersonBboxValid).fetch("KEY")
    for k in keys:
        tracks = (TrackingBbox & k).fetch1("tracks")
        track_id = np.unique([[t["track_id"] for t in t2] for t2 in tracks if len(t2) > 0])
        assert len(track_id) == 1, "Found two tracks, should not have"
        k.update({"video_subject_id": subject_id, "keep_tracks": track_id})
        PersonBboxValid.insert1(k)


def tracking_pipeline(keys: Union[Dict, List[Dict]], tracking_method_name: str = "MMTrack_deepsort", reserve_jobs: bool = False):
    """Run the pipeline on a video through to the tracking layer; returns the PersonBbox keys that resulted."""
    if isinstance(keys, dict):
        keys = [keys]
    tracking_keys = []
    for key in keys:
        VideoInfo.populate(key, reserve_jobs=reserve_jobs)
        tracking_key = key.copy()
        tracking_method = (TrackingBboxMethodLookup & f'tracking_method_name="{tracking_method_name}"').fetch1("tracking_method")
        tracking_key["tracking_method"] = tracking_method
        TrackingBboxMethod.insert1(tracking_key, skip_duplicates=True)
        TrackingBbox.populate(tracking_key, reserve_jobs=reserve_jobs)
        annotate_single_person(key)                         # auto-annotate single-identity videos
        PersonBbox.populate(tracking_key, reserve_jobs=True)
        DetectedFrames.populate(tracking_key, reserve_jobs=reserve_jobs)
        if len(PersonBbox & tracking_key) == 1:
            tracking_keys.append((PersonBbox & tracking_key).fetch1("KEY"))
    return tracking_keys


def top_down_pipeline(key: Union[Dict, List[Dict]], tracking_method_name: str = "MMTrack_deepsort",
                      top_down_method_name: str = "MMPose", reserve_jobs: bool = False):
    """... through to the top-down person layer; returns the TopDownPerson keys (False while annotation is pending)."""
    tracking_keys = tracking_pipeline(key, tracking_method_name, reserve_jobs=reserve_jobs)
    top_down_person_keys = []
    for tracking_key in tracking_keys:
        PersonBbox.populate(tracking_key, reserve_jobs=True)
        if len(PersonBbox & tracking_key) == 0:
            if len(PersonBboxValid & tracking_key) == 1 and (PersonBboxValid & tracking_key).fetch1("video_subject_id") < 0:
                print(f"Video {key} marked as invalid.")
                return False
            print(f"Waiting for annotation of subject of interest. {tracking_key}")
            return False
        top_down_key = (PersonBbox & tracking_key).fetch1("KEY")
        top_down_method = (TopDownMethodLookup & f'top_down_method_name="{top_down_method_name}"').fetch1("top_down_method")
        top_down_key["top_down_method"] = top_down_method
        TopDownMethod.insert1(top_down_key, skip_duplicates=True)
        TopDownPerson.populate(top_down_key, reserve_jobs=reserve_jobs)
        top_down_person_keys.append(top_down_key)
    return top_down_person_keys


def lifting_pipeline(key, tracking_method_name: str = "MMTrack_deepsort", top_down_method_name: str = "MMPose",
                     lifting_method_name: str = "VideoPose3D", reserve_jobs: bool = False):
    """... through to the lifting layer; returns whether a LiftingPerson row now exists for the video."""
    res = top_down_pipeline(key, tracking_method_name, top_down_method_name, reserve_jobs=reserve_jobs)
    if not res:
        return res
    tracking_key = key.copy()
    tracking_key["tracking_method"] = (TrackingBboxMethodLookup & f'tracking_method_name="{tracking_method_name}"').fetch1("tracking_method")
    top_down_key = (PersonBbox & tracking_key).fetch1("KEY")
    top_down_key["top_down_method"] = (TopDownMethodLookup & f'top_down_method_name="{top_down_method_name}"').fetch1("top_down_method")
    if len(TopDownPerson & top_down_key) == 0:
        print(f"Top down job must be reserved and not completed. {top_down_key}")
        return False
    lifting_key = top_down_key.copy()
    lifting_key["lifting_method"] = (LiftingMethodLookup & f'lifting_method_name="{lifting_method_name}"').fetch1("lifting_method")
    LiftingMethod.insert1(lifting_key, skip_duplicates=True)
    LiftingPerson.populate(key, reserve_jobs=reserve_jobs)
    if len(LiftingPerson & lifting_key) == 0:
        print(f"Lifting job must be reserved and not completed. {lifting_key}")
        return False
    VideoInfo.populate(key, reserve_jobs=reserve_jobs)
    DetectedFrames.populate(key, reserve_jobs=reserve_jobs)
    return len(LiftingPerson & key) > 0
