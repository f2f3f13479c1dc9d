"""mmtrack SortTracker with its ReID branch (mot/deepsort/deepsort_faster-rcnn_fpn_4e_mot17-private-half.py:43-54): the product
tracker (posepipeline_amd.tracking.SortReidTracker: C++ Kalman filter + Hungarian solver behind the C ABI) against the
independent numpy / scipy oracle (oracle/reid_mm.py) on seeded multi-person sequences with synthetic appearance
embeddings -- ids bit-exact.  Host code only: runs without a GPU.  Both sides are restatements of un-vendored mmtrack 0.x
(parity unpinned); what this pins is that the C++ primitives and the bookkeeping give the same ids as plain numpy."""
import numpy as np
import pytest

from oracle import reid_mm as orm
from posepipeline_amd.models import reid_r50
from posepipeline_amd.tracking import SortReidTracker


def scene(rng, n_frames, n_persons, miss_p=0.12, swap_at=None):
    """persons with distinct appearance directions walking / crossing; occasional misses and low-score detections"""
    app = rng.standard_normal((n_persons, 128)).astype(np.float32)
    app *= np.float32(1.4) / np.linalg.norm(app, axis=1, keepdims=True).astype(np.float32)
    pos = rng.uniform(50, 900, (n_persons, 2))
    vel = rng.uniform(-9, 9, (n_persons, 2))
    frames = []
    for t in range(n_frames):
        dets, emb = [], []
        for p in range(n_persons):
            if rng.random() < miss_p or (t > 20 and p == 0 and 25 <= t < 32):       # person 0 disappears for 7 frames: ReID brings the id back
                continue
            x, y = pos[p] + vel[p] * t
            w, h = 60 + 8 * p, 160 + 15 * p
            score = 0.97 - 0.03 * p if rng.random() > 0.1 else 0.42                 # sometimes below obj_score_thr
            dets.append([x, y, x + w, y + h, score])
            emb.append(app[p] + rng.normal(0, 0.05, 128).astype(np.float32))
        if rng.random() < 0.3:                                                      # a false positive with a random look
            x, y = rng.uniform(0, 1000, 2)
            dets.append([x, y, x + 50, y + 120, 0.7])
            emb.append(rng.standard_normal(128).astype(np.float32))
        frames.append((np.array(dets, np.float32).reshape(-1, 5), np.array(emb, np.float32).reshape(-1, 128)))
    return frames


@pytest.mark.parametrize("seed,n_persons", [(0, 3), (1, 5), (2, 8), (3, 2)])
def test_ids_equal_oracle(seed, n_persons):
    rng = np.random.default_rng(seed)
    frames = scene(rng, 60, n_persons)
    trk, ref = SortReidTracker(), orm.SortReidTrackerRef()
    seen_back = False
    ids_of_first = None
    for t, (dets, emb) in enumerate(frames):
        keep = trk.keep(dets)
        assert np.array_equal(keep, ref.keep(dets))
        got = trk.step(dets[keep], emb[keep])
        want = ref.step(dets[keep], emb[keep], t)
        assert np.array_equal(got, want), (t, got[:, 0], want[:, 0])
        assert set(trk.tracks) == set(ref.tracks)
    assert trk.num_tracks == ref.num_tracks >= n_persons


def test_reid_recovers_identity_after_a_gap_and_gating_blocks_far_matches():
    """a person unseen for 7 frames keeps its id through the appearance stage (SORT without ReID would issue a new one);
    an identical-looking detection far outside the Kalman gate does not steal it"""
    e = np.zeros((1, 128), np.float32)
    e[0, 3] = 1.0
    trk = SortReidTracker()
    box = lambda x: np.array([[x, 100, x + 60, 280, 0.9]], np.float32)
    ids = [int(trk.step(box(100 + 4 * t), e)[0, 0]) for t in range(5)]
    assert ids == [0] * 5 and not trk.tracks[0]["tentative"]
    for _ in range(7):
        assert trk.step(np.zeros((0, 5), np.float32), np.zeros((0, 128), np.float32)).shape == (0, 6)
    assert int(trk.step(box(100 + 4 * 12), e)[0, 0]) == 0                      # re-identified
    far = trk.step(np.array([[900, 600, 960, 780, 0.9]], np.float32), e)      # same look, gated by the motion model
    assert int(far[0, 0]) == 1


def test_crop_rects_match_oracle():
    rng = np.random.default_rng(5)
    boxes = np.concatenate([rng.uniform(-50, 1900, (40, 2)), rng.uniform(-50, 1100, (40, 2))], 1)[:, [0, 2, 1, 3]].astype(np.float32)
    boxes[:, 2:] = boxes[:, :2] + rng.uniform(0, 300, (40, 2)).astype(np.float32)
    boxes[0] = [10.2, 10.2, 10.4, 300]                                         # empty after truncation -> widened by one pixel
    boxes[1] = [2000, 1200, 2100, 1300]                                        # outside: clamped to the border, then widened
    sf = np.array([1088 / 1920, 612 / 1080, 1088 / 1920, 612 / 1080], np.float32)
    got = reid_r50.crop_rects(boxes, sf, (612, 1088))
    want = orm.crop_rects(boxes, sf, (612, 1088))
    assert np.array_equal(got, want)
    assert (got[:, 2] > got[:, 0]).all() and (got[:, 3] > got[:, 1]).all()
