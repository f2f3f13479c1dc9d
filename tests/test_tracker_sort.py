"""Product tracker mode 1 (mmtrack SortTracker without ReID, C++) vs oracle/tracking.py (numpy + scipy's
own Hungarian) on synthetic multi-person sequences: ids bit-exact.  Host code, runs without a GPU."""
import numpy as np
import pytest

from oracle.tracking import SortTrackerRef
from posepipeline_amd.tracking import Tracker


def sequence(rng, n_frames, n_people, drop=0.1):
    people = [dict(x=rng.uniform(0, 1500), y=rng.uniform(0, 500), w=rng.uniform(60, 160), h=rng.uniform(200, 500),
                   vx=rng.uniform(-12, 12), vy=rng.uniform(-3, 3)) for _ in range(n_people)]
    for t in range(n_frames):
        rows = []
        for p in people:
            if rng.uniform() < drop:
                continue
            x, y = p["x"] + p["vx"] * t + rng.uniform(-3, 3), p["y"] + p["vy"] * t + rng.uniform(-3, 3)
            rows.append([x, y, x + p["w"], y + p["h"], rng.uniform(0.3, 1.0)])
        if rng.uniform() < 0.1:
            rows.append([rng.uniform(0, 1700), rng.uniform(0, 800), 0, 0, rng.uniform(0.5, 0.8)])
            rows[-1][2] = rows[-1][0] + rng.uniform(40, 150)
            rows[-1][3] = rows[-1][1] + rng.uniform(100, 300)
        rng.shuffle(rows)
        yield np.array(rows, np.float32).reshape(-1, 5)


@pytest.mark.parametrize("seed,n_people", [(0, 1), (1, 4), (2, 8), (3, 15)])
def test_sort_tracker_ids_bit_exact(seed, n_people):
    rng = np.random.default_rng(seed)
    ref = SortTrackerRef()
    trk = Tracker(mode=1, match_iou_thr=0.5, obj_score_thr=0.5)
    seen = set()
    for t, dets in enumerate(sequence(rng, 120, n_people)):
        if t in (40, 41):
            dets = dets[:0]                      # nobody detected: every track is lost, ids restart fresh
        want = ref.step(dets)
        ids, boxes, info = trk.step(dets[:, :4].astype(np.float64), dets[:, 4].astype(np.float64))
        assert np.array_equal(ids, want[:, 0].astype(np.int64)), f"frame {t}"
        assert np.array_equal(boxes.astype(np.float32), want[:, 1:5])
        assert np.array_equal(dets[info[:, 1], 4], want[:, 5])
        seen.update(ids.tolist())
    assert len(seen) >= n_people
