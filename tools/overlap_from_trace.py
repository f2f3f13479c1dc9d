#!/usr/bin/env python
"""Which kernels actually co-run?  Reads a rocprofv3 --kernel-trace CSV of `python bench.py --light ...` and reports, over the second
half of the trace (the timed steps), how long kernels of the DETECTOR's queues and of the POSE stage's queues were executing, and for
how long both were executing at once -- the evidence behind the detector look-ahead (cascade.py overlap_detector) and the lane counts.
usage: python tools/overlap_from_trace.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict


def union_len(iv):
    iv = sorted(iv)
    tot, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


def merged(iv):
    out = []
    for s, e in sorted(iv):
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return out


def intersect_len(a, b):
    a, b = merged(a), merged(b)
    i = j = 0
    tot = 0
    while i < len(a) and j < len(b):
        s, e = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if e > s:
            tot += e - s
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    return tot


def main():
    rows = list(csv.DictReader(open(sys.argv[1], newline="")))
    t0 = min(int(r["Start_Timestamp"]) for r in rows)
    t1 = max(int(r["End_Timestamp"]) for r in rows)
    lo = t0 + (t1 - t0) // 2
    rows = [r for r in rows if int(r["Start_Timestamp"]) >= lo]
    # HIP maps many streams to a few hardware queues and rocprofv3 reports one Thread_Id: the STREAM separates the families.  The
    # detector's context stream carries its pre / post-processing kernels, the pose context's the crop / decode / lifting kernels; a
    # program's lane streams carry only its layer kernels -- HRNet's are the ones with fuse sums (upsample_add), the others the detector's
    DET = ("det_preprocess", "roi_align", "rpn_select", "final_decode", "rpn_compact")
    POSE = ("crop_affine", "flip_merge_decode", "upsample_add", "lifting")
    by_s = defaultdict(list)
    mark = defaultdict(lambda: [0, 0])
    for r in rows:
        q = r["Stream_Id"]
        by_s[q].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
        mark[q][0] += any(k in r["Kernel_Name"] for k in DET)
        mark[q][1] += any(k in r["Kernel_Name"] for k in POSE)
    fam = {q: ("det" if mark[q][0] > 0 else "pose" if mark[q][1] > 0 else "det-lane") for q in by_s}
    det = [x for q in by_s if fam[q] != "pose" for x in by_s[q]]
    pose = [x for q in by_s if fam[q] == "pose" for x in by_s[q]]
    span = (t1 - lo) / 1e6
    u_all = union_len(det + pose) / 1e6
    u_det, u_pose = union_len(det) / 1e6, union_len(pose) / 1e6
    both = intersect_len(det, pose) / 1e6
    print(f"window {span:.1f} ms ({len(rows)} kernels on {len(by_s)} streams: " + ", ".join(f"{q}:{fam[q]}" for q in sorted(by_s)) + ")")
    print(f"some kernel executing            {u_all:8.1f} ms  ({100 * u_all / span:.1f} % of the window)")
    print(f"detector kernels executing       {u_det:8.1f} ms")
    print(f"pose / lifting kernels executing {u_pose:8.1f} ms")
    print(f"BOTH families executing at once  {both:8.1f} ms  ({100 * both / max(u_pose, 1e-9):.1f} % of the pose stage's kernel time runs beside detector kernels)")
    print(f"serial sum {u_det + u_pose:.1f} ms vs union {u_all:.1f} ms: {100 * (1 - u_all / (u_det + u_pose)):.1f} % of the serial time saved by co-running")


if __name__ == "__main__":
    main()
