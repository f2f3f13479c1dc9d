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
    qkey = "Queue_Id" if "Queue_Id" in rows[0] else "Stream_Id"
    t0 = min(int(r["Start_Timestamp"]) for r in rows)
    t1 = max(int(r["End_Timestamp"]) for r in rows)
    lo = t0 + (t1 - t0) // 2
    rows = [r for r in rows if int(r["Start_Timestamp"]) >= lo]
    by_q = defaultdict(list)
    names = defaultdict(set)
    for r in rows:
        q = r[qkey]
        by_q[q].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
        names[q].add(r["Kernel_Name"].split("(")[0][-60:])
    det_q = {q for q, ns in names.items() if any(k in n for n in ns for k in ("det_preprocess", "roi_align", "rpn_select", "final_decode"))}
    pose_q = {q for q, ns in names.items() if any(k in n for n in ns for k in ("crop_affine", "flip_merge_decode"))} - det_q
    # lanes: a program's lane streams carry only conv / pool kernels; attach each to the family whose main queue it overlaps most
    fam = {q: ("det" if q in det_q else "pose" if q in pose_q else None) for q in by_q}
    for q in by_q:
        if fam[q] is None:
            d = sum(intersect_len(by_q[q], by_q[x]) for x in det_q)
            p = sum(intersect_len(by_q[q], by_q[x]) for x in pose_q)
            # a lane runs BETWEEN its program's main-stream kernels: use adjacency in time instead when there is no overlap
            fam[q] = "det" if d >= p else "pose"
    det = [iv for q in by_q if fam[q] == "det" for iv in by_q[q]]
    pose = [iv for q in by_q if fam[q] == "pose" for iv in by_q[q]]
    span = (t1 - lo) / 1e6
    u_all = union_len(det + pose) / 1e6
    u_det, u_pose = union_len(det) / 1e6, union_len(pose) / 1e6
    both = intersect_len(det, pose) / 1e6
    print(f"window {span:.1f} ms ({len(rows)} kernels on {len(by_q)} queues: {sum(f == 'det' for f in fam.values())} detector, {sum(f == 'pose' for f in fam.values())} pose)")
    print(f"some kernel executing          {u_all:8.1f} ms  ({100 * u_all / span:.1f} % of the window)")
    print(f"detector kernels executing     {u_det:8.1f} ms")
    print(f"pose / lifting kernels executing {u_pose:6.1f} ms")
    print(f"BOTH families executing at once {both:7.1f} ms  ({100 * both / max(u_pose, 1e-9):.1f} % of the pose stage's time runs beside detector kernels)")
    print(f"serial sum {u_det + u_pose:.1f} ms vs union {u_all:.1f} ms: {100 * (1 - u_all / (u_det + u_pose)):.1f} % saved by co-running")


if __name__ == "__main__":
    main()
