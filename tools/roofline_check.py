#!/usr/bin/env python
"""Recompute bench.py's `roofline` from a rocprofv3 --kernel-trace --stats summary of the SAME run.

usage: python tools/roofline_check.py <kernel_stats.csv> <bench line .json>
The run must be `python bench.py --profile-serial ...` (every launch on one stream, no detector look-ahead): per-kernel
durations are additive only then.  For the kernel family the line names (`conv_split_*` or the float32 MFMA kernels):
launches, total and average duration from the CSV; average x launches_per_step against the line's ms_per_step_serial
(HIP events inside bench.py); the roofline fraction recomputed from the CSV."""
import csv
import json
import sys


def main():
    stats, line = sys.argv[1], sys.argv[2]
    with open(line) as f:
        txt = f.read()
    out = json.loads(txt[txt.index("{"):].splitlines()[0])
    roof = out["roofline"]
    fams = {"conv_split": ("conv_split_kernel", "conv_split_gemm_kernel", "conv_split_s2_kernel", "conv_split48_kernel"),
            "fp32": ("conv_igemm_kernel", "conv_p3_kernel")}
    rows = list(csv.DictReader(open(stats, newline="")))
    tot = {}
    for r in rows:
        name = r["Name"]
        for fam, pats in fams.items():
            if any(name.startswith(p) or (" " + p) in name or ("::" + p) in name or p + "<" in name for p in pats):
                t = tot.setdefault(fam, [0, 0.0])
                t[0] += int(r["Calls"])
                t[1] += float(r["TotalDurationNs"])
    all_ns = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"# {stats}  vs  {line}")
    print(f"all kernels: {all_ns / 1e6:.1f} ms in the profiled process")
    for fam, key in (("conv_split", None), ("fp32", "fp32_mfma_kernels")):
        if fam not in tot:
            continue
        calls, ns = tot[fam]
        r = roof if key is None else roof.get(key)
        if not r:
            continue
        avg_us = ns / calls / 1e3
        pred = avg_us * r["launches_per_step"] / 1e3
        frac = r["flops_per_launch"] / (avg_us * 1e-6) / 1e12 / r["peak"]
        print(f"{fam}: {calls} launches, {ns / 1e6:.1f} ms, AverageNs {avg_us:.1f} us ({100 * ns / all_ns:.1f} % of kernel time)")
        print(f"  rocprof: average x {r['launches_per_step']} launches per step = {pred:.2f} ms; bench line (HIP events): "
              f"ms_per_step_serial {r['ms_per_step_serial']:.2f} ms, avg_launch_ms {r['avg_launch_ms'] * 1e3:.1f} us -> {100 * (pred / r['ms_per_step_serial'] - 1):+.1f} %")
        print(f"  roofline frac from the CSV: {r['flops_per_launch'] / 1e9:.2f} GFLOP / {avg_us:.1f} us = {frac * r['peak']:.1f} TFLOP/s "
              f"/ {r['peak']:.1f} = {frac:.3f}   (line: {r['frac']:.3f})")


if __name__ == "__main__":
    main()
