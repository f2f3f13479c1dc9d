#!/usr/bin/env python
"""Summarise rocprofv3 --pmc passes (counter_collection CSVs) for the conv kernel.

usage: python tools/pmc_summary.py <fetch_dir> <write_dir> <launches_per_step> <label> [kernel name substring = conv_split_kernel]
Each dir is the -d output of ONE separate pass
    rocprofv3 --kernel-trace --pmc FETCH_SIZE  --output-format csv -d <fetch_dir> -- python bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE  --output-format csv -d <write_dir> -- python bench.py ...
FETCH_SIZE / WRITE_SIZE are reported in KiB; FETCH_SIZE is doubled (gfx950 counts the 128-byte requests of
16-byte-per-lane streams as 64 B -- MI355X_MICROARCH.md, HBM section).  Prints the per-launch HBM-side traffic of
the named kernel (default: the dominant one, conv_split_kernel) and the line to paste into profiles/pmc_traffic.json."""
import csv
import glob
import os
import sys


KERNEL = "conv_split_kernel"


def collect(d, counter):
    tot, n, other = 0.0, 0, {}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] != counter:
                    continue
                v = float(row["Counter_Value"])
                name = row["Kernel_Name"]
                if KERNEL in name:
                    tot += v
                    n += 1
                else:
                    key = name.split("(")[0].split("<")[0][-40:]
                    other[key] = other.get(key, 0.0) + v
    return tot, n, other


def main():
    global KERNEL
    fetch_dir, write_dir, per_step, label = sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4]
    if len(sys.argv) > 5:
        KERNEL = sys.argv[5]
    f_kib, f_n, f_other = collect(fetch_dir, "FETCH_SIZE")
    w_kib, w_n, w_other = collect(write_dir, "WRITE_SIZE")
    print(f"# {label}: {KERNEL}, {per_step:g} launches per step (the profiled run also contains warm-up, serial and streamed steps)")
    print(f"FETCH_SIZE {KERNEL} total {f_kib / 1024:.1f} MiB over {f_n} launches; other kernels (MiB): "
          f"{ {k: round(v / 1024, 1) for k, v in sorted(f_other.items(), key=lambda kv: -kv[1])[:6]} }")
    print(f"WRITE_SIZE {KERNEL} total {w_kib / 1024:.1f} MiB over {w_n} launches; other kernels (MiB): "
          f"{ {k: round(v / 1024, 1) for k, v in sorted(w_other.items(), key=lambda kv: -kv[1])[:6]} }")
    if f_n == 0 or w_n == 0:
        print("no conv launches found")
        return
    fetch_b = 2.0 * f_kib * 1024.0
    write_b = w_kib * 1024.0
    per_launch = fetch_b / f_n + write_b / w_n
    print(f"corrected conv traffic per launch: fetch {fetch_b / f_n / 1e6:.1f} MB + write {write_b / w_n / 1e6:.1f} MB = "
          f"{per_launch / 1e6:.1f} MB; per step {per_launch * per_step / 1e9:.2f} GB")
    print(f"TRAFFIC_BYTES_PER_LAUNCH={int(per_launch)}")


if __name__ == "__main__":
    main()
