#!/usr/bin/env python
"""Record a measured HBM-traffic figure in profiles/pmc_traffic.json together with the fingerprint of the kernel sources it
was measured on (bench.py reports the figure only while the fingerprint matches: bench.kernel_source_sha).

  python tools/pmc_traffic_update.py <key> <pmc_summary.txt containing TRAFFIC_BYTES_PER_LAUNCH=...> <source note> [out.json]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    key, summary, note = sys.argv[1:4]
    out = sys.argv[4] if len(sys.argv) > 4 else os.path.join(ROOT, "profiles", "pmc_traffic.json")
    m = re.search(r"TRAFFIC_BYTES_PER_LAUNCH=(\d+)", open(summary).read())
    if not m:
        sys.exit(f"{summary}: no TRAFFIC_BYTES_PER_LAUNCH line")
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    data = json.load(open(path)) if os.path.exists(path) else {}
    data[key] = {"traffic_bytes_per_launch": int(m.group(1)), "kernel_sha": bench.kernel_source_sha(), "source": note}
    json.dump(data, open(out, "w"), indent=1)
    print(out, key, data[key])


if __name__ == "__main__":
    main()
