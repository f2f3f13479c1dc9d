#!/bin/bash
# One HBM-bound kernel of an arbitrary command: average launch duration (rocprofv3 --kernel-trace --stats) and HBM-side traffic
# (FETCH_SIZE / WRITE_SIZE in their own --pmc passes, FETCH x2 as MI355X_MICROARCH.md prescribes for gfx950) -> GB/s.
#   bash tools/hbm_kernel_profile.sh <kernel-name-substring> <label> <algorithmic bytes per launch> <command...>
# Output: gpurun_out/hbm_<label>.txt
PAT=$1; LABEL=$2; ALGO=$3; shift 3
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/hbm_$LABEL.d
rm -rf $O; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- "$@" > $O/stats.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- "$@" > $O/fetch.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- "$@" > $O/write.log 2>&1
PMC_PAT="$PAT" PMC_DIR="$O" PMC_CMD="$*" PMC_ALGO="$ALGO" python - <<'PY' > $R/gpurun_out/hbm_$LABEL.txt 2>&1
import csv, glob, os
O, pat, algo = os.environ["PMC_DIR"], os.environ["PMC_PAT"], float(os.environ["PMC_ALGO"])
print(f"# kernel matching '{pat}' in: {os.environ['PMC_CMD']}")
dur = []
for path in glob.glob(os.path.join(O, "stats", "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(path, newline="")):
        if pat in row["Kernel_Name"]:
            dur.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"]), row["Kernel_Name"][:60], row.get("Grid_Size_X", "")))
cnt = {}
for name in ("fetch", "write"):
    for path in glob.glob(os.path.join(O, name, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path, newline="")):
            if pat in row["Kernel_Name"]:
                cnt.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
if not dur:
    print("no launch found")
    raise SystemExit
# launches are grouped by grid size (a command may run the kernel at several shapes); the LARGEST group's first shape is reported per group
groups = {}
for d, n, g in dur:
    groups.setdefault(g, []).append(d)
for g, ds in groups.items():
    ds = sorted(ds)
    print(f"grid {g}: {len(ds)} launches, duration avg {sum(ds) / len(ds) / 1e3:.1f} us  median {ds[len(ds) // 2] / 1e3:.1f} us  min {ds[0] / 1e3:.1f} us")
g0 = max(groups, key=lambda k: len(groups[k]))
ds = sorted(groups[g0])
avg = sum(ds) / len(ds) * 1e-9
med = ds[len(ds) // 2] * 1e-9
f = cnt.get("FETCH_SIZE", [])
w = cnt.get("WRITE_SIZE", [])
print(f"kernel: {dur[0][1]}")
if f and w:
    fb, wb = 2 * 1024 * sum(f) / len(f), 1024 * sum(w) / len(w)
    print(f"HBM-side traffic per launch (all {len(f)} launches averaged): fetch {fb / 1e6:.2f} MB (x2 gfx950 correction) + write {wb / 1e6:.2f} MB = {(fb + wb) / 1e6:.2f} MB")
    print(f"counter traffic / median duration = {(fb + wb) / med / 1e12:.3f} TB/s")
print(f"algorithmic bytes per launch {algo / 1e6:.2f} MB / median duration {med * 1e6:.1f} us = {algo / med / 1e12:.3f} TB/s = {algo / med / 8e12:.3f} of the 8 TB/s HBM3E peak "
      f"({algo / med / 6.29e12:.3f} of the 6.29 TB/s a float4 copy sustains); on the average duration {avg * 1e6:.1f} us: {algo / avg / 1e12:.3f} TB/s")
PY
rm -rf $O
cat $R/gpurun_out/hbm_$LABEL.txt
