#!/bin/bash
# Effective shader clock under the conv kernels: GRBM_GUI_ACTIVE (cycles the GPU was busy during a dispatch) divided by the
# dispatch's duration from the same rocprofv3 counter-collection rows (MI355X_MICROARCH.md "DVFS give-back": the chip clocks
# to its power budget, so the 157.3 TFLOP/s fp32-MFMA peak -- quoted at 2.4 GHz -- is not what a loaded chip can reach).
#   bash tools/pmc_clock.sh <label> <command...>      -> gpurun_out/pmc_clock_<label>/summary.txt
LABEL=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_clock_$LABEL
rm -rf $O; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/p1 -- "$@" > $O/p1.log 2>&1
PMC_DIR="$O" PMC_CMD="$*" python - <<'PY' > $O/summary.txt 2>&1
import csv, glob, os
O = os.environ["PMC_DIR"]
rows = []
for path in glob.glob(os.path.join(O, "p1", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path, newline="")):
        if row["Counter_Name"] == "GRBM_GUI_ACTIVE":
            rows.append(row)
print(f"# effective clock = GRBM_GUI_ACTIVE / dispatch duration, per kernel family, in: {os.environ['PMC_CMD']}")
if rows and "Start_Timestamp" not in rows[0]:
    print("counter_collection.csv has no timestamps; columns:", list(rows[0]))
fam = {}
for r in rows:
    try:
        dt = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    except (KeyError, ValueError):
        continue
    if dt <= 0:
        continue
    name = r["Kernel_Name"].split("(")[0][-60:]
    f = fam.setdefault(name, [0.0, 0.0, 0])
    f[0] += float(r["Counter_Value"]); f[1] += dt; f[2] += 1
for name, (cyc, ns, n) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"{name:62s} {n:6d} launches  {ns / 1e6:10.2f} ms  {cyc / ns:6.3f} GHz")
conv = [v for k, v in fam.items() if "conv_igemm" in k]
if conv:
    cyc, ns = sum(v[0] for v in conv), sum(v[1] for v in conv)
    print(f"all conv_igemm_kernel launches: {cyc / ns:.3f} GHz effective -> fp32 MFMA ceiling at that clock {157.3 * cyc / ns / 2.4:.1f} TFLOP/s")
PY
find $O -name "*.csv" -delete
cat $O/summary.txt
