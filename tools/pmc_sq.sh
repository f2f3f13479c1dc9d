#!/bin/bash
# SQ counter passes for the cascade bench (separate rocprofv3 --pmc runs, --kernel-trace only).  Output: gpurun_out/pmc_sq/summary.txt
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_sq
rm -rf $O; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES --output-format csv -d $O/p1 -- python bench.py --profile-serial --steps 2 --warmup 1 > $O/p1.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/p2 -- python bench.py --profile-serial --steps 2 --warmup 1 > $O/p2.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/p3 -- python bench.py --profile-serial --steps 2 --warmup 1 > $O/p3.log 2>&1
python - <<'PY' > $O/summary.txt 2>&1
import csv, glob, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "gpurun_out/pmc_sq")
KERNEL = os.environ.get("PMC_KERNEL", "conv_split_kernel")
tot = {}
n = {}
for p in ("p1", "p2", "p3"):
    for path in glob.glob(os.path.join(O, p, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path, newline="")):
            if KERNEL not in row["Kernel_Name"]:
                continue
            k = row["Counter_Name"]
            tot[k] = tot.get(k, 0.0) + float(row["Counter_Value"])
            n[k] = n.get(k, 0) + 1
print(f"# SQ counters summed over all {KERNEL} launches of: python bench.py --profile-serial --steps 2 --warmup 1 (cascade, 64 frames/step, every launch on one stream)")
for k in sorted(tot):
    print(f"{k:32s} {tot[k]:.4e}  over {n[k]} launches")
g = tot.get
if g("SQ_VALU_MFMA_BUSY_CYCLES") and g("SQ_BUSY_CU_CYCLES"):
    print("MFMA busy / CU busy cycles = %.3f (4 SIMDs per CU: %.1f %% of the matrix pipes' time)" % (g("SQ_VALU_MFMA_BUSY_CYCLES") / g("SQ_BUSY_CU_CYCLES"), 25 * g("SQ_VALU_MFMA_BUSY_CYCLES") / g("SQ_BUSY_CU_CYCLES")))
if g("SQ_INSTS_VALU") and g("SQ_INSTS_MFMA"):
    print("VALU instructions per MFMA instruction (incl. prologue / epilogue) = %.2f" % ((g("SQ_INSTS_VALU") - g("SQ_INSTS_MFMA", 0)) / g("SQ_INSTS_MFMA")))
if g("SQ_LDS_IDX_ACTIVE"):
    print("LDS bank-conflict cycles / LDS active cycles = %.4f" % (g("SQ_LDS_BANK_CONFLICT", 0.0) / g("SQ_LDS_IDX_ACTIVE")))
PY
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
cat $O/summary.txt
