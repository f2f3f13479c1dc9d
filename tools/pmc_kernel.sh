#!/bin/bash
# SQ / TCC counter passes for ONE kernel family of an arbitrary command (separate rocprofv3 --pmc runs, --kernel-trace only).
#   bash tools/pmc_kernel.sh <kernel-name-substring> <label> <command...>
# PMC_FULL=1 adds the TA / TCP / TCC passes (slow: minutes per pass).  Output: gpurun_out/pmc_<label>/summary.txt
PAT=$1; LABEL=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_$LABEL
rm -rf $O; mkdir -p $O
cd $R
export TMPDIR=/tmp
SETS=("SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAVES"
      "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS"
      "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
      "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
      "FETCH_SIZE" "WRITE_SIZE")
if [ -n "$PMC_FULL" ]; then
    SETS+=("TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum"
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_BUSY_avr")
fi
i=0
for SET in "${SETS[@]}"; do
    i=$((i + 1))
    timeout 900 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $O/p$i -- "$@" > $O/p$i.log 2>&1
done
PMC_PAT="$PAT" PMC_DIR="$O" PMC_CMD="$*" python - <<'PY' > $O/summary.txt 2>&1
import csv, glob, os
O, pat = os.environ["PMC_DIR"], os.environ["PMC_PAT"]
tot, n = {}, {}
for path in glob.glob(os.path.join(O, "p*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path, newline="")):
        if pat not in row["Kernel_Name"]:
            continue
        k = row["Counter_Name"]
        tot[k] = tot.get(k, 0.0) + float(row["Counter_Value"])
        n[k] = n.get(k, 0) + 1
print(f"# SQ / TCC counters summed over all launches of kernels matching '{pat}' in: {os.environ['PMC_CMD']}")
for k in sorted(tot):
    print(f"{k:32s} {tot[k]:.4e}  over {n[k]} launches")
g = tot.get
if g("SQ_VALU_MFMA_BUSY_CYCLES") and g("SQ_BUSY_CU_CYCLES"):
    r = g("SQ_VALU_MFMA_BUSY_CYCLES") / g("SQ_BUSY_CU_CYCLES")
    print("MFMA busy / CU busy cycles = %.3f (4 SIMDs per CU: %.1f %% of the matrix pipes' time)" % (r, 25 * r))
if g("SQ_INSTS_VALU") and g("SQ_INSTS_MFMA"):
    print("VALU instructions per MFMA instruction = %.2f" % ((g("SQ_INSTS_VALU") - g("SQ_INSTS_MFMA", 0)) / g("SQ_INSTS_MFMA")))
if g("SQ_LDS_IDX_ACTIVE"):
    print("LDS bank-conflict cycles / LDS active cycles = %.4f" % (g("SQ_LDS_BANK_CONFLICT", 0.0) / g("SQ_LDS_IDX_ACTIVE")))
if g("SQ_LDS_IDX_ACTIVE") and g("SQ_BUSY_CU_CYCLES"):
    print("LDS active cycles / CU busy cycles = %.3f" % (g("SQ_LDS_IDX_ACTIVE") / g("SQ_BUSY_CU_CYCLES")))
if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None and g("TCC_HIT_sum") + g("TCC_MISS_sum") > 0:
    print("L2 hit rate = %.3f" % (g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))))
if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
    fb, wb = 2 * g("FETCH_SIZE") * 1024 / n["FETCH_SIZE"], g("WRITE_SIZE") * 1024 / n["WRITE_SIZE"]
    print("HBM-side traffic per launch: fetch %.1f MB (x2 gfx950 correction applied) + write %.1f MB" % (fb / 1e6, wb / 1e6))
    print("TRAFFIC_BYTES_PER_LAUNCH=%d" % int(fb + wb))
PY
find $O -name "*.csv" -delete
cat $O/summary.txt
