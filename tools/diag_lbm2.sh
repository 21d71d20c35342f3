#!/bin/bash
# PMC diagnosis of the two-step kernel: where do the waves spend their time?  bash tools/diag_lbm2.sh <tag> [policy]
set -o pipefail
TAG=${1:-r02diag}
POL=${2:-0}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $OUT/counters.txt 2>&1
pass() {
  name=$1; shift
  timeout -k 10 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pmc_$name -- python3 $GRAFT_REPO_ROOT/tools/lbm2_driver.py $POL 8 > $OUT/pmc_$name.log 2>&1
  echo "pass $name rc $?"
}
pass A SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS && \
pass B SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE && \
pass C SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES && \
pass D TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum
python3 - <<PY
import csv, glob, statistics, json
out = {}
for f in glob.glob("$OUT/pmc_*/*/*_counter_collection.csv"):
    acc = {}
    for row in csv.DictReader(open(f)):
        if "lbm2_kernel" in row["Kernel_Name"]:
            acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    for k, v in acc.items():
        out[k] = statistics.median(v)
wc = out.get("SQ_WAVE_CYCLES", 0)
if wc:
    for k in sorted(out):
        print(f"{k:32s} {out[k]:16.0f}  {out[k]/wc:8.4f} of wave cycles")
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
PY
