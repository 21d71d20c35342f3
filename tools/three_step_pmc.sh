#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc3; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout -k 10 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/a -- python3 $GRAFT_REPO_ROOT/tools/three_step_pmc.py > /dev/null 2>> $OUT/err.txt; echo "a rc $?"
timeout -k 10 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/b -- python3 $GRAFT_REPO_ROOT/tools/three_step_pmc.py > /dev/null 2>> $OUT/err.txt; echo "b rc $?"
python3 - <<'PY'
import csv, glob, os, statistics
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc3"
for d in sorted(glob.glob(out + "/[ab]")):
    acc = {}
    for f in glob.glob(d + "/*/*_counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            k = "lbm3" if "lbm3_kernel" in row["Kernel_Name"] else ("lbm2" if "lbm2_kernel" in row["Kernel_Name"] else None)
            if k: acc.setdefault((k, row["Counter_Name"]), []).append(float(row["Counter_Value"]))
    for (k, c), v in sorted(acc.items()): print(k, c, statistics.median(v))
PY
