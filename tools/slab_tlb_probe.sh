#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/tlb; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for shape in "256 256 256 128" "512 256 128 64" "512 512 64 64" "1024 512 32 32"; do
  tag=$(echo $shape | tr ' ' 'x')
  timeout -k 10 120 rocprofv3 --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum --kernel-trace --output-format csv -d $OUT/a_$tag -- python3 $GRAFT_REPO_ROOT/tools/slab_tlb_probe.py $shape > /dev/null 2>> $OUT/err.txt; echo "$tag a rc $?"
  timeout -k 10 120 rocprofv3 --pmc TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_THRASHING_STALL_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/b_$tag -- python3 $GRAFT_REPO_ROOT/tools/slab_tlb_probe.py $shape > /dev/null 2>> $OUT/err.txt; echo "$tag b rc $?"
done
python3 - <<'PY'
import csv, glob, os, statistics
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/tlb"
for d in sorted(glob.glob(out + "/[ab]_*")):
    acc = {}
    for f in glob.glob(d + "/*/*_counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            if "lbm2_kernel" in row["Kernel_Name"]:
                acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    print(os.path.basename(d), {k: statistics.median(v) for k, v in acc.items()})
PY
