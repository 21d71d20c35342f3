#!/bin/bash
# Address-translation counters of the two-step launch at 256^3 (2.6 GB of populations in flight) against 512^3 (20 GB):
# L1 TLB requests / hits / misses per launch and the share of time the L2 TLB is busy.  One --pmc pass per size, with the
# kernel trace only, each under its own short time limit.  Usage on the GPU box: bash tools/pmc_tlb.sh <tag>
set -o pipefail
TAG=${1:-r04zi}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for EDGE in 256 512; do
  export LT_PROFILE_EDGE=$EDGE
  timeout -k 10 120 rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/e$EDGE -- python3 $GRAFT_REPO_ROOT/tools/profile_workload.py cfg2 12 > $OUT/e$EDGE.json 2> $OUT/e$EDGE.err
  rc=$?; echo "edge $EDGE rc $rc"
  if [ $rc -ne 0 ]; then tail -3 $OUT/e$EDGE.err; break; fi
done
python3 - <<P
import csv, glob, json, collections, statistics
out = {}
for edge in (256, 512):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/e%d/**/*counter_collection.csv" % edge, recursive=True):
        for r in csv.DictReader(open(f)):
            if "lbm2_kernel" in r["Kernel_Name"]:
                per[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in per.items():
        d = {c: statistics.median(v) for c, v in cs.items()}
        d["launches_sampled"] = len(next(iter(cs.values())))
        if d.get("TCP_UTCL1_REQUEST_sum"):
            d["l1_tlb_miss_share"] = d["TCP_UTCL1_TRANSLATION_MISS_sum"] / d["TCP_UTCL1_REQUEST_sum"]
        if d.get("GRBM_GUI_ACTIVE"):
            d["l2_tlb_busy_share"] = d.get("GRBM_UTCL2_BUSY", 0.0) / d["GRBM_GUI_ACTIVE"]
        out["%d^3: %s" % (edge, k)] = d
json.dump(out, open("$OUT/tlb_counters.json", "w"), indent=1)
print(json.dumps(out, indent=1))
P
rm -rf $OUT/e256 $OUT/e512
