#!/bin/bash
# Memory-side counters of the two-step sweep on a 256^3 domain (256 KiB planes) against one rank's 512 x 512 x 64 slab
# (1 MiB planes): request counts and occupancy ("LEVEL": sum of requests in flight per cycle -> average latency =
# LEVEL / requests), credit / tag / queue stalls at the L2's memory side, L1 -> L2 latencies.  Separate passes (a pass
# holds a few counters), --pmc with the kernel trace only.  Usage on the GPU box: bash tools/pmc_memory_side.sh <tag>
set -o pipefail
TAG=${1:-r04u}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
# (credit / tag / FIFO stall counters, the L1 -> L2 latency counters and the request-size counters made the profiled
# process abort inside rocprofv3 on this image -- four passes lost to their time limit in round 4; one pass is kept)
PASSES=(
 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum GRBM_GUI_ACTIVE"
)
for W in cfg2 slab; do
  i=0
  for P in "${PASSES[@]}"; do
    timeout -k 10 120 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/$W/pass$i -- python3 $GRAFT_REPO_ROOT/tools/profile_workload.py $W 24 > /dev/null 2>> $OUT/$W.err; echo "$W pass $i rc $?"
    i=$((i+1))
  done
done
python3 - <<P
import csv, glob, json, collections, statistics
out = {}
for w in ("cfg2", "slab"):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/%s/pass*/**/*counter_collection.csv" % w, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "lbm2_kernel" in k:
                per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out[w] = {k: {c: statistics.median(v) for c, v in cs.items()} for k, cs in per.items()}
    for k, cs in out[w].items():
        d = cs
        if d.get("TCC_EA0_RDREQ_sum"): d["avg_read_latency_cycles_at_memory_side"] = round(d["TCC_EA0_RDREQ_LEVEL_sum"] / d["TCC_EA0_RDREQ_sum"], 1)
        if d.get("TCC_EA0_WRREQ_sum"): d["avg_write_latency_cycles_at_memory_side"] = round(d["TCC_EA0_WRREQ_LEVEL_sum"] / d["TCC_EA0_WRREQ_sum"], 1)
        if d.get("TCP_TCC_READ_REQ_sum"): d["avg_l1_to_l2_read_latency_cycles"] = round(d["TCP_TCC_READ_REQ_LATENCY_sum"] / d["TCP_TCC_READ_REQ_sum"], 1)
        if d.get("TCP_TCC_WRITE_REQ_sum"): d["avg_l1_to_l2_write_latency_cycles"] = round(d["TCP_TCC_WRITE_REQ_LATENCY_sum"] / d["TCP_TCC_WRITE_REQ_sum"], 1)
json.dump(out, open("$OUT/memory_side_counters.json", "w"), indent=1)
for w in out:
    for k, d in out[w].items():
        print(w, k[10:70], {c: d[c] for c in d if c.startswith("avg_")})
P
rm -rf $OUT/cfg2/pass* $OUT/slab/pass*
