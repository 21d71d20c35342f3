#!/bin/bash
# Round 3: the slab rehearsal (bench.py --slab: 512 x 512 x 64, the rank exchanges its halo messages with itself through
# RCCL) with the schedules / paddings / segment lengths of the two-step slab driver.
# Usage on the GPU box: bash tools/slab_direct_probe.sh <outdir> [quick]
OUT=${1:-gpurun_out/r03d}
mkdir -p $OUT
run() {  # name, env... -- bench args after "--"
  local name=$1; shift
  local envs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done; [ "$1" == "--" ] && shift
  env "${envs[@]}" timeout -k 10 300 python bench.py --slab --steps 100 --warmup 20 "$@" > $OUT/$name.json 2> $OUT/$name.err
  echo "$name exit $? $(python -c "import json,sys; d=json.load(open('$OUT/$name.json')); t=d['config']['transport']; print(d['ms_per_step'], d['value'], t['chosen'], t['warmup_ms_per_step'], t['failures'], d['batches_ms_per_step'])" 2>&1 | tail -1)"
}
run auto                  X=1
run auto_nop2p            LT_SLAB_FORCE_P2P=0
run all                   X=1 -- --transport all
[ "$2" == "quick" ] && exit 0
run auto_p2pch2           NCCL_MAX_P2P_NCHANNELS=2 NCCL_MIN_P2P_NCHANNELS=2
run auto_p2pch4           NCCL_MAX_P2P_NCHANNELS=4 NCCL_MIN_P2P_NCHANNELS=4
run auto_seg30            LT_SLAB_RCCL_SEGMENT=30
run auto_nopad            LT_SLAB_PAD=0
run cfg5                  X=1 -- --workload cfg5
