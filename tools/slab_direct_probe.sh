#!/bin/bash
# Round 3: the slab rehearsal (bench.py --slab, 512 x 512 x 64, self exchange through RCCL) with the schedules /
# paddings / segment lengths of the two-step slab driver.  Usage on the GPU box: bash tools/slab_direct_probe.sh <outdir>
OUT=${1:-gpurun_out/r03d}
mkdir -p $OUT
run() {  # name, env... -- bench args after "--"
  local name=$1; shift
  local envs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done; [ "$1" == "--" ] && shift
  env "${envs[@]}" timeout -k 10 300 python bench.py --slab --steps 100 --warmup 20 "$@" > $OUT/$name.json 2> $OUT/$name.err
  echo "$name exit $? $(python -c "import json,sys; d=json.load(open('$OUT/$name.json')); t=d['config']['transport']; print(d['ms_per_step'], d['value'], t['chosen'], t['warmup_ms_per_step'], t['checks'], t['failures'], d['batches_ms_per_step'])" 2>&1 | tail -1)"
}
run all_pad               X=1 -- --transport all
run all_nopad             LT_SLAB_PAD=0 -- --transport all
run auto_seg0             LT_SLAB_RCCL_SEGMENT=0
run auto_seg30            LT_SLAB_RCCL_SEGMENT=30
run auto_seg20            LT_SLAB_RCCL_SEGMENT=20
run auto_seg15            LT_SLAB_RCCL_SEGMENT=15
run auto_edge4_seg0       LT_SLAB_EDGE_PLANES=4 LT_SLAB_RCCL_SEGMENT=0
run auto_pad2368          LT_SLAB_PAD=2368 LT_SLAB_RCCL_SEGMENT=0
run auto_pad1M            LT_SLAB_PAD=1048640 LT_SLAB_RCCL_SEGMENT=0
