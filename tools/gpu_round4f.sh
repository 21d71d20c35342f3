#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out/r04f; mkdir -p $OUT
r() { n=$1; shift; timeout -k 10 300 python tools/slab_order_probe.py "$@" > $OUT/$n.jsonl 2> $OUT/$n.err; echo "== $n rc=$?"; cat $OUT/$n.jsonl; }
r o1 two-step/copy two-step/rccl two-step/copy two-step/rccl
r o2 single-step/rccl two-step/copy two-step/copy
r o3 two-step/copy two-step/copy single-step/copy two-step/copy
LT_PROBE_KEEP_CACHE=1 r o4 two-step/rccl two-step/copy two-step/rccl
