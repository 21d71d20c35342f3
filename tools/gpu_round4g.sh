#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out/r04g; mkdir -p $OUT
r() { n=$1; shift; timeout -k 10 300 python tools/slab_order_probe.py two-step/rccl two-step/copy two-step/rccl two-step/copy > $OUT/$n.jsonl 2> $OUT/$n.err; echo "== $n rc=$?"; grep candidate $OUT/$n.jsonl; }
r default
GPU_MAX_HW_QUEUES=8 r hwq8
GPU_MAX_HW_QUEUES=16 r hwq16
LT_SLAB_COPY_STREAMS=1 r one_copy_stream
GPU_MAX_HW_QUEUES=2 r hwq2
