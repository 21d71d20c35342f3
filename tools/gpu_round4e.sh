#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out/r04e; mkdir -p $OUT
run() { n=$1; shift; env "$@" timeout -k 10 600 python bench.py --slab ${ARGS} > $OUT/$n.json 2> $OUT/$n.err; echo "$n rc=$? $(python -c "import json;d=json.load(open('$OUT/$n.json'));t=d['config']['transport'];print(d['ms_per_step'], t['chosen'], t['warmup_ms_per_step'], t['failures'])")"; }
ARGS="--transport copy --driver two-step" run A_copy_alone X=1
ARGS="" run B_auto X=1
ARGS="" run C_auto_window_from_torch LT_SLAB_WINDOW_TORCH=1
ARGS="--transport copy --driver two-step" run D_copy_alone_window_from_torch LT_SLAB_WINDOW_TORCH=1
ARGS="--transport copy" run E_copy_both_drivers X=1
ARGS="" run F_auto_no_cache PYTORCH_NO_CUDA_MEMORY_CACHING=1
