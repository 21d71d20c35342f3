#!/bin/bash
# rocprofv3 kernel trace of the other BASELINE configurations (tools/bench_configs.py).
set -o pipefail
TAG=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_configs -- python3 $GRAFT_REPO_ROOT/tools/bench_configs.py cfg4 cfg5 cfg4bgk cfg4bgk1 obst19 obst19_1 > $OUT/configs.jsonl 2> $OUT/configs.err; echo "rc $?"
grep config $OUT/configs.jsonl | cut -c1-300
