#!/bin/bash
# rocprofv3 kernel trace of the slab rehearsal (bench.py --slab, one rank exchanging with itself through RCCL)
OUT=${1:-gpurun_out/trace}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
shift
env "$@" LT_SLAB_FORCE_P2P=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT -- python3 $GRAFT_REPO_ROOT/bench.py --slab --steps 100 --warmup 20 --batches 2 --driver two-step > $GRAFT_REPO_ROOT/$OUT/bench.json 2> $GRAFT_REPO_ROOT/$OUT/bench.err
echo "trace exit $?"
cd $GRAFT_REPO_ROOT
find $OUT -name "*kernel_stats.csv" | head -2
