#!/bin/bash
# One GPU session: parity tests, smoke, bench, rocprofv3 kernel trace + PMC passes.
# Usage (on the GPU box, from the repo root): bash tools/gpu_round.sh <tag>
set -o pipefail
TAG=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
echo "== pytest -m gpu" && timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
echo "== smoke" && timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/smoke.log; tail -2 $OUT/smoke.log
echo "== bench" && timeout -k 10 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; cat $OUT/bench.json
export TMPDIR=/tmp
cd /tmp
echo "== rocprofv3 kernel trace"
# the same command as the judged bench line (default flags)
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py > $OUT/trace_bench.json 2> $OUT/trace.err; echo "trace exit $?"
for C in FETCH_SIZE WRITE_SIZE; do
  echo "== rocprofv3 --pmc $C"
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --batches 1 --no-verify --no-cpu-baseline > $OUT/pmc_${C}_bench.json 2> $OUT/pmc_$C.err; echo "pmc $C exit $?"
done
echo "== rocprofv3 --pmc SQ (occupancy / issue mix)"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $OUT/pmc_SQ -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --batches 1 --no-verify --no-cpu-baseline > $OUT/pmc_SQ_bench.json 2> $OUT/pmc_SQ.err; echo "pmc SQ exit $?"
echo "== rocprofv3 --pmc TCC (L2 hit/miss)"
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_TCC -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --batches 1 --no-verify --no-cpu-baseline > $OUT/pmc_TCC_bench.json 2> $OUT/pmc_TCC.err; echo "pmc TCC exit $?"
cd $GRAFT_REPO_ROOT
find $OUT -name "*counter_collection.csv" | head
du -sh $OUT
