#!/bin/bash
# One GPU session at the end of a round: parity tests, smoke, bench (default and driver flags), the slab rehearsal
# (all automatic candidates: RCCL and copy-engine transports) and its kernel timeline, rocprofv3 kernel trace of the
# judged command, PMC passes (FETCH_SIZE / WRITE_SIZE in separate runs, then SQ and TCC) for the cfg2 line, the other
# BASELINE workloads and the two slab workloads (tools/profile_workload.py), and the small probes whose output lives
# under profiles/.  tools/pmc_traffic.py <tag> then turns gpurun_out/<tag>/ into profiles/.
# Usage (on the GPU box, from the repo root): bash tools/gpu_round.sh <tag> [notests]
set -o pipefail
TAG=${1:-r04z}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
if [ "$2" != "notests" ]; then
echo "== pytest -m gpu" && timeout -k 10 1200 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
echo "== smoke" && timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/smoke.log; tail -4 $OUT/smoke.log
fi
echo "== bench" && timeout -k 10 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; cut -c1-600 $OUT/bench.json
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_flags.json 2>> $OUT/bench.err; echo "bench (driver flags) exit $?"
echo "== slab rehearsal" && timeout -k 10 400 python bench.py --slab > $OUT/slab_self_exchange.json 2> $OUT/slab.err; echo "slab exit $?"; cut -c1-400 $OUT/slab_self_exchange.json
timeout -k 10 300 python bench.py --slab --steps 20 --warmup 5 --no-cpu-baseline > $OUT/slab_self_exchange_driver_flags.json 2>> $OUT/slab.err; echo "slab (driver flags) exit $?"
timeout -k 10 300 python bench.py --slab --workload cfg5 --no-cpu-baseline > $OUT/slab_self_exchange_cfg5.json 2>> $OUT/slab.err; echo "slab cfg5 exit $?"
export TMPDIR=/tmp
cd /tmp
echo "== rocprofv3 kernel trace of the judged command"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py > $OUT/trace_bench.json 2> $OUT/trace.err; echo "trace exit $?"
echo "== rocprofv3 kernel trace of the slab rehearsal"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_slab -- python3 $GRAFT_REPO_ROOT/bench.py --slab --steps 100 --batches 3 --driver two-step --no-cpu-baseline > $OUT/trace_slab_bench.json 2> $OUT/trace_slab.err; echo "slab trace exit $?"
python3 $GRAFT_REPO_ROOT/tools/slab_timeline.py $OUT/trace_slab $OUT/slab_timeline.json > /dev/null; echo "timeline exit $?"
for W in cfg2 cfg4 cfg4bgk obst19 cfg5 slab slab5; do
  echo "== $W: kernel trace"
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/w_$W/trace -- python3 $GRAFT_REPO_ROOT/tools/profile_workload.py $W 200 > $OUT/w_$W.json 2> $OUT/w_$W.err; echo "rc $?"
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/w_$W/pmc_$C -- python3 $GRAFT_REPO_ROOT/tools/profile_workload.py $W 24 > /dev/null 2>> $OUT/w_$W.err; echo "pmc $W $C rc $?"
  done
done
for W in cfg2 cfg4bgk cfg4 slab; do
  echo "== $W: SQ / TCC"
  timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $OUT/w_$W/pmc_SQ -- python3 $GRAFT_REPO_ROOT/tools/profile_workload.py $W 24 > /dev/null 2>> $OUT/w_$W.err; echo "pmc SQ $W rc $?"
  timeout -k 10 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/w_$W/pmc_TCC -- python3 $GRAFT_REPO_ROOT/tools/profile_workload.py $W 24 > /dev/null 2>> $OUT/w_$W.err; echo "pmc TCC $W rc $?"
done
cd $GRAFT_REPO_ROOT
du -sh $OUT
echo "== small grids" && timeout -k 10 300 python tools/small_grid_bench.py > $OUT/small_grids.jsonl 2> $OUT/small_grids.err; echo "rc $?"
echo "== other configs" && timeout -k 10 300 python tools/bench_configs.py cfg1 cfg4 cfg4bgk cfg4bgk1 obst19 obst19_1 cfg5 > $OUT/other_configs.jsonl 2> $OUT/other_configs.err; echo "rc $?"
