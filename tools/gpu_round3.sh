#!/bin/bash
# One GPU session at the end of a round: parity tests, smoke, bench, rocprofv3 kernel trace of the judged command, PMC
# passes (FETCH_SIZE / WRITE_SIZE in separate runs, then SQ and TCC) for the cfg2 line AND for the other BASELINE
# workloads (tools/profile_workload.py), and the slab rehearsal with its kernel trace.
# Usage (on the GPU box, from the repo root): bash tools/gpu_round3.sh <tag> [notests]
set -o pipefail
TAG=${1:-r03}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
if [ "$2" != "notests" ]; then
echo "== pytest -m gpu" && timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
echo "== smoke" && timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/smoke.log; tail -3 $OUT/smoke.log
fi
echo "== bench" && timeout -k 10 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; cut -c1-600 $OUT/bench.json
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_flags.json 2>> $OUT/bench.err; echo "bench (driver flags) exit $?"
echo "== slab rehearsal" && timeout -k 10 300 python bench.py --slab > $OUT/slab_self_exchange.json 2> $OUT/slab.err; echo "slab exit $?"; cut -c1-400 $OUT/slab_self_exchange.json
timeout -k 10 300 python bench.py --slab --transport all > $OUT/slab_self_exchange_all.json 2>> $OUT/slab.err; echo "slab all exit $?"
timeout -k 10 300 python bench.py --slab --workload cfg5 > $OUT/slab_self_exchange_cfg5.json 2>> $OUT/slab.err; echo "slab cfg5 exit $?"
export TMPDIR=/tmp
cd /tmp
echo "== rocprofv3 kernel trace of the judged command"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py > $OUT/trace_bench.json 2> $OUT/trace.err; echo "trace exit $?"
echo "== rocprofv3 kernel trace of the slab rehearsal"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_slab -- python3 $GRAFT_REPO_ROOT/bench.py --slab --steps 100 --batches 2 --driver two-step > $OUT/trace_slab_bench.json 2> $OUT/trace_slab.err; echo "slab trace exit $?"
for W in cfg2 cfg4 cfg4bgk obst19 cfg5; do
  echo "== $W: kernel trace"
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/w_$W/trace -- python3 $GRAFT_REPO_ROOT/tools/profile_workload.py $W 200 > $OUT/w_$W.json 2> $OUT/w_$W.err; echo "rc $?"
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/w_$W/pmc_$C -- python3 $GRAFT_REPO_ROOT/tools/profile_workload.py $W 24 > /dev/null 2>> $OUT/w_$W.err; echo "pmc $W $C rc $?"
  done
done
echo "== cfg2: SQ / TCC"
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $OUT/w_cfg2/pmc_SQ -- python3 $GRAFT_REPO_ROOT/tools/profile_workload.py cfg2 24 > /dev/null 2>> $OUT/w_cfg2.err; echo "pmc SQ rc $?"
timeout -k 10 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/w_cfg2/pmc_TCC -- python3 $GRAFT_REPO_ROOT/tools/profile_workload.py cfg2 24 > /dev/null 2>> $OUT/w_cfg2.err; echo "pmc TCC rc $?"
for W in cfg4bgk cfg4; do
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $OUT/w_$W/pmc_SQ -- python3 $GRAFT_REPO_ROOT/tools/profile_workload.py $W 24 > /dev/null 2>> $OUT/w_$W.err; echo "pmc SQ $W rc $?"
done
cd $GRAFT_REPO_ROOT
du -sh $OUT
echo "== small grids" && timeout -k 10 300 python tools/small_grid_bench.py > $OUT/small_grids.jsonl 2> $OUT/small_grids.err; echo "rc $?"
echo "== cfg4 with KBC in the two-step kernel" && timeout -k 10 300 python tools/kbc_two_step_probe.py > $OUT/kbc_two_step.jsonl 2> $OUT/kbc_two_step.err; echo "rc $?"
echo "== other configs" && timeout -k 10 300 python tools/bench_configs.py cfg1 cfg4 cfg4bgk cfg4bgk1 obst19 obst19_1 cfg5 > $OUT/other_configs.jsonl 2> $OUT/other_configs.err; echo "rc $?"
echo "== three steps per launch" && timeout -k 10 300 python tools/three_step_probe.py > $OUT/three_step.jsonl 2> $OUT/three_step.err; echo "rc $?"
