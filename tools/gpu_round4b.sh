#!/bin/bash
# round 4, second GPU pass: the copy-engine halo transport.  (1) its tests (processes sharing the GPU map each other's
# windows), (2) the slab rehearsal with all automatic candidates, (3) the same with the copy engine / flag variants,
# (4) a kernel trace of the copy candidate.  Output under gpurun_out/r04b/.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r04b; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_slab_gloo.py -x -q -m gpu -k "copy" > $OUT/pytest_copy.log 2>&1
rc=$?; echo "pytest copy rc=$rc"; tail -15 $OUT/pytest_copy.log
[ $rc -eq 0 ] || exit $rc
LT_SLAB_FORCE_P2P=1 timeout -k 10 600 python bench.py --slab > $OUT/slab_auto.json 2> $OUT/slab_auto.err; echo "slab auto rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04b/slab_auto.json"))
t = d["config"]["transport"]
print("value", d["value"], "ms/step", d["ms_per_step"], "chosen", t["chosen"]); print(" warmup", t["warmup_ms_per_step"]); print(" checks", t["checks"]); print(" failures", t["failures"]); print(" copy_engine", t.get("copy_engine"))
PY
for v in "LT_SLAB_COPY_ENGINE=0" "LT_SLAB_FLAG_HOW=0" "LT_SLAB_COPY_ENGINE=0 LT_SLAB_FLAG_HOW=0"; do
  n=$(echo $v | tr ' =' '__')
  env $v LT_SLAB_FORCE_P2P=1 timeout -k 10 600 python bench.py --slab --transport copy --driver two-step > $OUT/slab_$n.json 2> $OUT/slab_$n.err
  echo "$v rc=$? $(python -c "import json;d=json.load(open('$OUT/slab_$n.json'));print(d['ms_per_step'], d['config']['transport'].get('copy_engine'), d['config']['transport']['failures'])")"
done
export TMPDIR=/tmp
R=$PWD
cd /tmp && LT_SLAB_FORCE_P2P=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/trace_copy -- python3 $R/bench.py --slab --steps 100 --warmup 20 --batches 2 --driver two-step --transport copy > $R/$OUT/trace_copy.json 2> $R/$OUT/trace_copy.err
echo "trace rc=$?"; cd $R
find $OUT/trace_copy -name "*kernel_stats.csv" | head -1 | xargs head -12
