#!/bin/bash
# round 4: same-box numbers for the slab rehearsal: the single-domain bench line, the rehearsal with all automatic
# candidates (RCCL and copy transports), edge_planes variants, and a traced run of the copy candidate.
cd "$(dirname "$0")/.."
OUT=gpurun_out/${1:-r04d}; mkdir -p $OUT
timeout -k 10 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$? $(python -c "import json;d=json.load(open('$OUT/bench.json'));print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])")"
timeout -k 10 600 python bench.py --slab > $OUT/slab_self_exchange.json 2> $OUT/slab.err; echo "slab rc=$?"
python - <<PY
import json
d = json.load(open("$OUT/slab_self_exchange.json")); b = json.load(open("$OUT/bench.json"))
t = d["config"]["transport"]
print("slab", d["value"], "ms/step", d["ms_per_step"], "chosen", t["chosen"], "=", round(b["ms_per_step"] / d["ms_per_step"], 4), "of the single-domain rate")
print(" warmup", t["warmup_ms_per_step"]); print(" checks", t["checks"]); print(" failures", t["failures"], t.get("copy_engine"))
PY
for e in 4 8; do
  LT_SLAB_EDGE_PLANES=$e timeout -k 10 600 python bench.py --slab --transport copy --driver two-step > $OUT/slab_copy_edge$e.json 2> $OUT/slab_copy_edge$e.err
  echo "edge planes $e rc=$? $(python -c "import json;d=json.load(open('$OUT/slab_copy_edge$e.json'));print(d['ms_per_step'], d['batches_ms_per_step'])")"
done
timeout -k 10 600 python bench.py --slab --steps 20 --warmup 5 > $OUT/slab_driver_flags.json 2>> $OUT/slab.err; echo "slab (driver flags) rc=$? $(python -c "import json;d=json.load(open('$OUT/slab_driver_flags.json'));print(d['ms_per_step'], d['config']['transport']['chosen'])")"
export TMPDIR=/tmp
R=$PWD
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/trace_slab -- python3 $R/bench.py --slab --steps 100 --warmup 20 --batches 3 --driver two-step > $R/$OUT/trace_slab.json 2> $R/$OUT/trace_slab.err
echo "trace rc=$?"; cd $R
python tools/slab_timeline.py $OUT/trace_slab $OUT/slab_timeline.json
