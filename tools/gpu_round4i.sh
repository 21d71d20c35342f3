#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out/r04i; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_engine.py -q -m gpu -k "fast_arithmetic or kbc" > $OUT/pytest_fast.log 2>&1; echo "pytest rc=$?"; tail -12 $OUT/pytest_fast.log
timeout -k 10 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["physical_frac"])
for r in d["other_configs"]:
    print(r["config"][:70], r["kernel"], r["avg_launch_ms"], r["MLUPS_wall"], r["physical_frac_of_8TBs"], r["check"].get("bit_identical"), r["check"].get("max_abs_diff"))
PY
