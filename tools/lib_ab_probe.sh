#!/bin/bash
# A/B of builds of the engine library on the judged command: tools/lib_ab_probe.sh <tag> <other.so> [<other2.so> ...]
# (separate processes: +- 3 % from clocks and page placement alone -- tools/same_buffer_ab.py compares in one process)
TAG=$1; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT
line() { python - "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d["roofline"]
print(sys.argv[1], "GLUPS", round(d["value"] / 1e3, 2), "launch_ms", r.get("avg_launch_ms"), "verified", d.get("verified"))
PY
}
for i in 1 2; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-other-configs > $OUT/base$i.json 2> $OUT/base$i.err && line $OUT/base$i.json
  for lib in "$@"; do
    n=$(basename $lib .so)
    LT_ENGINE_LIBRARY=$lib timeout -k 10 200 python bench.py --no-cpu-baseline --no-other-configs > $OUT/$n$i.json 2> $OUT/$n$i.err && line $OUT/$n$i.json
  done
done
