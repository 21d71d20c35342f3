#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out/r04j; mkdir -p $OUT
C=lettuce_amd/csrc/build
libs=""
for v in v1 v2 v3 v2ilp v0default; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/lib_fast_$v.so $(ls $C/*.o | grep -v "inst2_d3q19_f32.o") tools/experiments/fastvar/$v.o || exit 1
  libs="$libs /tmp/lib_fast_$v.so"
done
LT_AB_ARITH=fast timeout -k 10 300 python tools/same_buffer_ab.py $libs > $OUT/fast_variants.jsonl 2> $OUT/fast_variants.err; echo "rc=$?"; cat $OUT/fast_variants.jsonl
timeout -k 10 300 python tools/same_buffer_ab.py > $OUT/exact.jsonl 2>> $OUT/fast_variants.err; cat $OUT/exact.jsonl
timeout -k 10 1500 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest_gpu.log
