#!/bin/bash
# round 4, first GPU pass: (1) the failing commit of round 3 with and without the compiler pass at fault, (2) today's
# sources with two units built WITHOUT the Makefile's flag against the new slab-layout tests (does the first-use check
# catch the miscompile?), (3) the GPU test suite, (4) the bench line.  Output under gpurun_out/r04a/.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r04a; mkdir -p $OUT
bash tools/experiments/d977/run_variants.sh > $OUT/d977.txt 2>&1; cat $OUT/d977.txt
C=lettuce_amd/csrc/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/lib_noflag.so $(ls $C/*.o | grep -v "inst_d3q19_f32.o\|inst_d3q27_f32.o") tools/experiments/noflag/*.o || exit 1
LT_ENGINE_LIBRARY=/tmp/lib_noflag.so timeout -k 10 600 python -m pytest tests/test_gpu_engine.py -q -m gpu -k "slab_layout_is_bit_identical or first_use_check" -x > $OUT/noflag_pytest.log 2>&1
echo "noflag rc=$?"; tail -5 $OUT/noflag_pytest.log
LT_ENGINE_LIBRARY=/tmp/lib_noflag.so timeout -k 10 600 python -m pytest tests/test_gpu_engine.py -q -m gpu -k "slab_layout_is_bit_identical" > $OUT/noflag_pytest_all.log 2>&1
echo "noflag (all) rc=$?"; tail -3 $OUT/noflag_pytest_all.log
timeout -k 10 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -5 $OUT/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json
