#!/bin/bash
# one traced rehearsal with the rccl and copy candidates in the same process on the same buffers' box
cd "$(dirname "$0")/.."
OUT=gpurun_out/r04c; mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
cd /tmp && LT_SLAB_FORCE_P2P=1 timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/trace -- python3 $R/bench.py --slab --steps 100 --warmup 20 --batches 3 --driver two-step > $R/$OUT/trace.json 2> $R/$OUT/trace.err
echo "trace rc=$?"; cd $R
python tools/slab_timeline.py $OUT/trace $OUT/timeline.json
python -c "
import json;d=json.load(open('$OUT/trace.json'));t=d['config']['transport'];print(d['ms_per_step'],t['chosen'],t['warmup_ms_per_step'],t['failures'])"
for v in "LT_SLAB_COPY_STREAMS=2" "LT_SLAB_COPY_STREAMS=1" "LT_SLAB_COPY_STREAMS=2 LT_SLAB_COPY_ENGINE=0"; do
  n=$(echo $v | tr ' =' '__')
  env $v LT_SLAB_FORCE_P2P=1 timeout -k 10 600 python bench.py --slab --transport copy --driver two-step > $OUT/slab_$n.json 2> $OUT/slab_$n.err
  echo "$v rc=$? $(python -c "import json;d=json.load(open('$OUT/slab_$n.json'));print(d['ms_per_step'], d['batches_ms_per_step'], d['config']['transport'].get('copy_engine'), d['config']['transport']['failures'])")"
done
LT_SLAB_FORCE_P2P=1 timeout -k 10 600 python bench.py --slab --transport rccl --driver two-step > $OUT/slab_rccl.json 2> $OUT/slab_rccl.err
echo "rccl rc=$? $(python -c "import json;d=json.load(open('$OUT/slab_rccl.json'));print(d['ms_per_step'], d['batches_ms_per_step'])")"
