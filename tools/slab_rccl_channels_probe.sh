OUT=gpurun_out/r03k; mkdir -p $OUT
run() { local name=$1; shift; env "$@" timeout -k 10 200 python bench.py --slab --steps 100 --warmup 20 --driver two-step > $OUT/$name.json 2> $OUT/$name.err; echo "$name $(python -c "import json; d=json.load(open('$OUT/$name.json')); print(d['ms_per_step'], d['batches_ms_per_step'])" 2>&1 | tail -1)"; }
run default X=1
run ch8  NCCL_MAX_P2P_NCHANNELS=8 NCCL_MIN_P2P_NCHANNELS=8
run ch16 NCCL_MAX_P2P_NCHANNELS=16 NCCL_MIN_P2P_NCHANNELS=16
run ch32 NCCL_MAX_P2P_NCHANNELS=32 NCCL_MIN_P2P_NCHANNELS=32
run lowprio TORCH_NCCL_HIGH_PRIORITY=0
run comm0 LT_SLAB_COMM_PRIORITY=0
run default2 X=1
