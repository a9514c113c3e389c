#!/bin/bash
export TMPDIR=/tmp
BENCH_DEBUG_HANG=150 timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 10 --warmup 3 --backend gloo --single-device > gpurun_out/dp2.json 2> gpurun_out/dp2.err || { tail -30 gpurun_out/dp2.err; exit 1; }
tail -1 gpurun_out/dp2.json | cut -c1-700
