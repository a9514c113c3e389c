#!/bin/bash
# 2 data-parallel ranks sharing the box's one GPU over gloo: rehearses the N > 1 code path of bench.py
# (two graphs + amax all-reduce on the side stream, roofline block, JSON line); timings mean nothing here
export TMPDIR=/tmp
mkdir -p gpurun_out
BENCH_DEBUG_HANG=150 timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29552 bench.py --gpus 2 --steps 5 --warmup 2 --backend gloo --single-device > gpurun_out/dp2.json 2> gpurun_out/dp2.err || { tail -20 gpurun_out/dp2.err; exit 1; }
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/dp2.json").read().strip().splitlines()[-1])
print(d["n_gpus"], round(d["value"],1), round(d["ms_per_step"],3), d["config"]["parallelism"], d["config"].get("amax_reduction"), "frac", round(d["roofline"]["frac"],3), sorted(d.keys()))
PY
