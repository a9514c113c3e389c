#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
for layer in column row ln_column; do
  BENCH_DEBUG_HANG=150 timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29551 bench.py --gpus 2 --steps 5 --warmup 2 --backend gloo --single-device --parallelism tp_sp --tp-layer $layer --tokens 2048 > gpurun_out/tp2_$layer.json 2> gpurun_out/tp2_$layer.err || { tail -20 gpurun_out/tp2_$layer.err; exit 1; }
  python3 - "$layer" <<'PY'
import json,sys
d=json.loads(open(f"gpurun_out/tp2_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], round(d["value"],1), round(d["ms_per_step"],3), "comm_ms", round(d["comm_ms"],3), "overlap", d["overlap_frac"], d["config"]["parallelism"])
PY
done
