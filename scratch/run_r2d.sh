#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests/test_distributed_gpu.py -v --timeout 240 --timeout-method thread > gpurun_out/dist.log 2>&1
rc=$?
grep -E "PASSED|FAILED|SKIPPED|passed|failed" gpurun_out/dist.log | tail -20
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "dist tests rc=$rc: stopping"; exit $rc; fi
for layer in column row ln_column; do
  echo "== tp bench $layer"
  BENCH_DEBUG_HANG=100 timeout -k 10 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 6 --warmup 2 --parallelism tp_sp --tp-layer $layer --backend gloo --single-device --tokens 2048 > gpurun_out/tp_$layer.json 2> gpurun_out/tp_$layer.err || { tail -40 gpurun_out/tp_$layer.err; exit 1; }
  tail -1 gpurun_out/tp_$layer.json
done
