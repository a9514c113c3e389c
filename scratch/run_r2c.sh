#!/bin/bash
# GPU box: distributed parity tests, then the round-2 profile passes.  Stops at the first step that
# ends with anything but a clean pass / ordinary test failure.
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests/test_distributed_gpu.py -x -v --timeout 240 --timeout-method thread > gpurun_out/dist.log 2>&1
rc=$?
tail -5 gpurun_out/dist.log
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "dist tests rc=$rc: stopping"; exit $rc; fi
bash profiles/collect.sh r2 || exit $?
F=$(find gpurun_out/prof_r2/fetch -name "*.db" | head -1); W=$(find gpurun_out/prof_r2/write -name "*.db" | head -1); K=$(find gpurun_out/prof_r2/ktrace -name "*.db" | head -1)
python3 profiles/parse_pmc.py $F $W gpurun_out/round2_traffic.json || exit $?
python3 profiles/export_stats.py $K gpurun_out/round2_pmc_target_kernel_stats.csv || exit $?
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r2/bench -o bench -- python3 bench.py --no-cpu-baseline --no-grouped > gpurun_out/round2_bench_under_rocprof.json 2> gpurun_out/bench_rocprof.err || exit $?
B=$(find gpurun_out/prof_r2/bench -name "*.db" | head -1)
python3 profiles/export_stats.py $B gpurun_out/round2_bench_kernel_stats.csv || exit $?
timeout -k 10 400 python3 bench.py > gpurun_out/round2_bench.json 2> gpurun_out/bench.err || exit $?
tail -c 600 gpurun_out/round2_bench.json
for layer in column row ln_column; do
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 6 --warmup 2 --parallelism tp_sp --tp-layer $layer --backend gloo --single-device --tokens 2048 > gpurun_out/tp_$layer.json 2> gpurun_out/tp_$layer.err || exit $?
  cat gpurun_out/tp_$layer.json
done
