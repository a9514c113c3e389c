#!/bin/bash
# GPU box: the round's profile passes (kernel trace of pmc_target, FETCH/WRITE passes, bench under rocprofv3, bench)
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/prof_r2
bash profiles/collect.sh r2 || exit $?
F=$(find gpurun_out/prof_r2/fetch -name "*.db" | head -1); W=$(find gpurun_out/prof_r2/write -name "*.db" | head -1); K=$(find gpurun_out/prof_r2/ktrace -name "*.db" | head -1)
python3 profiles/parse_pmc.py $F $W gpurun_out/round2_traffic.json || exit $?
python3 profiles/export_stats.py $K gpurun_out/round2_pmc_target_kernel_stats.csv || exit $?
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r2/bench -o bench -- python3 bench.py --no-cpu-baseline --no-grouped > gpurun_out/round2_bench_under_rocprof.json 2> gpurun_out/bench_rocprof.err || exit $?
B=$(find gpurun_out/prof_r2/bench -name "*.db" | head -1)
python3 profiles/export_stats.py $B gpurun_out/round2_bench_kernel_stats.csv || exit $?
timeout -k 10 400 python3 bench.py > gpurun_out/round2_bench.json 2> gpurun_out/bench.err || exit $?
tail -c 400 gpurun_out/round2_bench.json
