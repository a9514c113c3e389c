#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/prof_norm
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_norm -o norm -- python3 scratch/perf_norm.py > gpurun_out/perf_norm.log 2> gpurun_out/perf_norm.err || exit $?
cat gpurun_out/perf_norm.log
K=$(find gpurun_out/prof_norm -name "*.db" | head -1)
python3 profiles/export_stats.py $K gpurun_out/norm_kernel_stats.csv || exit $?
grep -i "norm" gpurun_out/norm_kernel_stats.csv | cut -c1-260
