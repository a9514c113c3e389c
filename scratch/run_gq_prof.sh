set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/prof_gq
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_gq -o gq --output-format csv -- python3 scratch/perf_gq.py > gpurun_out/gq_prof.log 2>&1 || { tail -5 gpurun_out/gq_prof.log; exit 1; }
F=$(find gpurun_out/prof_gq -name "*kernel_stats.csv" | head -1)
cut -c1-200 $F | head -14 > gpurun_out/gq_kernel_stats.txt
