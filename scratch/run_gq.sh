set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/prof_gq2
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_gq2 -o gq --output-format csv -- python3 scratch/perf_gq2.py > gpurun_out/gq2.log 2>&1 || { tail -5 gpurun_out/gq2.log; exit 1; }
