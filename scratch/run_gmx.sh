set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -u scratch/perf_grouped_mx.py > gpurun_out/gmx.log 2>&1; echo "exit $?" >> gpurun_out/gmx.log
tail -12 gpurun_out/gmx.log
