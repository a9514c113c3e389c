set -o pipefail
mkdir -p gpurun_out
timeout -k 10 200 python -u scratch/perf_mxq.py > gpurun_out/mxq_base.log 2>&1 || exit 1
