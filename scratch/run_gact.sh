set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -u scratch/perf_gact.py > gpurun_out/gact.log 2>&1 || { tail -8 gpurun_out/gact.log; exit 1; }
