set -o pipefail
mkdir -p gpurun_out
timeout -k 10 200 python -u scratch/perf_c3_step.py > gpurun_out/c3_step.log 2>&1 || { tail -5 gpurun_out/c3_step.log; exit 1; }
