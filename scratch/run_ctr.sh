set -o pipefail
mkdir -p gpurun_out
timeout -k 10 200 python -u scratch/perf_ct_ragged.py > gpurun_out/ctr.log 2>&1 || { tail -5 gpurun_out/ctr.log; exit 1; }
grep GB gpurun_out/ctr.log
