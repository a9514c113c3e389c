set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/full_tests.log 2>&1; rc=$?
tail -4 gpurun_out/full_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python -u scratch/perf_c3_step.py > gpurun_out/c3_step.log 2>&1 || { tail -5 gpurun_out/c3_step.log; exit 1; }
grep "ms/step" gpurun_out/c3_step.log
