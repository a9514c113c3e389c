set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/full_tests.log 2>&1; rc=$?
tail -4 gpurun_out/full_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -u scratch/perf_grouped_mx.py > gpurun_out/gmx.log 2>&1 || exit 1
grep -v amdgpu gpurun_out/gmx.log | tail -8
