set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_linear_gpu.py tests/test_modules_gpu.py tests/test_grouped_mxfp8_gpu.py tests/test_distributed_gpu.py tests/test_normalization_gpu.py -x -q --timeout 120 --timeout-method thread > gpurun_out/mxswz_tests.log 2>&1; rc=$?
tail -5 gpurun_out/mxswz_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -u scratch/perf_grouped_mx.py > gpurun_out/gmx.log 2>&1; echo "exit $?" >> gpurun_out/gmx.log
tail -12 gpurun_out/gmx.log
