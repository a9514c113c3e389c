set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_grouped_gpu.py tests/test_grouped_mxfp8_gpu.py tests/test_activation_gpu.py tests/test_zz_no_alloc.py -x -q --timeout 120 --timeout-method thread > gpurun_out/gq_tests.log 2>&1; rc=$?
tail -3 gpurun_out/gq_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python -u scratch/perf_gq.py > gpurun_out/gq.log 2>&1 || exit 1
timeout -k 10 200 python -u scratch/perf_gq1.py > gpurun_out/gq1.log 2>&1 || exit 1
timeout -k 10 300 python -u scratch/perf_grouped_mx.py > gpurun_out/gmx.log 2>&1 || exit 1
