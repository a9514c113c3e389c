set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gemm_gpu.py tests/test_parity_r2_gpu.py tests/test_parity_r3_gpu.py tests/test_grouped_gpu.py tests/test_grouped_mxfp8_gpu.py tests/test_multi_gemm_gpu.py tests/test_zz_no_alloc.py tests/test_linear_gpu.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/gemm_tests.log 2>&1; rc=$?
tail -30 gpurun_out/gemm_tests.log
exit $rc
