set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_parity_r3_gpu.py tests/test_rccl_single_gpu.py tests/test_grouped_mxfp8_gpu.py tests/test_cpp_api.py tests/test_distributed_gpu.py -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/r3_new.log 2>&1; rc=$?
tail -25 gpurun_out/r3_new.log
exit $rc
