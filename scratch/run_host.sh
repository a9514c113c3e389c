#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 120 python3 scratch/prof_host.py 2>&1 | grep "host enqueue" || exit $?
timeout -k 10 600 python3 -u -m pytest tests/test_linear_gpu.py tests/test_modules_gpu.py tests/test_gemm_gpu.py tests/test_quantize_gpu.py tests/test_graph_gpu.py tests/test_null_pointers_gpu.py tests/test_parity_r2_gpu.py tests/test_distributed_gpu.py -x -q --timeout 120 --timeout-method thread > gpurun_out/host_tests.log 2>&1 || { tail -40 gpurun_out/host_tests.log; exit 1; }
tail -1 gpurun_out/host_tests.log
