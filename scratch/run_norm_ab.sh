#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python3 -u -m pytest tests/test_normalization_gpu.py tests/test_linear_gpu.py tests/test_modules_gpu.py -x -q --timeout 120 --timeout-method thread > gpurun_out/norm_tests.log 2>&1 || { tail -30 gpurun_out/norm_tests.log; exit 1; }
tail -2 gpurun_out/norm_tests.log
timeout -k 10 200 python3 scratch/perf_norm.py 2>&1 | grep -v amdgpu.ids || exit $?
