#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 400 python3 -u -m pytest tests/test_quantize_gpu.py tests/test_parity_r2_gpu.py -x -q --timeout 120 --timeout-method thread > gpurun_out/q_tests.log 2>&1 || { tail -30 gpurun_out/q_tests.log; exit 1; }
tail -2 gpurun_out/q_tests.log
timeout -k 10 300 python3 scratch/perf_all_quant.py 2>&1 | grep -v amdgpu.ids > gpurun_out/perf_all_quant.log || exit $?
cat gpurun_out/perf_all_quant.log
