#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 400 python3 -u -m pytest tests/test_scaled_activation_gpu.py tests/test_activation_gpu.py -x -q --timeout 120 --timeout-method thread > gpurun_out/sc_tests.log 2>&1 || { tail -40 gpurun_out/sc_tests.log; exit 1; }
tail -2 gpurun_out/sc_tests.log
timeout -k 10 200 python3 scratch/perf_scaled_act.py 2>&1 | grep -v amdgpu.ids || exit $?
