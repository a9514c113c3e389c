#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 400 python3 -u -m pytest tests/test_normalization_gpu.py -x -q --timeout 120 --timeout-method thread > gpurun_out/norm_tests.log 2>&1 || { tail -40 gpurun_out/norm_tests.log; exit 1; }
tail -2 gpurun_out/norm_tests.log
timeout -k 10 200 python3 scratch/perf_norm.py 2>&1 | grep "12288" || exit $?
