#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 400 python3 -u -m pytest tests/test_normalization_gpu.py -x -q --timeout 120 --timeout-method thread -k "few_rows" > gpurun_out/norm_tests.log 2>&1 || { tail -40 gpurun_out/norm_tests.log; exit 1; }
tail -2 gpurun_out/norm_tests.log
