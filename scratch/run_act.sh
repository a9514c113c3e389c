#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 120 python3 scratch/dbg_norm_graph.py 2>&1 | grep -v amdgpu.ids || exit $?
timeout -k 10 600 python3 -u -m pytest tests/test_activation_gpu.py tests/test_grouped_activation_gpu.py -x -q --timeout 120 --timeout-method thread > gpurun_out/act_tests.log 2>&1 || { tail -40 gpurun_out/act_tests.log; exit 1; }
tail -2 gpurun_out/act_tests.log
timeout -k 10 300 python3 scratch/perf_act.py 2>&1 | grep -v amdgpu.ids || exit $?
