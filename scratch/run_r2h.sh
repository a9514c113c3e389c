#!/bin/bash
export TMPDIR=/tmp
timeout -k 5 200 python -u -m pytest tests/test_graph_gpu.py tests/test_gemm_gpu.py -x -q --timeout 60 --timeout-method thread > gpurun_out/graph_tests.log 2>&1
tail -15 gpurun_out/graph_tests.log
