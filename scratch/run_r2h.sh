#!/bin/bash
export TMPDIR=/tmp
timeout -k 5 300 python -u -m pytest tests/test_grouped_activation_gpu.py -x -q --timeout 60 --timeout-method thread > gpurun_out/gact.log 2>&1
rc=$?
tail -40 gpurun_out/gact.log
[ $rc -eq 0 ] || exit $rc
timeout -k 5 300 python -u -m pytest tests/test_activation_gpu.py tests/test_grouped_gpu.py tests/test_grouped_mxfp8_gpu.py -x -q --timeout 60 --timeout-method thread 2>&1 | tail -4
