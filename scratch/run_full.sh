#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 240 --timeout-method thread > gpurun_out/full_gpu.log 2>&1
rc=$?
tail -6 gpurun_out/full_gpu.log
exit $rc
