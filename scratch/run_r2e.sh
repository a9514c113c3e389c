#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_grouped_mxfp8_gpu.py -x -v --timeout 180 --timeout-method thread > gpurun_out/gmx.log 2>&1
rc=$?
grep -E "PASSED|FAILED|SKIPPED|passed|failed|Error|error" gpurun_out/gmx.log | tail -40
if [ $rc -ne 0 ]; then tail -60 gpurun_out/gmx.log; exit $rc; fi
timeout -k 10 600 python -u -m pytest tests/test_grouped_gpu.py tests/test_zz_no_alloc.py -x -q --timeout 180 --timeout-method thread 2>&1 | tail -5
