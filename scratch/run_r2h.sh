#!/bin/bash
export TMPDIR=/tmp
timeout -k 5 300 python -u -m pytest tests/test_grouped_mxfp8_gpu.py -x -q --timeout 60 --timeout-method thread > gpurun_out/gmx2.log 2>&1
tail -25 gpurun_out/gmx2.log
