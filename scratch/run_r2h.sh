#!/bin/bash
export TMPDIR=/tmp
timeout -k 5 400 python -u -m pytest tests/test_distributed_gpu.py -x -q --timeout 200 --timeout-method thread -k "configs3 or layernorm" > gpurun_out/dist2.log 2>&1
tail -15 gpurun_out/dist2.log
