#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
PYTHONPATH=. timeout -k 10 300 python -u scratch/perf_adam3.py 2>&1 | tee gpurun_out/adam3.log
