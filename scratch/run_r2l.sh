#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 120 python3 scratch/dbg_overlap_update.py 2>&1 | grep -v amdgpu.ids || exit $?
bash scratch/run_r2h.sh
