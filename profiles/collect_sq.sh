#!/bin/bash
# Run ON THE GPU BOX from the repo root: bash profiles/collect_sq.sh <tag>
# MFMA-utilisation evidence for the GEMM (north_star): two separate --pmc passes of profiles/pmc_target.py
# (counters only, no trace domains), each 5 SQ/GRBM counters at most; summarise with parse_sq.py.
set -e
TAG=${1:-run}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE -d $OUT/sq_a -o sq_a -- python3 profiles/pmc_target.py 6 > $OUT/sq_a.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/sq_b -o sq_b -- python3 profiles/pmc_target.py 6 > $OUT/sq_b.log 2>&1
find $OUT -name "*.db"
