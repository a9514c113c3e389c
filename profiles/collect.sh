#!/bin/bash
# Run ON THE GPU BOX from the repo root (gpurun -- 'bash profiles/collect.sh <tag>').
# Three separate rocprofv3 runs of the same target: kernel trace + stats, FETCH_SIZE, WRITE_SIZE.
# Results land under gpurun_out/prof_<tag>/ ; summarise afterwards with export_stats.py / parse_pmc.py.
set -e
TAG=${1:-run}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/ktrace -o ktrace -- python3 profiles/pmc_target.py 30 > $OUT/ktrace.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o fetch -- python3 profiles/pmc_target.py 3 > $OUT/fetch.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o write -- python3 profiles/pmc_target.py 3 > $OUT/write.log 2>&1
find $OUT -name "*.db"
