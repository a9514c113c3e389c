#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/prof_c5
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_c5 -o c5 -- python3 scratch/perf_c5.py > gpurun_out/c5.log 2> gpurun_out/c5.err || { tail -20 gpurun_out/c5.err; exit 1; }
tail -1 gpurun_out/c5.log | cut -c1-300
K=$(find gpurun_out/prof_c5 -name "*.db" | head -1)
python3 profiles/export_stats.py $K gpurun_out/c5_kernel_stats.csv > /dev/null || exit $?
python3 - <<'PY'
import csv
rows=list(csv.reader(open("gpurun_out/c5_kernel_stats.csv")))[1:]
for r in rows[:14]: print(r[0][:90].replace("\n"," "), r[1], r[3], r[4])
PY
