#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r2j -o bench -- python3 bench.py --no-cpu-baseline --no-grouped > gpurun_out/r2j.json 2> gpurun_out/r2j.err || exit $?
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r2j.json").read().strip().splitlines()[-1])
print(d["roofline"]["launch_ms"]); print(d["roofline"]["launch_ms_min_median_max"])
PY
