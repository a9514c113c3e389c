#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 120 python3 scratch/prof_host.py 2>&1 | grep "host enqueue" || exit $?
timeout -k 10 900 python3 -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/full_tests.log 2>&1 || { tail -40 gpurun_out/full_tests.log; exit 1; }
tail -1 gpurun_out/full_tests.log
timeout -k 10 300 python3 bench.py --no-grouped > gpurun_out/bench_h.json 2> gpurun_out/bench_h.err || exit $?
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_h.json").read().strip().splitlines()[-1])
print(round(d["value"],1), round(d["ms_per_step"],4), d["cpu_baseline"]["configs0_full"])
PY
