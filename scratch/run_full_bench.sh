#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python3 -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/full_tests.log 2>&1 || { tail -40 gpurun_out/full_tests.log; exit 1; }
tail -2 gpurun_out/full_tests.log
timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 40 > gpurun_out/bench_q.json 2> gpurun_out/bench_q.err || exit $?
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_q.json").read().strip().splitlines()[-1])
print(round(d["value"],1), round(d["ms_per_step"],4), "frac", round(d["roofline"]["frac"],4), "ct", round(d["roofline_cast_transpose"]["frac"],4))
print(d["extra"].get("c5_grouped_linear"))
PY
