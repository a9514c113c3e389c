#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-grouped > gpurun_out/r2o.json 2> gpurun_out/r2o.err || { tail -20 gpurun_out/r2o.err; exit 1; }
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r2o.json").read().strip().splitlines()[-1])
print(round(d["value"],1), round(d["ms_per_step"],4), d["extra"]["power_bound_check"])
PY
