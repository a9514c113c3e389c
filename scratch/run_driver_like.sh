#!/bin/bash
# what the driver runs at round end: GPU tests, smoke(), default bench
mkdir -p gpurun_out
timeout -k 10 900 python3 -u -m pytest tests/ -x -q -m gpu --timeout 120 --timeout-method thread > gpurun_out/driver_tests.log 2>&1 || { tail -40 gpurun_out/driver_tests.log; exit 1; }
tail -1 gpurun_out/driver_tests.log
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2 || exit $?
timeout -k 10 600 python3 bench.py > gpurun_out/driver_bench.json 2> gpurun_out/driver_bench.err || { tail -20 gpurun_out/driver_bench.err; exit 1; }
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/driver_bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("metric","value","unit","n_gpus","steps","warmup","ms_per_step","scaling","vs_baseline","dtype")})
print(d["roofline"]["frac"], d["roofline"]["traffic"], d["cpu_baseline"]["value"], d["cpu_baseline"]["kind"])
PY
