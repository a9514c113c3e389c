#!/bin/bash
mkdir -p gpurun_out
TE_AMD_LIBRARY=$PWD/scratch/lib_serp/libtransformer_engine.so timeout -k 10 200 python3 -u -m pytest tests/test_gemm_gpu.py -x -q --timeout 120 --timeout-method thread > gpurun_out/serp_tests.log 2>&1 || { tail -30 gpurun_out/serp_tests.log; exit 1; }
tail -1 gpurun_out/serp_tests.log
for i in 1 2 3; do
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-grouped --steps 40 > gpurun_out/serp_base_$i.json 2>gpurun_out/serp.err || exit $?
  TE_AMD_LIBRARY=$PWD/scratch/lib_serp/libtransformer_engine.so timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-grouped --steps 40 > gpurun_out/serp_var_$i.json 2>gpurun_out/serp.err || exit $?
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/serp_*_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d["ms_per_step"],4), round(d["value"],1), {k:round(v*1e3,1) for k,v in d["roofline"]["launch_ms"].items()})
PY
