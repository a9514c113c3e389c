#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python3 -u -m pytest tests/test_multi_gemm_gpu.py -x -q --timeout 60 --timeout-method thread > gpurun_out/mg_tests.log 2>&1 || { tail -40 gpurun_out/mg_tests.log; exit 1; }
tail -2 gpurun_out/mg_tests.log
timeout -k 10 600 python3 -u -m pytest tests/test_grouped_gpu.py tests/test_grouped_mxfp8_gpu.py tests/test_zz_no_alloc.py tests/test_linear_gpu.py tests/test_modules_gpu.py tests/test_graph_gpu.py tests/test_gemm_gpu.py -x -q --timeout 120 --timeout-method thread > gpurun_out/mg_tests2.log 2>&1 || { tail -40 gpurun_out/mg_tests2.log; exit 1; }
tail -2 gpurun_out/mg_tests2.log
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-grouped --steps 40 > gpurun_out/bench_mg.json 2> gpurun_out/bench_mg.err || { tail -20 gpurun_out/bench_mg.err; exit 1; }
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_mg.json").read().strip().splitlines()[-1])
print(round(d["value"],1), round(d["ms_per_step"],4), "frac", round(d["roofline"]["frac"],4))
PY
