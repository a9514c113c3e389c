#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 400 python3 -u -m pytest tests/test_quantize_gpu.py tests/test_linear_gpu.py tests/test_parity_r2_gpu.py tests/test_graph_gpu.py tests/test_grouped_gpu.py tests/test_grouped_mxfp8_gpu.py -x -q --timeout 120 --timeout-method thread > gpurun_out/r2m_tests.log 2>&1 || { tail -40 gpurun_out/r2m_tests.log; exit 1; }
tail -2 gpurun_out/r2m_tests.log
rm -rf gpurun_out/prof_r2m
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r2m -o bench -- python3 bench.py --no-cpu-baseline --no-grouped > gpurun_out/r2m.json 2> gpurun_out/r2m.err || exit $?
B=$(find gpurun_out/prof_r2m -name "*.db" | head -1)
python3 profiles/export_stats.py $B gpurun_out/r2m_stats.csv > /dev/null || exit $?
grep -i "delayed\|gemm\|cast_transpose" gpurun_out/r2m_stats.csv | cut -c1-60,200-400
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r2m.json").read().strip().splitlines()[-1])
print(round(d["value"],1), round(d["ms_per_step"],4), d["extra"]["step_breakdown"]["recipe_kernels_and_launch_gaps_ms"])
PY
