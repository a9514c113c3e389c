#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gemm_splitk_gpu.py -x -v --timeout 180 --timeout-method thread > gpurun_out/splitk.log 2>&1
rc=$?
grep -E "PASSED|FAILED|passed|failed|Error" gpurun_out/splitk.log | tail -20
if [ $rc -ne 0 ]; then tail -50 gpurun_out/splitk.log; exit $rc; fi
timeout -k 10 900 python -u -m pytest tests/test_gemm_gpu.py tests/test_linear_gpu.py tests/test_grouped_gpu.py -x -q --timeout 180 --timeout-method thread 2>&1 | tail -5 || exit 1
timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-grouped > gpurun_out/bench_splitk.json 2> gpurun_out/bench_splitk.err || exit 1
python3 - <<'PY'
import json
b=json.loads(open('gpurun_out/bench_splitk.json').read().strip().splitlines()[-1])
print(b['value'], b['ms_per_step'], b['roofline']['frac'], b['roofline']['launch_ms'], b['extra']['gemm_ms'])
PY
NVTE_AMD_GEMM_SPLITK=0 timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-grouped > gpurun_out/bench_nosplitk.json 2> gpurun_out/bench_nosplitk.err || exit 1
python3 - <<'PY'
import json
b=json.loads(open('gpurun_out/bench_nosplitk.json').read().strip().splitlines()[-1])
print(b['value'], b['ms_per_step'], b['roofline']['frac'], b['roofline']['launch_ms'], b['extra']['gemm_ms'])
PY
