#!/bin/bash
export TMPDIR=/tmp
for lib in transformer_engine_amd/lib scratch/lib_nt transformer_engine_amd/lib scratch/lib_nt; do
  TE_AMD_LIBRARY=$lib/libtransformer_engine.so timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-grouped > gpurun_out/bench_x.json 2> gpurun_out/bench_x.err || exit 1
  python3 - "$lib" <<'PY'
import json,sys
b=json.loads(open('gpurun_out/bench_x.json').read().strip().splitlines()[-1])
print(sys.argv[1], round(b['value'],1), round(b['ms_per_step'],4), {k:round(v*1e3,1) for k,v in b['extra']['in_step_ms'].items()})
PY
done
