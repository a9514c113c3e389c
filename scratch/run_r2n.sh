#!/bin/bash
mkdir -p gpurun_out
for i in 1 2 3; do
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-grouped --steps 40 > gpurun_out/r2n_pair_$i.json 2>gpurun_out/r2n.err || exit $?
  TE_AMD_NO_PAIR=1 timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-grouped --steps 40 > gpurun_out/r2n_sep_$i.json 2>gpurun_out/r2n.err || exit $?
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2n_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d["ms_per_step"],4), round(d["value"],1))
PY
