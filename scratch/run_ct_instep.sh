set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/ct_instep.log
for p in 3 4 3 4 2; do
  NVTE_AMD_CT_WGS_PER_CU=$p timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-grouped --steps 40 > gpurun_out/b_$p.json 2> gpurun_out/b_$p.err || exit 1
  python3 - $p >> gpurun_out/ct_instep.log <<'PY'
import json,sys
d=json.loads(open(f"gpurun_out/b_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print("per_cu", sys.argv[1], "value", round(d["value"],1), "ms", round(d["ms_per_step"],4), "ct_dy", round(d["roofline_cast_transpose"]["launch_ms"]*1e3,1), "frac", round(d["roofline_cast_transpose"]["frac"],4), "ct_x", round(d["extra"]["in_step_ms"]["cast_transpose_x"]*1e3,1))
PY
done
cat gpurun_out/ct_instep.log
