set -o pipefail
mkdir -p gpurun_out
timeout -k 10 200 python -u -m pytest tests/test_parity_r3_gpu.py -m gpu -x -q --timeout 120 --timeout-method thread -k "transposes" 2>&1 | tail -3 || exit 1
timeout -k 10 400 python bench.py --no-cpu-baseline --no-grouped --steps 30 --warmup 5 > gpurun_out/bench_variants.json 2> gpurun_out/bench_variants.err || { tail -5 gpurun_out/bench_variants.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_variants.json").read().strip().splitlines()[-1])
print("all:", round(d["ms_per_step"],4), "frac", round(d["roofline"]["frac"],4))
for k in ():
    pass
print({k: round(v["ms_per_step"],4) for k,v in d["extra"]["step_variants"].items() if isinstance(v,dict)})
print(d["extra"]["step_breakdown"])
PY
