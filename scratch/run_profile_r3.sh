#!/bin/bash
# GPU box: the round's profile passes (kernel trace of pmc_target, FETCH/WRITE passes, bench under rocprofv3, bench)
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/prof_r3
bash profiles/collect.sh r3 || exit $?
F=$(find gpurun_out/prof_r3/fetch -name "*.db" | head -1); W=$(find gpurun_out/prof_r3/write -name "*.db" | head -1); K=$(find gpurun_out/prof_r3/ktrace -name "*.db" | head -1)
python3 profiles/parse_pmc.py $F $W gpurun_out/round3_traffic.json || exit $?
python3 profiles/export_stats.py $K gpurun_out/round3_pmc_target_kernel_stats.csv || exit $?
cp gpurun_out/round3_traffic.json profiles/round3_traffic.json   # the bench runs below read it
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r3/bench -o bench -- python3 bench.py --no-cpu-baseline --no-grouped > gpurun_out/round3_bench_under_rocprof.json 2> gpurun_out/bench_rocprof.err || exit $?
B=$(find gpurun_out/prof_r3/bench -name "*.db" | head -1)
python3 profiles/export_stats.py $B gpurun_out/round3_bench_kernel_stats.csv || exit $?
timeout -k 10 400 python3 bench.py > gpurun_out/round3_bench.json 2> gpurun_out/bench.err || exit $?
tail -c 400 gpurun_out/round3_bench.json

python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/round3_bench.json").read().strip().splitlines()[-1])
print("\nvalue", round(d["value"], 1), "ms", round(d["ms_per_step"], 4), "frac", round(d["roofline"]["frac"], 4), "traffic", d["roofline"]["traffic"])
print({k: round(v["ms_per_step"], 4) for k, v in d["extra"]["step_variants"].items() if isinstance(v, dict)})
print("ct", round(d["roofline_cast_transpose"]["frac"], 4), "cast", round(d.get("roofline_cast", {}).get("frac", 0), 4))
print(d["extra"]["step_breakdown"])
PY
