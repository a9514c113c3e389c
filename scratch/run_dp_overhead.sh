set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/dpo_n1.json 2>gpurun_out/dpo_n1.err || { tail gpurun_out/dpo_n1.err; exit 1; }
timeout -k 10 200 python bench.py --rehearse-multi --parallelism dp --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/dpo_dp.json 2>gpurun_out/dpo_dp.err || { tail gpurun_out/dpo_dp.err; exit 1; }
python - <<'PY'
import json
for t in ("n1","dp"):
    d=json.loads(open(f"gpurun_out/dpo_{t}.json").read().strip().splitlines()[-1])
    print(t, round(d["ms_per_step"],4), d["config"].get("timed_as"), d["config"].get("ms_per_step_other_launch_form"))
PY
rm -rf gpurun_out/dpo_trace
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/dpo_trace -- python3 bench.py --rehearse-multi --parallelism dp --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/dpo_trace.json 2> gpurun_out/dpo_trace.err || { tail gpurun_out/dpo_trace.err; exit 1; }
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/dpo_trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last 40 kernels: name, start rel, dur, queue/stream
t0 = int(rows[-60]["Start_Timestamp"])
prev_end = t0
for r in rows[-60:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f'{(s-t0)/1e3:9.1f} us  dur {(e-s)/1e3:7.1f}  gap {(s-prev_end)/1e3:6.1f}  q{r.get("Queue_Id","?")} {r["Kernel_Name"][:70]}')
    prev_end = max(prev_end, e)
PY
