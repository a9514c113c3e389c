# expert-parallel bench mode: N = 1 on the GPU, then a 2-rank rehearsal of the launch / barrier / max-over-ranks path
# (both ranks on the one GPU of the box, gloo for the timing all-reduce: the path itself has no collective)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --parallelism ep --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_ep1.json 2> gpurun_out/bench_ep1.err || { tail -5 gpurun_out/bench_ep1.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_ep1.json").read().strip().splitlines()[-1])
print("ep N=1:", round(d["value"], 1), "TFLOP/s", round(d["ms_per_step"], 3), "ms", d["config"]["parallelism"], d["scaling"])
PY
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --parallelism ep --backend gloo --single-device --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_ep2.json 2> gpurun_out/bench_ep2.err || { tail -5 gpurun_out/bench_ep2.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_ep2.json").read().strip().splitlines()[-1])
print("ep N=2 (rehearsal, both ranks on one GPU):", round(d["value"], 1), "TFLOP/s", round(d["ms_per_step"], 3), "ms", d["config"]["parallelism"])
PY
