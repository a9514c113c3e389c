# the N > 1 code paths of bench.py over a world-size-1 RCCL group (forced collectives): what one GPU can execute of the
# driver's multi-GPU run.  Numbers are not scaling results.
set -o pipefail
mkdir -p gpurun_out
for mode in "--parallelism dp" "--parallelism tp_sp --tp-layer column" "--parallelism tp_sp --tp-layer row" "--parallelism tp_sp --tp-layer ln_column --gemm-cus 224"; do
  tag=$(echo $mode | tr -d ' -' | tr '_' 'x')
  BENCH_DEBUG_HANG=200 timeout -k 10 260 python bench.py --rehearse-multi $mode --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/rehearse_$tag.json 2> gpurun_out/rehearse_$tag.err || { echo "FAILED: $mode"; tail -15 gpurun_out/rehearse_$tag.err; exit 1; }
  python - "$tag" <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/rehearse_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], "ok:", round(d["value"], 1), "TFLOP/s", round(d["ms_per_step"], 3), "ms", d["config"]["parallelism"],
      {k: (round(d[k], 3) if isinstance(d[k], float) else d[k]) for k in ("comm_ms", "serial_ms_per_step", "overlap_frac") if k in d},
      "roofline" in d)
PY
done
