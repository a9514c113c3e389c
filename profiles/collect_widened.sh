set -e
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
for s in perf_norm perf_act2 perf_act_mx perf_weight_update perf_grouped2; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_w_$s -o $s -- python3 scratch/$s.py > gpurun_out/prof_w_$s.log 2>&1
  DB=$(find gpurun_out/prof_w_$s -name "*.db" | head -1)
  python3 profiles/export_stats.py $DB gpurun_out/round1_v5_${s}_kernel_stats.csv | head -4
done
