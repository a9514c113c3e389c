set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/ctr.log
for p in 4 3 4 3; do
  echo "per_cu $p" >> gpurun_out/ctr.log
  NVTE_AMD_CT_WGS_PER_CU=$p timeout -k 10 200 python -u scratch/perf_ct_ragged.py 2>&1 | grep GB | tail -7 >> gpurun_out/ctr.log || exit 1
done
cat gpurun_out/ctr.log
