set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/norm_grid.log
for p in 100 75 50 150 100; do
  echo "pct $p" >> gpurun_out/norm_grid.log
  NVTE_AMD_NORM_GRID_PCT=$p timeout -k 10 200 python3 -u scratch/perf_norm.py 2>&1 | grep -v amdgpu >> gpurun_out/norm_grid.log || exit 1
done
