set -o pipefail
mkdir -p gpurun_out
for dbg in 0 210 211; do
  timeout -k 10 120 python -u scratch/perf_wait_stamps.py fprop $dbg 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/wait_stamps.log || exit 1
done
