set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/wait_stamps3.log
for role in fprop dgrad wgrad; do
  timeout -k 10 120 python -u scratch/perf_wait_stamps.py $role 0 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/wait_stamps3.log || exit 1
done
