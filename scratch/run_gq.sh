set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_grouped_gpu.py tests/test_grouped_mxfp8_gpu.py tests/test_zz_no_alloc.py -x -q --timeout 120 --timeout-method thread > gpurun_out/gq_tests.log 2>&1; rc=$?
tail -3 gpurun_out/gq_tests.log
[ $rc -eq 0 ] || exit $rc
: > gpurun_out/gq.log
for p in 4 8 16 64; do
  NVTE_AMD_GROUP_WGS_PER_CU=$p timeout -k 10 200 python -u scratch/perf_gq.py >> gpurun_out/gq.log 2>&1 || exit 1
done
