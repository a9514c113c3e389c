set -o pipefail
mkdir -p gpurun_out
TE_AMD_LIBRARY=$PWD/scratch/lib_pq/libtransformer_engine.so timeout -k 10 400 python -u -m pytest tests/test_gemm_gpu.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/pq_tests.log 2>&1; rc=$?
tail -5 gpurun_out/pq_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 500 python scratch/perf_gemm_ab.py scratch/lib_base/libtransformer_engine.so scratch/lib_pq/libtransformer_engine.so 3 2>&1 | tee gpurun_out/pq_ab.log
