#!/bin/bash
export TMPDIR=/tmp
timeout -k 5 300 python -u -m pytest tests/test_normalization_gpu.py tests/test_modules_gpu.py -x -q --timeout 60 --timeout-method thread > gpurun_out/norm_tests.log 2>&1
rc=$?
tail -8 gpurun_out/norm_tests.log
[ $rc -eq 0 ] || exit $rc
echo "== wave kernel"; timeout -k 5 120 python -u scratch/perf_norm.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/perf_norm_wave.log
echo "== NVTE_AMD_NORM_WAVE=0"; NVTE_AMD_NORM_WAVE=0 timeout -k 5 120 python -u scratch/perf_norm.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/perf_norm_old.log
