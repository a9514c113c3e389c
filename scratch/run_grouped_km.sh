set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_parity_r3_gpu.py tests/test_grouped_gpu.py tests/test_grouped_mxfp8_gpu.py tests/test_modules_gpu.py tests/test_zz_no_alloc.py tests/test_gemm_gpu.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/grouped_km_tests.log 2>&1; rc=$?
tail -6 gpurun_out/grouped_km_tests.log
[ $rc -eq 0 ] || exit $rc
for mode in all none; do NVTE_AMD_FP8_TRANSPOSES=$mode timeout -k 10 200 python - <<'PY' || exit 1
import os, sys; sys.path.insert(0, '.')
import torch, bench
r = bench.grouped_c5(torch.device("cuda"), 10)
print(os.environ["NVTE_AMD_FP8_TRANSPOSES"], {k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items() if k.startswith("ms") or k == "tflops"})
PY
done
