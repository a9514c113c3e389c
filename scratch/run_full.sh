set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/full_tests.log 2>&1; rc=$?
tail -3 gpurun_out/full_tests.log
exit $rc
