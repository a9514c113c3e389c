#!/bin/bash
for k in 2 3 4 5; do
  echo "== WGS_PER_CU $k"
  NVTE_AMD_CT_WGS_PER_CU=$k timeout -k 10 200 python3 scratch/perf_all_quant.py 2>&1 | grep "Q2" || exit $?
done
