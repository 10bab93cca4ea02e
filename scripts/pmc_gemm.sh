#!/bin/bash
# HBM traffic of the three FC1 GEMMs (one --pmc set per run).
export PYTHONPATH=. TMPDIR=/tmp
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf gpurun_out/pmc_gemm_$tag
  timeout -s KILL 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_gemm_$tag -- python scripts/bench_fc1_gemms.py > /dev/null 2> gpurun_out/pmc_gemm_$tag.err || exit 1
  echo "== $set"; python scripts/pmc_dispatch.py gpurun_out/pmc_gemm_$tag gemm_kernel 3
  rm -rf gpurun_out/pmc_gemm_$tag
done
