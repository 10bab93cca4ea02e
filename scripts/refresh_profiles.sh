#!/bin/bash
# Runs on the GPU box: the default bench line, the same command under rocprofv3 --kernel-trace --stats,
# and three counter passes (one --pmc set each) over a short run.  Everything lands in gpurun_out/prof_*.
set -o pipefail
export PYTHONPATH=. TMPDIR=/tmp
O=gpurun_out
rm -rf $O/prof_trace $O/prof_pmc_*
timeout -k 10 300 python bench.py > $O/prof_bench.json 2> $O/prof_bench.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_trace -- python bench.py --no-cpu-baseline \
    > $O/prof_bench_under_rocprof.json 2> $O/prof_rocprof.err || exit 1
find $O/prof_trace -name "*kernel_stats.csv" -exec cp {} $O/prof_kernel_stats.csv \;
python scripts/timeline.py $O/prof_trace $O/prof_timeline.txt || exit 1
find $O/prof_trace -name "*kernel_trace.csv" -delete
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1)
  VQA_BENCH_NO_KERNEL_TIMING=1 timeout -s KILL 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/prof_pmc_$tag -- \
      python bench.py --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2> $O/prof_pmc_$tag.err || exit 1
  for k in resblock_tiles "wgrad_tiles_kernel<9>" "gemm256_kernel" adam_fc1; do
    echo "== $k" >> $O/prof_pmc_summary.txt
    python scripts/pmc_dispatch.py $O/prof_pmc_$tag "$k" 2 >> $O/prof_pmc_summary.txt
  done
  if [ "$tag" = "SQ_VALU_MFMA_BUSY_CYCLES" ]; then rm -rf $O/prof_pmc_$tag; fi
done
python scripts/pmc_traffic.py $O/prof_pmc_FETCH_SIZE $O/prof_pmc_WRITE_SIZE resblock_tiles $O/prof_pmc_resblock_traffic.json || exit 1
python scripts/pmc_step_traffic.py $O/prof_pmc_FETCH_SIZE $O/prof_pmc_WRITE_SIZE $O/prof_pmc_step_traffic.json || exit 1
rm -rf $O/prof_pmc_FETCH_SIZE $O/prof_pmc_WRITE_SIZE
echo done
