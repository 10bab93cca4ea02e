#!/bin/bash
# Several builds of the library on one box, alternating, under bench.py: usage scripts/ab_libs_bench.sh runs libA.so libB.so ... [-- bench flags]
export PYTHONPATH=. VQA_BENCH_NO_KERNEL_TIMING=1
runs=$1; shift
libs=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do libs+=("$1"); shift; done
[ "$1" = "--" ] && shift
cp vqa_amd/libvqa_amd.so /tmp/lib_keep.so
for i in $(seq 1 $runs); do
  for l in "${libs[@]}"; do
    cp "$l" vqa_amd/libvqa_amd.so
    timeout -k 10 200 python bench.py --no-cpu-baseline --steps 40 "$@" > /tmp/ab.json 2>/tmp/ab.err || { tail -5 /tmp/ab.err; cp /tmp/lib_keep.so vqa_amd/libvqa_amd.so; exit 1; }
    python - "$l" <<'PY'
import json, sys; d = json.load(open('/tmp/ab.json')); print('%-24s %.3f ms/step   fwd+bwd %.3f' % (sys.argv[1], d['ms_per_step'], d.get('executor_fwd_bwd', {}).get('ms_per_step', 0)))
PY
  done
done
cp /tmp/lib_keep.so vqa_amd/libvqa_amd.so
