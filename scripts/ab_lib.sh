#!/bin/bash
# Two builds of the library on one box, alternating: usage scripts/ab_lib.sh runs libA.so libB.so
export PYTHONPATH=. VQA_BENCH_NO_KERNEL_TIMING=1
runs=$1; shift
cp vqa_amd/libvqa_amd.so /tmp/lib_keep.so
for i in $(seq 1 $runs); do
  for l in "$@"; do
    cp "$l" vqa_amd/libvqa_amd.so
    timeout -k 10 200 python bench.py --no-cpu-baseline --steps 40 > /tmp/ab.json || exit 1
    python - "$l" <<'PY'
import json, sys; d = json.load(open('/tmp/ab.json')); print('%-30s %.3f ms/step' % (sys.argv[1], d['ms_per_step']))
PY
  done
done
cp /tmp/lib_keep.so vqa_amd/libvqa_amd.so
