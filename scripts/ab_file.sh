#!/bin/bash
# Two versions of one source file on one box, alternating: usage scripts/ab_file.sh runs target fileA fileB [bench flags]
export PYTHONPATH=. VQA_BENCH_NO_KERNEL_TIMING=1
runs=$1; target=$2; fa=$3; fb=$4; shift 4
cp "$target" /tmp/ab_keep
for i in $(seq 1 $runs); do
  for f in "$fa" "$fb"; do
    cp "$f" "$target"
    timeout -k 10 200 python bench.py --no-cpu-baseline --steps 40 "$@" > /tmp/ab.json || { cp /tmp/ab_keep "$target"; exit 1; }
    python - "$f" <<'PY'
import json, sys; d = json.load(open('/tmp/ab.json')); print('%-40s %.3f ms/step' % (sys.argv[1], d['ms_per_step']))
PY
  done
done
cp /tmp/ab_keep "$target"
