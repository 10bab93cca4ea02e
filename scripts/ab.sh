#!/bin/bash
# A/B of an environment switch on one box: bench.py without kernel-timing events, alternating runs.
# usage: scripts/ab.sh VAR [runs]
export PYTHONPATH=. VQA_BENCH_NO_KERNEL_TIMING=1
for i in $(seq 1 ${2:-2}); do
  for v in "" 1; do
    if [ -z "$v" ]; then unset $1; else export $1=1; fi
    timeout -k 10 200 python bench.py --no-cpu-baseline --steps 40 > /tmp/ab.json || exit 1
    python - <<PY
import json; d = json.load(open('/tmp/ab.json')); print('$1=%s  %.3f ms/step' % ('$v' or '-', d['ms_per_step']))
PY
  done
done
