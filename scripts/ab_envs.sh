#!/bin/bash
# Several environment settings on one box, alternating, without kernel-timing events.
# usage: scripts/ab_envs.sh runs "VAR=a" "VAR=b" ...   (an empty string = no setting)
export PYTHONPATH=. VQA_BENCH_NO_KERNEL_TIMING=1
runs=$1; shift
for i in $(seq 1 $runs); do
  for e in "$@"; do
    env $e timeout -k 10 200 python bench.py --no-cpu-baseline --steps 40 > /tmp/ab.json || exit 1
    python - "$e" <<'PY'
import json, sys; d = json.load(open('/tmp/ab.json')); print('%-30s %.3f ms/step' % (sys.argv[1] or '(default)', d['ms_per_step']))
PY
  done
done
