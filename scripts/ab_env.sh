#!/bin/bash
# Several environment settings of bench.py on one box, alternating, without kernel-timing events.
# usage: scripts/ab_env.sh runs "" "VAR=value" "VAR=value2 OTHER=1" ...
export PYTHONPATH=. VQA_BENCH_NO_KERNEL_TIMING=1
runs=$1; shift
for i in $(seq 1 $runs); do
  for e in "$@"; do
    ( for kv in $e; do export "$kv"; done
      timeout -k 10 200 python bench.py --no-cpu-baseline --steps 40 > /tmp/ab.json ) || exit 1
    python - "$e" <<'PY'
import json, sys; d = json.load(open('/tmp/ab.json')); print('%-45s %.3f ms/step' % (sys.argv[1] or '(default)', d['ms_per_step']))
PY
  done
done
