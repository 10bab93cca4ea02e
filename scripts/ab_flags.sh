#!/bin/bash
# Several bench.py flag sets on one box, alternating, without kernel-timing events.
# usage: scripts/ab_flags.sh runs "flags A" "flags B" ...
export PYTHONPATH=. VQA_BENCH_NO_KERNEL_TIMING=1
runs=$1; shift
for i in $(seq 1 $runs); do
  for f in "$@"; do
    timeout -k 10 200 python bench.py --no-cpu-baseline --steps 40 $f > /tmp/ab.json || exit 1
    python - "$f" <<'PY'
import json, sys; d = json.load(open('/tmp/ab.json')); print('%-45s %.3f ms/step' % (sys.argv[1] or '(default)', d['ms_per_step']))
PY
  done
done
