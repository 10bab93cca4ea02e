#!/bin/bash
# One environment variable, several values, alternating on one box: usage scripts/ab_envvals.sh runs VAR v1 v2 ... [-- bench flags]
export PYTHONPATH=. VQA_BENCH_NO_KERNEL_TIMING=1
runs=$1; var=$2; shift 2
vals=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do vals+=("$1"); shift; done
[ "$1" = "--" ] && shift
for i in $(seq 1 $runs); do
  for v in "${vals[@]}"; do
    export $var=$v
    timeout -k 10 200 python bench.py --no-cpu-baseline --steps 40 "$@" > /tmp/ab.json 2>/tmp/ab.err || { tail -5 /tmp/ab.err; exit 1; }
    python - "$var=$v" <<'PY'
import json, sys; d = json.load(open('/tmp/ab.json')); print('%-30s %.3f ms/step   fwd+bwd %s' % (sys.argv[1], d['ms_per_step'], d.get('executor_fwd_bwd', {}).get('ms_per_step')))
PY
  done
done
