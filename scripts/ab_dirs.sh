#!/bin/bash
# Two sets of files (library and Python sources of vqa_amd/) on one box, alternating:
# usage scripts/ab_dirs.sh runs dirA dirB [bench flags]; every file in a directory replaces vqa_amd/<same name>
export PYTHONPATH=. VQA_BENCH_NO_KERNEL_TIMING=1
runs=$1; da=$2; db=$3; shift 3
mkdir -p /tmp/ab_keep && for f in $(ls $da); do cp vqa_amd/$f /tmp/ab_keep/; done
for i in $(seq 1 $runs); do
  for d in "$da" "$db"; do
    cp $d/* vqa_amd/
    timeout -k 10 200 python bench.py --no-cpu-baseline --steps 40 "$@" > /tmp/ab.json || { cp /tmp/ab_keep/* vqa_amd/; exit 1; }
    python - "$d" <<'PY'
import json, sys; d = json.load(open('/tmp/ab.json')); print('%-30s %.3f ms/step' % (sys.argv[1], d['ms_per_step']))
PY
  done
done
cp /tmp/ab_keep/* vqa_amd/
