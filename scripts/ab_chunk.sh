export PYTHONPATH=. VQA_BENCH_NO_KERNEL_TIMING=1
for c in 24 32 48 64; do
  VQA_WGRAD_CHUNK=$c timeout -k 10 200 python bench.py --no-cpu-baseline --steps 40 > /tmp/ab.json || exit 1
  python - $c <<'PY'
import json, sys; d = json.load(open('/tmp/ab.json')); print('wgrad_chunk %s  %.3f ms/step' % (sys.argv[1], d['ms_per_step']))
PY
done
