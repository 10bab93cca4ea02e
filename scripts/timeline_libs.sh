#!/bin/bash
# timeline of one step for several libs
export PYTHONPATH=. TMPDIR=/tmp
cp vqa_amd/libvqa_amd.so /tmp/lib_keep.so
for l in "$@"; do
  cp $l vqa_amd/libvqa_amd.so
  n=$(basename $l .so)
  rm -rf gpurun_out/tl_$n
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_$n -- python bench.py --no-cpu-baseline --steps 6 --warmup 3 > /dev/null 2> gpurun_out/tl_$n.err || { cp /tmp/lib_keep.so vqa_amd/libvqa_amd.so; exit 1; }
  python scripts/timeline.py gpurun_out/tl_$n gpurun_out/tl_$n.txt || exit 1
  rm -rf gpurun_out/tl_$n
done
cp /tmp/lib_keep.so vqa_amd/libvqa_amd.so
