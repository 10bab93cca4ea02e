#!/bin/bash
# Two builds of the library on one box, alternating, under any script: usage scripts/ab_lib_script.sh runs libA.so libB.so -- python script.py args
export PYTHONPATH=.
runs=$1; a=$2; b=$3; shift 4
cp vqa_amd/libvqa_amd.so /tmp/lib_keep.so
for i in $(seq 1 $runs); do
  for l in "$a" "$b"; do
    cp "$l" vqa_amd/libvqa_amd.so
    echo "== $l"
    timeout -k 10 200 "$@" 2>&1 | grep -v amdgpu.ids || { cp /tmp/lib_keep.so vqa_amd/libvqa_amd.so; exit 1; }
  done
done
cp /tmp/lib_keep.so vqa_amd/libvqa_amd.so
