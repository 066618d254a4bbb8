#!/bin/bash
# Same-box comparison of several builds of the product library.  Usage: tools/abn_libs.sh <rounds> <lib1.so> <lib2.so> ...
# Cycles through the builds <rounds> times on the box's one GPU (the default bench line, no CPU / host / other-workload legs), printing ms_per_step of every run.
N=$1; shift; L=audio_denoiser_onnx_amd/libade.so
cp $L /tmp/_keep.so
for i in $(seq $N); do
  for lib in "$@"; do
    cp $lib $L
    timeout 300 python bench.py --cpu-seconds 0 --other-steps 0 --host-steps 0 --steps 400 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$(basename $lib)', d['ms_per_step'])"
  done
done
cp /tmp/_keep.so $L
