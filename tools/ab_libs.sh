#!/bin/bash
# Same-box A/B of two builds of the product library.  Usage: tools/ab_libs.sh <libA.so> <libB.so> [rounds]
# Alternates A B A B ... on the box's one GPU (the default bench line, no CPU leg), printing ms_per_step of every run.
A=$1; B=$2; N=${3:-3}; L=audio_denoiser_onnx_amd/libade.so
cp $L /tmp/_keep.so
run() { cp $1 $L; timeout 300 python bench.py --cpu-seconds 0 --other-steps 0 --host-steps 0 --steps 400 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', d['ms_per_step'])"; }
for i in $(seq $N); do run $A A; run $B B; done
cp /tmp/_keep.so $L
