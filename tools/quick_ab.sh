#!/bin/bash
# Quick measurement of the default bench line + VALU instruction counters of the chunk kernel.  Usage: tools/quick_ab.sh <tag>
TAG=$1; R=$PWD; O=$R/gpurun_out; mkdir -p $O
for i in 1 2; do timeout 300 python bench.py --cpu-seconds 0 --other-steps 0 --host-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], 'frac', d['roofline']['frac'])"; done
timeout 900 bash tools/pmc_pass.sh gpurun_out/${TAG}_pmc --other-steps 0 > $O/${TAG}_pmc.txt 2>&1; python tools/pmc_summary.py gpurun_out/${TAG}_pmc 2>&1 | grep "kernel\|k_gtcrn\|k_front\|k_gtblock\|k_dpgrnn\|k_back" | tee $O/${TAG}_pmc_summary.txt
rm -rf $O/${TAG}_pmc
