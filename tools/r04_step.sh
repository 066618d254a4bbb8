#!/bin/bash
# One GPU visit of the round-4 kernel work: a GPU test subset, a same-box comparison of builds of libade.so, and the instruction counters of the current build.
# Usage: tools/r04_step.sh <tag> "<lib1.so lib2.so ...>" [pytest args...]
TAG=$1; LIBS=$2; shift; shift
R=$PWD; O=$R/gpurun_out; mkdir -p $O
if [ $# -gt 0 ]; then timeout 1500 python -m pytest "$@" -x -q -m gpu 2>&1 | tail -15 | tee $O/${TAG}_tests.txt; fi
if [ -n "$LIBS" ]; then bash tools/abn_libs.sh 3 $LIBS 2>&1 | tee $O/${TAG}_ab.txt; fi
timeout 900 bash tools/pmc_pass.sh gpurun_out/${TAG}_pmc --other-steps 0 > $O/${TAG}_pmc.log 2>&1
python tools/pmc_summary.py gpurun_out/${TAG}_pmc 2>&1 | grep "kernel\|k_gtcrn\|k_front\|k_gtblock\|k_dpgrnn\|k_back" | tee $O/${TAG}_pmc_summary.txt
rm -rf $O/${TAG}_pmc
