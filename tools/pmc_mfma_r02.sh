#!/bin/bash
# MFMA-busy counters of bench.py's GEMM-family workloads (counters only: --kernel-trace + --pmc, nothing else).  Usage: tools/pmc_mfma_r02.sh <outdir>
OUT=$1
R=$PWD
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
CNT="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE"
for W in "zipenhancer f32" "mossformer f32" "melband f32" "melband bf16"; do      # (round 2 also ran "zipenhancer bf16": that mode was removed in round 4)
  set -- $W
  tag=$1; [ "$2" != "f32" ] && tag=$1_$2
  rocprofv3 --kernel-trace --pmc $CNT -f csv -d $R/$OUT/$tag -- python $R/bench.py --workload $1 --dtype $2 --steps 2 --warmup 1 --cpu-seconds 0 --host-steps 0 > $R/$OUT/$tag.log 2>&1 || tail -3 $R/$OUT/$tag.log
  python $R/tools/pmc_mfma_summary.py $R/$OUT/$tag --json $R/$OUT/r02_${tag}_mfma_busy.json > $R/$OUT/r02_${tag}_mfma_busy.txt 2>&1
  tail -1 $R/$OUT/r02_${tag}_mfma_busy.txt
  rm -rf $R/$OUT/$tag/*/*kernel_trace.csv $R/$OUT/$tag/*/*counter_collection.csv   # keep gpurun_out small (the summaries are what is committed)
done
