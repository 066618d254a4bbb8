#!/bin/bash
# MFMA-busy counters of the GEMM-family benches (counters only: --kernel-trace + --pmc, nothing else).  Usage: tools/pmc_mfma.sh <outdir>
OUT=$1
R=$PWD
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE -f csv -d $R/$OUT/melband -- python $R/tools/bench_melband.py --batches 16 --steps 2 > $R/$OUT/melband.log 2>&1 || tail -3 $R/$OUT/melband.log
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE -f csv -d $R/$OUT/mossformer -- python $R/tools/bench_mossformer.py --batches 32 --steps 2 > $R/$OUT/mossformer.log 2>&1 || tail -3 $R/$OUT/mossformer.log
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE -f csv -d $R/$OUT/dfsmn -- python $R/tools/bench_dfsmn.py > $R/$OUT/dfsmn.log 2>&1 || tail -3 $R/$OUT/dfsmn.log
find $R/$OUT -name "*counter_collection.csv" | head
