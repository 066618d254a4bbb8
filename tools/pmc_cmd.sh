#!/bin/bash
# The two rocprofv3 PMC passes of tools/pmc_pass.sh for an arbitrary command (counters only: no tracing domains besides --kernel-trace).
# Usage: tools/pmc_cmd.sh <outdir> <command ...>      then      python tools/pmc_summary.py <outdir>
OUT=$1; shift
R=$PWD
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA -f csv -d $R/$OUT/p1 -- "$@" > $R/$OUT/p1.log 2>&1 || tail -5 $R/$OUT/p1.log
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SMEM -f csv -d $R/$OUT/p2 -- "$@" > $R/$OUT/p2.log 2>&1 || tail -5 $R/$OUT/p2.log
