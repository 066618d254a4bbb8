#!/bin/bash
# rocprofv3 PMC passes for a bench.py --workload line (counters only: --kernel-trace + --pmc).  Usage: tools/pmc_workload.sh <outdir> <bench args...>
# Passes: stall split | instruction mix | matrix-core busy + LDS conflicts | FETCH_SIZE | WRITE_SIZE.  Summary: tools/pmc_kernels.py <outdir>
OUT=$1; shift
R=$PWD
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
run() { n=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -f csv -d $R/$OUT/$n -- python $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --host-steps 0 --no-deviation --no-graph "${ARGS[@]}" > $R/$OUT/$n.log 2>&1 || tail -5 $R/$OUT/$n.log; }
ARGS=("$@")
run p1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
run p2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_MFMA
run p3 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
run p4 FETCH_SIZE
run p5 WRITE_SIZE
cd $R && python tools/pmc_kernels.py $OUT
