#!/bin/bash
cd /tmp; export TMPDIR=/tmp
R=/root/repo
run() { n=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -f csv -d /tmp/pm/$n -- $R/tests/unit/_build/gemm16_unit_gpu -t 1537920 384 1536 > /tmp/pm/$n.log 2>&1 || tail -3 /tmp/pm/$n.log; }
mkdir -p /tmp/pm
run p1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run p2 SQ_WAVES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM
run p3 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run p4 SQ_WAVES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
python3 - <<'P'
import csv,glob,collections
for n in ['p1','p2','p3','p4']:
    fs=glob.glob('/tmp/pm/%s/**/*counter_collection.csv'%n, recursive=True)
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for f in fs:
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'][:60]
            agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
    for k,v in agg.items():
        if 'BfStore' in k and 'Bias' not in k: print(n, k[-30:], {a:round(b/1e6,2) for a,b in v.items()})
P
