#!/bin/bash
# HBM-side traffic of the bench kernels: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (TCC slots), counters only.
# Usage: tools/pmc_traffic.sh <outdir>
set -e
OUT=$1; R=$PWD; mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $R/$OUT/fetch -- python $R/bench.py --steps 5 --warmup 2 --cpu-seconds 0 > $R/$OUT/fetch.log 2>&1 || tail -3 $R/$OUT/fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $R/$OUT/write -- python $R/bench.py --steps 5 --warmup 2 --cpu-seconds 0 > $R/$OUT/write.log 2>&1 || tail -3 $R/$OUT/write.log
python3 - <<PY
import csv, glob, collections, re
for kind in ("fetch", "write"):
    f = glob.glob("$R/$OUT/%s/*/*counter_collection.csv" % kind)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"])
        if m: agg[m.group(1)].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(kind, k, "dispatches", len(v), "avg counter", sum(v) / len(v))
PY
