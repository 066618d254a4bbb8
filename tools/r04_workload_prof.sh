#!/bin/bash
# One GPU visit for a --workload line: bench.py (timed line), then the same command under rocprofv3 --kernel-trace --stats (kernel_stats.csv kept).
# Usage: tools/r04_workload_prof.sh <tag> <workload> <dtype> [pytest files...]
TAG=$1; WL=$2; DT=$3; shift; shift; shift
R=$PWD; O=$R/gpurun_out; mkdir -p $O
if [ $# -gt 0 ]; then timeout 900 python -m pytest "$@" -x -q -m gpu -s 2>&1 | tail -12 | tee $O/${TAG}_tests.txt; fi
timeout 600 python bench.py --workload $WL --dtype $DT --cpu-seconds 0 --host-steps 0 > $O/${TAG}_${WL}_${DT}_bench.json 2> $O/${TAG}_bench.err; python -c "
import json,sys; d=json.loads(open('$O/${TAG}_${WL}_${DT}_bench.json').read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'dev', d.get('deviation_from_f32'))"
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $O/${TAG}_prof -- python $R/bench.py --workload $WL --dtype $DT --steps 3 --warmup 1 --cpu-seconds 0 --host-steps 0 --no-deviation > /dev/null 2> $O/${TAG}_prof.err)
find $O/${TAG}_prof -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_${WL}_${DT}_kernel_stats.csv \;
rm -rf $O/${TAG}_prof
python - <<EOF
import csv
rows=list(csv.DictReader(open('$O/${TAG}_${WL}_${DT}_kernel_stats.csv')))
for r in rows[:14]:
    print(r['Name'][:120].replace('ade::(anonymous namespace)::',''), r['Calls'], 'tot_ms', round(float(r['TotalDurationNs'])/1e6,1), 'max_ms', round(float(r['MaxNs'])/1e6,3), r['Percentage'])
EOF
