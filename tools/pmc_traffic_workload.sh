#!/bin/bash
# Fabric-side traffic of one bench.py workload step: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (counters only), summed over every kernel of a step.
# Usage: tools/pmc_traffic_workload.sh <outdir> <workload> [dtype]
OUT=$1; W=$2; DT=${3:-f32}; R=$PWD; mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
STEPS=2; WARM=1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -f csv -d $R/$OUT/$C -- python $R/bench.py --workload $W --dtype $DT --steps $STEPS --warmup $WARM --cpu-seconds 0 --host-steps 0 > $R/$OUT/$C.log 2>&1 || tail -3 $R/$OUT/$C.log
done
python3 - <<PY
import csv, glob, json
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("$R/$OUT/%s/*/*counter_collection.csv" % c)[0]
    tot[c] = sum(float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == c)
steps = $STEPS + $WARM
out = {"_how": "tools/pmc_traffic_workload.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of bench.py --workload $W --dtype $DT; counters summed over every "
               "kernel of the run and divided by its %d steps (set-up launches are negligible); bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 wide-read correction, "
               "MI355X_MICROARCH.md); fabric-side, Infinity-Cache hits included" % steps,
       "workload": "$W", "dtype": "$DT", "fetch_kib_per_step": tot["FETCH_SIZE"] / steps, "write_kib_per_step": tot["WRITE_SIZE"] / steps,
       "bytes_per_step": int((2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024 / steps)}
json.dump(out, open("$R/$OUT/r02_${W}_${DT}_traffic.json", "w"), indent=1)
print(out)
PY
rm -rf $R/$OUT/FETCH_SIZE $R/$OUT/WRITE_SIZE
