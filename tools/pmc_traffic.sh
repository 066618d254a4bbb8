#!/bin/bash
# HBM-side traffic of the bench kernels: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (TCC slots), counters only.
# Writes profiles/traffic_pmc.json (tagged with the sha1 of csrc/, which bench.py checks before it reports `roofline.traffic`).
# Usage: tools/pmc_traffic.sh <outdir> <build label> [bench args]
set -e
OUT=$1; LABEL=$2; shift; shift; R=$PWD; mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $R/$OUT/fetch -- python $R/bench.py --steps 5 --warmup 2 --cpu-seconds 0 --host-steps 0 --other-steps 0 "$@" > $R/$OUT/fetch.log 2>&1 || tail -3 $R/$OUT/fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $R/$OUT/write -- python $R/bench.py --steps 5 --warmup 2 --cpu-seconds 0 --host-steps 0 --other-steps 0 "$@" > $R/$OUT/write.log 2>&1 || tail -3 $R/$OUT/write.log
cd $R
python3 - "$OUT" "$LABEL" <<'PY'
import csv, glob, collections, json, re, sys
sys.path.insert(0, ".")
from bench import source_sha1
out, label = sys.argv[1], sys.argv[2]
names = {"k_gtcrn_chunk": "gtcrn_chunk", "k_front": "front", "k_gtblock": "gtblock", "k_dpgrnn": "dpgrnn", "k_back": "back"}
res = collections.defaultdict(dict)
for kind in ("fetch", "write"):
    f = glob.glob("%s/%s/*/*counter_collection.csv" % (out, kind))[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"])
        if m and m.group(1) in names:
            agg[names[m.group(1)]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        res[k][kind + "_kib"] = round(sum(v) / len(v), 2)
        res[k]["dispatches"] = len(v)
        print(kind, k, "dispatches", len(v), "avg counter", sum(v) / len(v))
doc = {"_how": "tools/pmc_traffic.sh on one MI355X: rocprofv3 --kernel-trace --pmc FETCH_SIZE and, in a SEPARATE pass, --pmc WRITE_SIZE, of `python bench.py --steps 5 --warmup 2 "
               "--cpu-seconds 0` (batch 256). Values are per-dispatch averages in the counters' native KiB. Correction per MI355X_MICROARCH.md (HBM section): on gfx950 "
               "FETCH_SIZE reports half of the bytes of wide (16 B/lane) coalesced reads, so bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024; WRITE_SIZE is taken as-is (uncalibrated). "
               "Infinity-Cache hits are counted, so this is fabric-side traffic, an upper bound on HBM traffic.",
       "build": label, "source_sha1": source_sha1(), "batch": 256, "kernels": res}
json.dump(doc, open("profiles/traffic_pmc.json", "w"), indent=2)
print(json.dumps(doc["kernels"]))
PY
