#!/bin/bash
# Round-6 final evidence on ONE MI355X box: the whole GPU test suite, the default bench line, GTCRN kernel stats / traffic / counters, and for the other BASELINE configs the
# --workload line, rocprofv3 kernel stats, fabric traffic and matrix-core busy.   Usage: tools/r06_final_evidence.sh <tag>      (writes gpurun_out/<tag>_*)
TAG=$1; R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -x -q > $O/${TAG}_gpu_tests.txt 2>&1; echo "gpu tests rc $?"; tail -2 $O/${TAG}_gpu_tests.txt
timeout 600 python bench.py > $O/${TAG}_gtcrn_bench.json 2> $O/${TAG}_bench.err; echo "bench rc $?"
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $O/${TAG}_prof -- python $R/bench.py --steps 100 --warmup 10 --cpu-seconds 0 --host-steps 0 --other-steps 0 > $O/${TAG}_prof_bench.json 2> $O/${TAG}_prof.err); echo "rocprof rc $?"
find $O/${TAG}_prof -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_gtcrn_kernel_stats.csv \; ; rm -rf $O/${TAG}_prof
timeout 900 bash tools/pmc_traffic.sh gpurun_out/${TAG}_traffic "$TAG" > $O/${TAG}_traffic.txt 2>&1; echo "traffic rc $?"; cp profiles/traffic_pmc.json $O/${TAG}_traffic_pmc.json
timeout 900 bash tools/pmc_pass.sh gpurun_out/${TAG}_pmc --other-steps 0 > $O/${TAG}_pmc.txt 2>&1; python tools/pmc_summary.py gpurun_out/${TAG}_pmc > $O/${TAG}_gtcrn_pmc_summary.txt 2>&1; echo "pmc rc $?"
rm -rf $O/${TAG}_traffic/fetch $O/${TAG}_traffic/write $O/${TAG}_pmc/p1 $O/${TAG}_pmc/p2
for W in "zipenhancer f32" "zipenhancer bf16" "melband f32" "melband bf16" "mossformer f32"; do
  set -- $W; N=$1; D=$2; S=${N}_${D}
  timeout 600 python bench.py --workload $N --dtype $D --cpu-seconds 0 --host-steps 0 > $O/${TAG}_${S}_bench.json 2>> $O/${TAG}_bench.err
  (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $O/${TAG}_wprof -- python $R/bench.py --workload $N --dtype $D --steps 3 --warmup 1 --cpu-seconds 0 --host-steps 0 --no-deviation > /dev/null 2>> $O/${TAG}_prof.err)
  find $O/${TAG}_wprof -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_${S}_kernel_stats.csv \; ; rm -rf $O/${TAG}_wprof
  (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -f csv -d $O/${TAG}_wmfma -- python $R/bench.py --workload $N --dtype $D --steps 2 --warmup 1 --cpu-seconds 0 --host-steps 0 --no-deviation --no-graph > /dev/null 2>> $O/${TAG}_prof.err)
  python tools/pmc_mfma_summary.py $O/${TAG}_wmfma --json $O/${TAG}_${S}_mfma_busy.json > $O/${TAG}_${S}_mfma_busy.txt 2>&1; rm -rf $O/${TAG}_wmfma
  for C in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --pmc $C -f csv -d $O/${TAG}_wt_$C -- python $R/bench.py --workload $N --dtype $D --steps 2 --warmup 1 --cpu-seconds 0 --host-steps 0 --no-deviation > /dev/null 2>> $O/${TAG}_prof.err)
  done
  python3 - <<PY
import csv, glob, json
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("$O/${TAG}_wt_%s/*/*counter_collection.csv" % c)[0]
    tot[c] = sum(float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == c)
steps = 3
out = {"_how": "tools/r06_final_evidence.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of bench.py --workload $N --dtype $D; counters summed over every "
               "kernel of the run and divided by its %d steps; bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 wide-read correction, MI355X_MICROARCH.md); fabric-side, "
               "Infinity-Cache hits included" % steps,
       "workload": "$N", "dtype": "$D", "build": "$TAG", "fetch_kib_per_step": tot["FETCH_SIZE"] / steps, "write_kib_per_step": tot["WRITE_SIZE"] / steps,
       "bytes_per_step": int((2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024 / steps)}
json.dump(out, open("$O/${TAG}_${S}_traffic.json", "w"), indent=1)
print("$S", out["bytes_per_step"] / 1e9, "GB per step")
PY
  rm -rf $O/${TAG}_wt_FETCH_SIZE $O/${TAG}_wt_WRITE_SIZE
  python -c "import json; d=json.loads(open('$O/${TAG}_${S}_bench.json').read().strip().splitlines()[-1]); print('$S', d['ms_per_step'], d['roofline']['frac'])"
done
head -c 1200 $O/${TAG}_gtcrn_bench.json; echo; head -4 $O/${TAG}_gtcrn_kernel_stats.csv; cat $O/${TAG}_gtcrn_pmc_summary.txt; tail -3 $O/${TAG}_traffic.txt
# canonical names bench.py looks for (profiles/r06_<workload>_<dtype>_traffic.json, r06_<workload>[_bf16]_mfma_busy.json)
for W in "zipenhancer f32" "zipenhancer bf16" "melband f32" "melband bf16" "mossformer f32"; do set -- $W; N=$1; D=$2
  cp $O/${TAG}_${N}_${D}_traffic.json $O/r06_${N}_${D}_traffic.json 2>/dev/null
  if [ $D = f32 ]; then cp $O/${TAG}_${N}_${D}_mfma_busy.json $O/r06_${N}_mfma_busy.json 2>/dev/null; else cp $O/${TAG}_${N}_${D}_mfma_busy.json $O/r06_${N}_${D}_mfma_busy.json 2>/dev/null; fi
done
for D in 2 3; do timeout 300 python tools/pipeline_probe.py $D 100; done > $O/${TAG}_pipeline_probe.txt 2>&1; grep depth $O/${TAG}_pipeline_probe.txt
for B in 1 64 256; do timeout 300 python tools/phase_latency.py $B > $O/${TAG}_phase_latency_B$B.txt 2>&1; done
# round 6 additions: DFSMN line + kernel stats, UL-UNAS / H-GTCRN lines, the bf16 error budget, the STFT operator's kernel durations
timeout 600 python bench.py --workload dfsmn --cpu-seconds 1 > $O/${TAG}_dfsmn_bench.json 2>> $O/${TAG}_bench.err
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $O/${TAG}_dprof -- python $R/bench.py --workload dfsmn --steps 3 --warmup 1 --cpu-seconds 0 --host-steps 0 > /dev/null 2>> $O/${TAG}_prof.err)
find $O/${TAG}_dprof -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_dfsmn_kernel_stats.csv \; ; rm -rf $O/${TAG}_dprof
timeout 300 python tools/bench_ulunas.py > $O/${TAG}_ulunas_bench.txt 2>&1; timeout 300 python tools/bench_hgtcrn.py > $O/${TAG}_hgtcrn_bench.txt 2>&1
timeout 600 python tools/zip_bf16_budget.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_zip_bf16_budget.txt
bash tools/r06_job_l.sh > /dev/null 2>&1; cp $O/r06_l_stft_kernel_us.txt $O/${TAG}_stft_kernel_us.txt
# per-kernel counters of the ZipEnhancer step, both dtypes (tools/pmc_workload.sh: stall split, instruction mix, matrix-core busy, LDS conflicts, FETCH / WRITE per kernel)
for D in bf16 f32; do timeout 1200 bash tools/pmc_workload.sh gpurun_out/${TAG}_zpmc_$D --workload zipenhancer --dtype $D > $O/${TAG}_zipenhancer_${D}_pmc_summary.txt 2>&1; rm -rf $O/${TAG}_zpmc_$D; done
