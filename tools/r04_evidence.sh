#!/bin/bash
# Round-4 evidence on one MI355X box: GPU tests, the bench line for both fused-path geometries, rocprofv3 kernel stats, PMC traffic + stall / instruction counters.
# Usage: tools/r04_evidence.sh <tag>     (writes gpurun_out/<tag>_*)
TAG=$1; R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/${TAG}_gpu_tests.txt 2>&1; echo "gpu tests rc $?"; tail -3 $O/${TAG}_gpu_tests.txt
timeout 300 python bench.py --geometry 0 --cpu-seconds 0 --other-steps 0 > $O/${TAG}_gtcrn_geometry0_bench.json 2> $O/${TAG}_bench0.err; echo "bench geo0 rc $?"
timeout 600 python bench.py > $O/${TAG}_gtcrn_bench.json 2> $O/${TAG}_bench.err; echo "bench rc $?"
timeout 300 python bench.py --geometry 1 --cpu-seconds 0 --other-steps 0 > $O/${TAG}_gtcrn_geometry1_bench.json 2>> $O/${TAG}_bench0.err
timeout 300 python bench.py --geometry 0 --cpu-seconds 0 --other-steps 0 > $O/${TAG}_gtcrn_geometry0_bench_b.json 2>> $O/${TAG}_bench0.err
timeout 300 python bench.py --cpu-seconds 0 --other-steps 0 > $O/${TAG}_gtcrn_bench_b.json 2>> $O/${TAG}_bench.err
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $O/${TAG}_prof -- python $R/bench.py --steps 100 --warmup 10 --cpu-seconds 0 --host-steps 0 --other-steps 0 > $O/${TAG}_prof_bench.json 2> $O/${TAG}_prof.err); echo "rocprof rc $?"
find $O/${TAG}_prof -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_gtcrn_kernel_stats.csv \;
timeout 900 bash tools/pmc_traffic.sh gpurun_out/${TAG}_traffic "$TAG" > $O/${TAG}_traffic.txt 2>&1; echo "traffic rc $?"; cp profiles/traffic_pmc.json $O/${TAG}_traffic_pmc.json
timeout 900 bash tools/pmc_pass.sh gpurun_out/${TAG}_pmc --other-steps 0 > $O/${TAG}_pmc.txt 2>&1; python tools/pmc_summary.py gpurun_out/${TAG}_pmc > $O/${TAG}_gtcrn_pmc_summary.txt 2>&1; echo "pmc rc $?"
rm -rf $O/${TAG}_prof $O/${TAG}_traffic/fetch $O/${TAG}_traffic/write $O/${TAG}_pmc/p1 $O/${TAG}_pmc/p2
head -c 1500 $O/${TAG}_gtcrn_bench.json; echo; head -c 600 $O/${TAG}_gtcrn_geometry0_bench.json; echo; cat $O/${TAG}_gtcrn_kernel_stats.csv | head -5; cat $O/${TAG}_gtcrn_pmc_summary.txt; tail -3 $O/${TAG}_traffic.txt
