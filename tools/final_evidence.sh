set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
bash tools/run_all_benches.sh gpurun_out/final/r02_s_all_benches.txt
for W in "gtcrn f32" "zipenhancer f32" "melband f32" "melband bf16" "mossformer f32"; do      # (round 2 also ran "zipenhancer bf16" and "mossformer bf16": that mode was removed in round 4)
  set -- $W
  python bench.py --workload $1 --dtype $2 2>/dev/null | tail -1 > gpurun_out/final/r02_s_$1_$2_bench.json
done
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/final/prof_gtcrn -- python $R/bench.py --steps 20 --warmup 5 --cpu-seconds 0 > $R/gpurun_out/final/prof_gtcrn.log 2>&1
cp $(ls $R/gpurun_out/final/prof_gtcrn/*/*kernel_stats.csv | head -1) $R/gpurun_out/final/r02_s_gtcrn_kernel_stats.csv
rm -rf $R/gpurun_out/final/prof_gtcrn
cd $R
bash tools/pmc_mfma_r02.sh gpurun_out/final/pmc
ls -la gpurun_out/final gpurun_out/final/pmc | head -40
