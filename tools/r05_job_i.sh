O=gpurun_out; mkdir -p $O
timeout 1500 bash tools/pmc_workload.sh $O/r05_i_zip_bf16_pmc --workload zipenhancer --dtype bf16 > $O/r05_i_zip_bf16_pmc_summary.txt 2>&1; cat $O/r05_i_zip_bf16_pmc_summary.txt | cut -c1-230 | head -40
rm -rf $O/r05_i_zip_bf16_pmc/p*/
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/$O/r05_i_wprof -- python $GRAFT_REPO_ROOT/bench.py --workload zipenhancer --dtype bf16 --steps 3 --warmup 1 --cpu-seconds 0 --host-steps 0 --no-deviation > /dev/null 2>> $GRAFT_REPO_ROOT/$O/r05_i_bench.err)
find $O/r05_i_wprof -name "*kernel_stats.csv" -exec cp {} $O/r05_i_zip_bf16_kernel_stats.csv \; 2>/dev/null; rm -rf $O/r05_i_wprof
