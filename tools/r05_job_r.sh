O=gpurun_out; mkdir -p $O
timeout 300 python tools/bench_hgtcrn.py 2>&1 | tail -4
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/$O/r05_r_prof -- python $GRAFT_REPO_ROOT/tools/bench_hgtcrn.py --batches 256 --steps 5 > /dev/null 2>&1)
find $O/r05_r_prof -name "*kernel_stats.csv" -exec cp {} $O/r05_r_hgtcrn_kernel_stats.csv \; ; rm -rf $O/r05_r_prof; head -24 $O/r05_r_hgtcrn_kernel_stats.csv | cut -c1-70,140-230
