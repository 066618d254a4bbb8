O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_hgtcrn.py -m gpu -x -q > $O/r05_t_tests.txt 2>&1; echo "tests rc $?"; tail -3 $O/r05_t_tests.txt
for K in 1 1; do ADE_HG_FUSED=$K timeout 300 python tools/bench_hgtcrn.py --batches 256 --steps 20 2>/dev/null | grep "B=" | sed "s/^/fused=$K /"; done
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/$O/r05_t_prof -- python $GRAFT_REPO_ROOT/tools/bench_hgtcrn.py --batches 256 --steps 5 > /dev/null 2>&1)
find $O/r05_t_prof -name "*kernel_stats.csv" -exec cp {} $O/r05_t_hgtcrn_kernel_stats.csv \; ; rm -rf $O/r05_t_prof; head -8 $O/r05_t_hgtcrn_kernel_stats.csv | cut -c1-60,140-230
