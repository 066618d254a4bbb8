O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_ulunas.py tests/test_ulunas_dynamic.py tests/test_float_io.py -m gpu -x -q > $O/r05_q_tests.txt 2>&1; echo "tests rc $?"; tail -2 $O/r05_q_tests.txt
for K in 1 0 1 0; do ADE_ULU_TA2=$K timeout 300 python tools/bench_ulunas.py --batches 256 --steps 20 | sed "s/^/ta2=$K /"; done
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/$O/r05_q_prof -- python $GRAFT_REPO_ROOT/tools/bench_ulunas.py --batches 256 --steps 10 > /dev/null 2>&1)
find $O/r05_q_prof -name "*kernel_stats.csv" -exec cp {} $O/r05_q_ulunas_kernel_stats.csv \; ; rm -rf $O/r05_q_prof; head -12 $O/r05_q_ulunas_kernel_stats.csv | cut -c1-150
