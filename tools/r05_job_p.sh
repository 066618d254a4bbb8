O=gpurun_out; mkdir -p $O; R=$PWD
timeout 900 python -m pytest tests/test_ulunas.py tests/test_ulunas_dynamic.py -m gpu -x -q > $O/r05_p_tests.txt 2>&1; echo "tests rc $?"; tail -3 $O/r05_p_tests.txt
for C in 0 1; do for G in 1 2; do echo "chain $C groups $G"; ADE_ULU_CHAIN=$C ADE_ULU_GROUPS=$G timeout 300 python tools/bench_ulunas.py --batches 256 --steps 20 2>&1 | tail -1; done; done
(cd /tmp && export TMPDIR=/tmp && ADE_ULU_GROUPS=1 timeout 300 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_p -- python $R/tools/bench_ulunas.py --batches 256 --steps 10 > /dev/null 2>&1)
find /tmp/prof_p -name "*kernel_stats.csv" -exec cp {} $O/r05_p_ulunas_kernel_stats.csv \;
head -12 $O/r05_p_ulunas_kernel_stats.csv | cut -c1-90,160-
