O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_zipenhancer.py -m gpu -x -q -s > $O/r05_e_tests.txt 2>&1; echo "tests rc $?"; tail -3 $O/r05_e_tests.txt; grep -a "bf16" $O/r05_e_tests.txt | cut -c1-600
timeout 600 python bench.py --workload zipenhancer --dtype bf16 --cpu-seconds 0 --host-steps 0 > $O/r05_e_zip_bf16_bench.json 2> $O/r05_e_bench.err; echo "zip bf16 rc $?"; cut -c1-400 $O/r05_e_zip_bf16_bench.json
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/$O/r05_e_wprof -- python $GRAFT_REPO_ROOT/bench.py --workload zipenhancer --dtype bf16 --steps 3 --warmup 1 --cpu-seconds 0 --host-steps 0 --no-deviation > /dev/null 2>> $GRAFT_REPO_ROOT/$O/r05_e_bench.err)
find $O/r05_e_wprof -name "*kernel_stats.csv" -exec cp {} $O/r05_e_zip_bf16_kernel_stats.csv \; 2>/dev/null; rm -rf $O/r05_e_wprof
timeout 600 python -m pytest tests/test_pipeline.py -m gpu -x -q > $O/r05_e_pipe_tests.txt 2>&1; tail -2 $O/r05_e_pipe_tests.txt
timeout 600 python bench.py --other-steps 0 --cpu-seconds 0 > $O/r05_e_gtcrn_bench.json 2>> $O/r05_e_bench.err; python -c "
import json; d=json.loads(open('$O/r05_e_gtcrn_bench.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['host_inclusive'])"
