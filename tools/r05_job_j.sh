O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_zipenhancer.py tests/test_float_io.py -m gpu -x -q -s > $O/r05_j_tests.txt 2>&1; echo "tests rc $?"; tail -2 $O/r05_j_tests.txt; grep -a "bf16" $O/r05_j_tests.txt | cut -c1-500
for D in f32 bf16; do timeout 600 python bench.py --workload zipenhancer --dtype $D --cpu-seconds 0 --host-steps 0 > $O/r05_j_zip_$D.json 2> $O/r05_j_bench_$D.err; python -c "
import json; d=json.loads(open('$O/r05_j_zip_$D.json').read().strip().splitlines()[-1]); print('zip $D', d['ms_per_step'], d['roofline']['frac'], d.get('deviation_from_f32'))"; done
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/$O/r05_j_wprof -- python $GRAFT_REPO_ROOT/bench.py --workload zipenhancer --dtype bf16 --steps 3 --warmup 1 --cpu-seconds 0 --host-steps 0 --no-deviation > /dev/null 2>> $GRAFT_REPO_ROOT/$O/r05_j_bench.err)
find $O/r05_j_wprof -name "*kernel_stats.csv" -exec cp {} $O/r05_j_zip_bf16_kernel_stats.csv \; 2>/dev/null; rm -rf $O/r05_j_wprof
