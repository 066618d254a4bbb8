O=gpurun_out; mkdir -p $O
timeout 1200 python -m pytest tests/test_mossformer.py tests/test_hgtcrn.py -m gpu -x -q > $O/r05_y2_tests.txt 2>&1; echo "tests rc $?"; tail -2 $O/r05_y2_tests.txt
timeout 600 python bench.py --workload mossformer --dtype f32 --steps 5 --warmup 1 --cpu-seconds 0 --host-steps 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mossformer', d['ms_per_step'], d['roofline']['frac'])"
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_y -- python $GRAFT_REPO_ROOT/bench.py --workload mossformer --dtype f32 --steps 3 --warmup 1 --cpu-seconds 0 --host-steps 0 --no-deviation > /dev/null 2>&1)
find /tmp/prof_y -name "*kernel_stats.csv" -exec cp {} $O/r05_y2_mossformer_kernel_stats.csv \;
head -5 $O/r05_y2_mossformer_kernel_stats.csv | cut -c1-100,200-
