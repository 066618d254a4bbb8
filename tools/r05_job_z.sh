O=gpurun_out; mkdir -p $O; R=$PWD
timeout 1200 python -m pytest tests/test_zipenhancer.py -m gpu -x -q > $O/r05_za_tests.txt 2>&1; echo "tests rc $?"; tail -3 $O/r05_za_tests.txt
for D in bf16 f32; do timeout 600 python bench.py --workload zipenhancer --dtype $D --steps 10 --warmup 2 --cpu-seconds 0 --host-steps 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('zipenhancer $D', d['ms_per_step'], d['roofline']['frac'], d.get('deviation_from_f32'))"; done
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_za -- python $R/bench.py --workload zipenhancer --dtype bf16 --steps 3 --warmup 1 --cpu-seconds 0 --host-steps 0 --no-deviation > /dev/null 2>&1)
find /tmp/prof_za -name "*kernel_stats.csv" -exec cp {} $O/r05_za_zip_bf16_kernel_stats.csv \;
grep -E "attn16" $O/r05_za_zip_bf16_kernel_stats.csv | cut -c1-70,330-420
