O=gpurun_out; mkdir -p $O
timeout 1200 python -m pytest tests/test_zipenhancer.py tests/test_float_io.py tests/test_full_size_properties.py -m gpu -x -q -s > $O/r05_l_tests.txt 2>&1; echo "tests rc $?"; tail -2 $O/r05_l_tests.txt; grep -a "bf16" $O/r05_l_tests.txt | cut -c1-300
for D in f32 bf16; do timeout 600 python bench.py --workload zipenhancer --dtype $D --cpu-seconds 0 --host-steps 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('zip $D', d['ms_per_step'], d['roofline']['frac'], d.get('deviation_from_f32'))"; done
