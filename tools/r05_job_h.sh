O=gpurun_out; mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -x -q > $O/r05_h_gpu_tests.txt 2>&1; echo "gpu tests rc $?"; tail -3 $O/r05_h_gpu_tests.txt
for D in f32 bf16; do timeout 600 python bench.py --workload zipenhancer --dtype $D --cpu-seconds 0 --host-steps 0 > $O/r05_h_zip_$D.json 2> $O/r05_h_bench_$D.err; python -c "
import json; d=json.loads(open('$O/r05_h_zip_$D.json').read().strip().splitlines()[-1]); print('zip $D', d['ms_per_step'], d['roofline']['frac'], d.get('deviation_from_f32'))"; done
timeout 600 python bench.py --other-steps 0 --cpu-seconds 0 > $O/r05_h_gtcrn_bench.json 2>> $O/r05_h_bench.err; python -c "
import json; d=json.loads(open('$O/r05_h_gtcrn_bench.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['host_inclusive'])"
