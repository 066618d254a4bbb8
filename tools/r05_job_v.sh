O=gpurun_out; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_segments.py tests/test_streaming.py tests/test_pipeline.py tests/test_hgtcrn.py tests/test_gtcrn_sandwich.py -m gpu -x -q > $O/r05_v_tests.txt 2>&1; echo "tests rc $?"; tail -3 $O/r05_v_tests.txt
for i in 1 2 3; do timeout 300 python bench.py --steps 100 --warmup 20 --cpu-seconds 0 --host-steps 0 --other-steps 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'])"; done
timeout 300 python tools/bench_hgtcrn.py --batches 256 2>&1 | tail -1
bash tools/pmc_workload.sh gpurun_out/r05_v_gtcrn_pmc --other-steps 0 > gpurun_out/r05_v_gtcrn_pmc_kernels.txt 2>&1; rm -rf gpurun_out/r05_v_gtcrn_pmc/p?; cut -c1-50,66-260 gpurun_out/r05_v_gtcrn_pmc_kernels.txt | head -6
