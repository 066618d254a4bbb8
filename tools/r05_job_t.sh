O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests/test_gemm16.py tests/test_melband.py tests/test_mossformer.py tests/test_dfsmn.py tests/test_stft_process.py tests/test_zipenhancer.py tests/test_ulunas.py -m gpu -x -q > $O/r05_t_tests.txt 2>&1; echo "tests rc $?"; tail -3 $O/r05_t_tests.txt
for W in "melband f32" "mossformer f32" "zipenhancer f32"; do set -- $W; timeout 600 python bench.py --workload $1 --dtype $2 --steps 5 --warmup 1 --cpu-seconds 0 --host-steps 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2', d['ms_per_step'], d['roofline']['frac'])"; done
