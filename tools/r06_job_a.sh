# round 6, job a: the new full-depth parity tests + the whole GPU suite + the default bench line with its new blocks (dfsmn, stft_operator, cpu threads_1)
O=gpurun_out; mkdir -p $O; R=$PWD
timeout 900 python -m pytest tests/test_melband.py tests/test_mossformer.py -m gpu -x -q -s -k "full_depth or production_size or 24_layer or bf16_path_vs_reference" > $O/r06_a_full_depth_tests.txt 2>&1; echo "full-depth tests rc $?"
grep -E "k = |depth 6|24 layers|passed|failed|Error|assert|bf16 vs reference|d6_151|l24" $O/r06_a_full_depth_tests.txt | cut -c1-400 | tail -60
timeout 2400 python -m pytest tests -m gpu -x -q > $O/r06_a_gpu_tests.txt 2>&1; echo "gpu suite rc $?"; tail -5 $O/r06_a_gpu_tests.txt
timeout 1500 python bench.py > $O/r06_a_bench.json 2> $O/r06_a_bench.err; echo "bench rc $?"; tail -c 3000 $O/r06_a_bench.json; tail -3 $O/r06_a_bench.err
timeout 600 python bench.py --workload dfsmn --cpu-seconds 1 > $O/r06_a_dfsmn_bench.json 2>> $O/r06_a_bench.err; echo "dfsmn rc $?"; cut -c1-1500 $O/r06_a_dfsmn_bench.json
