# round 6, job i: run-form STFT operator -- its tests, the families that call it, the timing of both forms
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_stft_process.py tests/test_hgtcrn.py tests/test_ulunas.py tests/test_dfsmn.py -m gpu -x -q 2>&1 | tail -3
python tools/bench_stft.py 2>&1 | grep -v "amdgpu\|neighbour" | tee $O/r06_i_stft_run_form.txt
ADE_STFT_RUN=0 python tools/bench_stft.py 2>&1 | grep -v "amdgpu\|neighbour" | tee $O/r06_i_stft_pair_form.txt
for rp in 4 8 16; do echo "ADE_STFT_RUN_PAIRS=$rp"; ADE_STFT_RUN_PAIRS=$rp python tools/bench_stft.py 2>&1 | grep -v "amdgpu\|neighbour"; done | tee $O/r06_i_stft_run_pairs.txt
