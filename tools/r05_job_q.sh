O=gpurun_out; mkdir -p $O
timeout 600 python tools/gemm_bar_probe.py 2>&1 | tee $O/r05_q_gemm_bar_probe.txt
timeout 600 python -m pytest tests/test_ulunas.py tests/test_segments.py -m gpu -x -q 2>&1 | tail -3
