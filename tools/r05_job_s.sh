O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_hgtcrn.py tests/test_float_io.py -m gpu -x -q > $O/r05_s_tests.txt 2>&1; echo "tests rc $?"; tail -3 $O/r05_s_tests.txt
for K in 1 0 1 0; do ADE_HG_FUSED=$K timeout 300 python tools/bench_hgtcrn.py --batches 256 --steps 20 2>/dev/null | grep "B=" | sed "s/^/fused=$K /"; done
ADE_HG_FUSED=1 timeout 300 python tools/bench_hgtcrn.py --batches 16,64 --steps 20 2>/dev/null | grep "B="
