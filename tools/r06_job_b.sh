# round 6, job b: DPGRNN recurrences batched on the matrix cores -- GTCRN GPU tests, same-box A/B against the round-5 library, phase clocks
O=gpurun_out; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_segments.py tests/test_streaming.py tests/test_hgtcrn.py -m gpu -x -q > $O/r06_b_gtcrn_tests.txt 2>&1; echo "gtcrn tests rc $?"; tail -5 $O/r06_b_gtcrn_tests.txt
timeout 600 python tools/ab_bench.py _ab/libade_r05.so _ab/libade_dpmfma.so > $O/r06_b_ab.txt 2>&1; cat $O/r06_b_ab.txt
timeout 300 python tools/phase_clock.py 256 > $O/r06_b_phase_clock.txt 2>&1; cat $O/r06_b_phase_clock.txt
timeout 300 python tools/phase_clock.py 256 _ab/libade_r05.so > $O/r06_b_phase_clock_r05.txt 2>&1; cat $O/r06_b_phase_clock_r05.txt
timeout 300 python bench.py --cpu-seconds 0 --other-steps 0 --host-steps 0 2>/dev/null | tail -1 | cut -c1-600
