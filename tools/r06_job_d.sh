# round 6, job d: same-box A/B, early history flag
O=gpurun_out; mkdir -p $O
timeout 600 python tools/ab_bench.py _ab/libade_intra.so _ab/libade_flag.so 2>&1 | grep -v amdgpu.ids | tee $O/r06_d_ab.txt
cp _ab/libade_flag.so audio_denoiser_onnx_amd/libade.so
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_segments.py tests/test_streaming.py -m gpu -x -q 2>&1 | tail -3
python tools/phase_latency.py 256 2>&1 | grep "stage \[" | tee $O/r06_d_phase_latency_256.txt
