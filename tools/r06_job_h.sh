# round 6, job h: same-box A/B: TRA weight prefetch, depthwise taps before the segment skipped
O=gpurun_out; mkdir -p $O
for pair in "_ab/libade_base2.so _ab/libade_pf.so" "_ab/libade_pf.so _ab/libade_pf2.so" "_ab/libade_base2.so _ab/libade_pf2.so"; do
  timeout 600 python tools/ab_bench.py $pair 2>&1 | grep -v amdgpu.ids
done | tee $O/r06_h_ab.txt
cp _ab/libade_pf2.so audio_denoiser_onnx_amd/libade.so
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_segments.py tests/test_streaming.py -m gpu -x -q 2>&1 | tail -3
