# round 6, job c: same-box A/B of the DPGRNN variants (r05 | intra on MFMA + LDS hand-over | both recurrences on MFMA + hand-over)
O=gpurun_out; mkdir -p $O
for pair in "_ab/libade_r05.so _ab/libade_intra.so" "_ab/libade_r05.so _ab/libade_both.so" "_ab/libade_intra.so _ab/libade_both.so"; do
  timeout 600 python tools/ab_bench.py $pair 2>&1 | grep -v amdgpu.ids
done | tee $O/r06_c_ab.txt
cp _ab/libade_intra.so audio_denoiser_onnx_amd/libade.so
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_segments.py tests/test_streaming.py -m gpu -x -q 2>&1 | tail -3
