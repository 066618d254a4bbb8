# round 6, job w: k_zip_ffx's projection outputs as whole rows through LDS (rows) against 16 bytes per row and instruction (dw0 = the tree before), same box; then the tests
for l in dw0 rows dw0 rows; do cp _ab/libade_$l.so audio_denoiser_onnx_amd/libade.so; echo -n "$l "; timeout 600 python bench.py --workload zipenhancer --dtype bf16 --cpu-seconds 0 --no-deviation --host-steps 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; done | tee gpurun_out/r06_w_zip_ffx_rows_ab.txt
cp _ab/libade_rows.so audio_denoiser_onnx_amd/libade.so
timeout 900 python -m pytest tests/test_zipenhancer.py -m gpu -x -q -k bf16 2>&1 | tail -2
