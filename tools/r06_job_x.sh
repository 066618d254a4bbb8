# round 6, job x: the f32 attention core with the bf16 core's placement (heads of a sequence back to back on one XCD), mask as a branch, exp2 multiply-add, ordinary-register results: rows = before
for l in rows attnf rows attnf; do cp _ab/libade_$l.so audio_denoiser_onnx_amd/libade.so; echo -n "$l "; timeout 600 python bench.py --workload zipenhancer --dtype f32 --cpu-seconds 0 --no-deviation --host-steps 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; done | tee gpurun_out/r06_x_zip_f32_attn_ab.txt
cp _ab/libade_attnf.so audio_denoiser_onnx_amd/libade.so
timeout 900 python -m pytest tests/test_zipenhancer.py -m gpu -x -q 2>&1 | tail -2
