# round 6, job ag: Mel-Band bf16 attention with the softmax denominators in the O^T product (mbsum) against the vector-pipe sums (stats = the tree before), same box; then the bf16 tests
for l in stats mbsum stats mbsum; do cp _ab/libade_$l.so audio_denoiser_onnx_amd/libade.so; echo -n "$l "; timeout 900 python bench.py --workload melband --dtype bf16 --cpu-seconds 0 --host-steps 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], (d.get('deviation_from_f32') or {}).get('snr_db'))"; done | tee gpurun_out/r06_ag_melband_attn_sums_ab.txt
cp _ab/libade_mbsum.so audio_denoiser_onnx_amd/libade.so
timeout 1500 python -m pytest tests/test_melband.py -m gpu -x -q -s -k "bf16" 2>&1 | grep "dB\|passed\|failed" | cut -c1-400
