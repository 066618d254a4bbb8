# round 6, job m: ZipEnhancer bf16, fused row-local runs (k_zip_ffx): bit-equality with the separate launches, the step time of both forms
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_zipenhancer.py -m gpu -x -q -k "fused_row_runs or bf16" 2>&1 | tail -3
for f in 1 0; do echo "ADE_ZIP_FUSE=$f"; ADE_ZIP_FUSE=$f timeout 600 python bench.py --workload zipenhancer --dtype bf16 --cpu-seconds 0 --no-deviation 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('ms_min'), d.get('ms_max'))"; done | tee $O/r06_m_zip_fuse.txt
