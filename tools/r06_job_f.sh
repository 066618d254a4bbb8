# round 6, job f: ZipEnhancer's dense blocks on IEEE half -- error budget, GPU tests of the family, the bf16 bench line
O=gpurun_out; mkdir -p $O
python tools/zip_bf16_budget.py 2>&1 | grep -v amdgpu.ids > $O/r06_f_zip_bf16_budget.txt; grep -A4 "^parts = 7\|^parts = [14]" $O/r06_f_zip_bf16_budget.txt | grep "parts\|wave vs"
timeout 1500 python -m pytest tests/test_zipenhancer.py -m gpu -x -q -s 2>&1 | grep -E "zipenhancer bf16|passed|failed|Error" | cut -c1-700
timeout 600 python bench.py --workload zipenhancer --dtype bf16 --cpu-seconds 0 2>/dev/null | tail -1 > $O/r06_f_zip_bf16_bench.json; python -c "
import json; d=json.loads(open('$O/r06_f_zip_bf16_bench.json').read()); print(d['ms_per_step'], d.get('deviation_from_f32'), d['roofline'].get('frac'))"
