# round 6, job r: ZipEnhancer bf16 attention core, second form (mask + position term + row sum inside issued instructions): tests, step time, kernel stats
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_zipenhancer.py -m gpu -x -q -k "bf16" 2>&1 | tail -4
timeout 600 python bench.py --workload zipenhancer --dtype bf16 --cpu-seconds 0 --host-steps 0 > $O/r06_r_zip_bf16_bench.json 2>> $O/r06_r_bench.err
python -c "import json; d=json.loads(open('$O/r06_r_zip_bf16_bench.json').read().strip().splitlines()[-1]); print('zip bf16', d['ms_per_step'], d['roofline']['frac'], d.get('deviation_from_f32'))"
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $O/r06_r_wprof -- python $R/bench.py --workload zipenhancer --dtype bf16 --steps 3 --warmup 1 --cpu-seconds 0 --host-steps 0 --no-deviation > /dev/null 2>> $O/r06_r_bench.err)
find $O/r06_r_wprof -name "*kernel_stats.csv" -exec cp {} $O/r06_r_zip_bf16_kernel_stats.csv \; ; rm -rf $O/r06_r_wprof
grep attn16 $O/r06_r_zip_bf16_kernel_stats.csv | cut -c1-60,180-260
