# round 6, job u: ZipEnhancer after the streaming-kernel pass (16-byte down / up-sampling, sub-pixel store on permuted rows, octet history norm, sliding-window depthwise conv, finer statistics chunks)
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 1200 python -m pytest tests/test_zipenhancer.py -m gpu -x -q 2>&1 | tail -4
for D in bf16 f32; do
timeout 600 python bench.py --workload zipenhancer --dtype $D --cpu-seconds 0 --host-steps 0 > $O/r06_u_zip_${D}_bench.json 2>> $O/r06_u_bench.err
python -c "import json; d=json.loads(open('$O/r06_u_zip_${D}_bench.json').read().strip().splitlines()[-1]); print('zip $D', d['ms_per_step'], d['roofline']['frac'], d.get('deviation_from_f32'))"
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $O/r06_u_wprof -- python $R/bench.py --workload zipenhancer --dtype $D --steps 3 --warmup 1 --cpu-seconds 0 --host-steps 0 --no-deviation > /dev/null 2>> $O/r06_u_bench.err)
find $O/r06_u_wprof -name "*kernel_stats.csv" -exec cp {} $O/r06_u_zip_${D}_kernel_stats.csv \; ; rm -rf $O/r06_u_wprof
done
