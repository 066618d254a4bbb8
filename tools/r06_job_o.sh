# round 6, job o: ZipEnhancer bf16 kernel stats (post k_zip_ffx)
R=$PWD; O=$R/gpurun_out; mkdir -p $O
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $O/r06_o_wprof -- python $R/bench.py --workload zipenhancer --dtype bf16 --steps 3 --warmup 1 --cpu-seconds 0 --host-steps 0 --no-deviation > /dev/null 2>> $O/r06_o_bench.err)
find $O/r06_o_wprof -name "*kernel_stats.csv" -exec cp {} $O/r06_o_zip_bf16_kernel_stats.csv \; ; rm -rf $O/r06_o_wprof
head -30 $O/r06_o_zip_bf16_kernel_stats.csv | cut -c1-200
