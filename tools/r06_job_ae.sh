# round 6, job ae: ZipEnhancer bf16 kernel stats on the current tree
R=$PWD; O=$R/gpurun_out; mkdir -p $O
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $O/r06_ae_wprof -- python $R/bench.py --workload zipenhancer --dtype bf16 --steps 3 --warmup 1 --cpu-seconds 0 --host-steps 0 --no-deviation > /dev/null 2>> $O/r06_ae_bench.err)
find $O/r06_ae_wprof -name "*kernel_stats.csv" -exec cp {} $O/r06_ae_zip_bf16_kernel_stats.csv \; ; rm -rf $O/r06_ae_wprof
