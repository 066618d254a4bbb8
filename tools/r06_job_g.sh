cd /tmp && export TMPDIR=/tmp
R=/root/repo
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/stft_prof -- python $R/tools/bench_stft.py > $R/gpurun_out/r06_g_stft_bench.txt 2>&1
cat $R/gpurun_out/r06_g_stft_bench.txt | grep -v amdgpu
find $R/gpurun_out/stft_prof -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/r06_g_stft_kernel_stats.csv \;
rm -rf $R/gpurun_out/stft_prof
cut -d, -f1-4 $R/gpurun_out/r06_g_stft_kernel_stats.csv | cut -c1-200 | head -12
