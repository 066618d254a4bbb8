O=gpurun_out; mkdir -p $O; R=$PWD
timeout 300 python tools/bench_ulunas.py --batches 64,256 --steps 20 2>&1 | grep "B=" | tee $O/r05_z_ulunas_bench.txt
timeout 300 python tools/bench_hgtcrn.py --batches 64,256 --steps 20 2>&1 | grep "B=" | tee $O/r05_z_hgtcrn_bench.txt
for W in ulunas hgtcrn; do
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_$W -- python $R/tools/bench_$W.py --batches 256 --steps 10 > /dev/null 2>&1)
find /tmp/prof_$W -name "*kernel_stats.csv" -exec cp {} $O/r05_z_${W}_kernel_stats.csv \;
head -4 $O/r05_z_${W}_kernel_stats.csv | cut -c1-80
done
