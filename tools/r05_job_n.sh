O=gpurun_out; mkdir -p $O; R=$PWD
for M in 0 1; do
  ADE_HG_WPE_MFMA=$M timeout 400 bash tools/pmc_cmd.sh gpurun_out/r05_n_pmc$M python $R/tools/bench_hgtcrn.py --batches 256 --steps 3
  python tools/pmc_summary.py gpurun_out/r05_n_pmc$M > $O/r05_n_pmc${M}_summary.txt 2>&1; grep -E "kernel|k_hg_wpe" $O/r05_n_pmc${M}_summary.txt
  (cd /tmp && export TMPDIR=/tmp && ADE_HG_WPE_MFMA=$M timeout 300 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_n$M -- python $R/tools/bench_hgtcrn.py --batches 256 --steps 5 > /dev/null 2>&1)
  find /tmp/prof_n$M -name "*kernel_stats.csv" -exec cp {} $O/r05_n_hgtcrn_m${M}_kernel_stats.csv \;
  head -4 $O/r05_n_hgtcrn_m${M}_kernel_stats.csv | cut -c1-50,180-
done
rm -rf gpurun_out/r05_n_pmc0/p? gpurun_out/r05_n_pmc1/p?
