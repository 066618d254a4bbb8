# round 6, job k: run-form STFT operator, kernel durations against the run length (ADE_STFT_RUN_PAIRS)
O=$PWD/gpurun_out; R=$PWD; cd /tmp; export TMPDIR=/tmp
for rp in 1 2 4 8 16 32; do
  ADE_STFT_RUN_PAIRS=$rp timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/stft_prof_$rp -- python $R/tools/bench_stft.py > /dev/null 2>&1
  f=$(find $O/stft_prof_$rp -name "*kernel_stats.csv" | head -1)
  echo "ADE_STFT_RUN_PAIRS=$rp"; python3 -c "
import csv
rows=[r for r in csv.DictReader(open('$f')) if 'k_stft_run' in r['Name']]
rows.sort(key=lambda r: r['Name'])
print('   ' + ' | '.join('%s %6.1f' % (r['Name'].split('k_stft_run_')[1].split('(')[0].replace(', false',''), float(r['AverageNs'])/1e3) for r in rows))
"
  rm -rf $O/stft_prof_$rp
done 2>&1 | tee $O/r06_k_stft_run_pairs_kernel_us.txt
