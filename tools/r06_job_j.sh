# round 6, job j: kernel-level durations of the STFT operator, both forms (rocprofv3 --kernel-trace --stats)
O=$PWD/gpurun_out; R=$PWD; cd /tmp; export TMPDIR=/tmp
for form in 1 0; do
  ADE_STFT_RUN=$form timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/stft_prof_$form -- python $R/tools/bench_stft.py > /dev/null 2>&1
  find $O/stft_prof_$form -name "*kernel_stats.csv" -exec cp {} $O/r06_j_stft_form${form}_kernel_stats.csv \;
  rm -rf $O/stft_prof_$form
  echo "ADE_STFT_RUN=$form"; python3 -c "
import csv
for r in csv.DictReader(open('$O/r06_j_stft_form${form}_kernel_stats.csv')):
    if 'stft' in r['Name']: print('  %-70s calls %4s avg %8.1f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3))
"
done
