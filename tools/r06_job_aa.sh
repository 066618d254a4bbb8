# round 6, job aa: run-form STFT, transforms of <= 512 points on ONE wavefront each (in place, no workgroup barrier): tests, kernel durations for pairs-per-workgroup 8 / 16, the bench block
O=$PWD/gpurun_out; R=$PWD; mkdir -p $O
timeout 900 python -m pytest tests/test_stft_process.py tests/test_gpu_parity.py -m gpu -x -q -k "stft" 2>&1 | tail -2
timeout 600 python -m pytest tests/test_ulunas.py tests/test_hgtcrn.py -m gpu -x -q 2>&1 | tail -2
cd /tmp; export TMPDIR=/tmp
for rp in 0 8 16 32; do
  ADE_STFT_RUN_PAIRS=$rp timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/stft_prof_$rp -- python $R/tools/bench_stft.py > $O/r06_aa_stft_bench_rp$rp.txt 2>&1
  f=$(find $O/stft_prof_$rp -name "*kernel_stats.csv" | head -1); cp $f $O/r06_aa_stft_rp${rp}_kernel_stats.csv; rm -rf $O/stft_prof_$rp
  echo "ADE_STFT_RUN_PAIRS=$rp"; python3 -c "
import csv
for r in sorted(csv.DictReader(open('$O/r06_aa_stft_rp${rp}_kernel_stats.csv')), key=lambda r: r['Name']):
    if 'stft_run' in r['Name']: print('  %-40s %-60s calls %4s avg %8.1f us' % (r['Name'].split('(anonymous namespace)::')[-1].split('(')[0][:40], r['Name'].split('(anonymous namespace)::')[1][:0], r['Calls'], float(r['AverageNs'])/1e3))
"
done 2>&1 | tee $O/r06_aa_stft_kernel_us.txt
