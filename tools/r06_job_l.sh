# round 6, job l: run-form STFT operator with its default run lengths: kernel durations, the round-5 pair form beside them, and the bench block
O=$PWD/gpurun_out; R=$PWD; cd /tmp; export TMPDIR=/tmp
for form in 1 0; do
  ADE_STFT_RUN=$form timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/stft_prof_$form -- python $R/tools/bench_stft.py > $O/r06_l_stft_bench_form$form.txt 2>&1
  f=$(find $O/stft_prof_$form -name "*kernel_stats.csv" | head -1); cp $f $O/r06_l_stft_form${form}_kernel_stats.csv; rm -rf $O/stft_prof_$form
  echo "ADE_STFT_RUN=$form"; python3 -c "
import csv
for r in sorted(csv.DictReader(open('$O/r06_l_stft_form${form}_kernel_stats.csv')), key=lambda r: r['Name']):
    if 'stft' in r['Name']: print('  %-64s calls %4s avg %8.1f us' % (r['Name'].split('(anonymous namespace)::')[-1].split('(')[0][:64], r['Calls'], float(r['AverageNs'])/1e3))
"
done 2>&1 | tee $O/r06_l_stft_kernel_us.txt
cd $R; python - <<'PY'
import json, sys, torch
sys.argv = ["bench.py"]
import bench
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
d = bench.stft_operator_lines(st.cuda_stream)
json.dump(d, open("gpurun_out/r06_l_stft_operator_bench_block.json", "w"), indent=1)
for k in ("gtcrn_512_256", "melband_2048_441"):
    print(k, "analysis", d[k]["analysis"]["us"], "us frac", d[k]["analysis"]["frac"], "| synthesis", d[k]["synthesis"]["us"], "us frac", d[k]["synthesis"]["frac"])
PY
