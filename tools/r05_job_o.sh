O=gpurun_out; mkdir -p $O; R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQC\?_[A-Z_]*\(ICACHE\|IFETCH\|INST_LEVEL\|WAIT_IFETCH\|DCACHE\)[A-Z_]*" | sort -u | tr '\n' ' ' > $R/$O/r05_o_avail.txt; cat $R/$O/r05_o_avail.txt; echo
for SET in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAVES" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_WAIT_INST_ANY"; do
  N=$(echo $SET | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $SET -f csv -d /tmp/pmc_o_$N -- python $R/bench.py --steps 3 --warmup 2 --cpu-seconds 0 --host-steps 0 --other-steps 0 > /tmp/pmc_o_$N.log 2>&1 || tail -3 /tmp/pmc_o_$N.log
  python - /tmp/pmc_o_$N <<'PY' | tee -a $R/$O/r05_o_icache.txt
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+'/*/*counter_collection.csv')
if not f: print('no csv', sys.argv[1]); sys.exit()
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
for r in csv.DictReader(open(f[0])):
    k=r['Kernel_Name'][:60]
    if 'gtcrn_chunk' not in k: continue
    agg[k][r['Counter_Name']]+=float(r['Counter_Value']); n[k].add(r['Dispatch_Id'])
for k in agg:
    print(k, len(n[k]), {c: round(v/len(n[k])) for c,v in agg[k].items()})
PY
done
