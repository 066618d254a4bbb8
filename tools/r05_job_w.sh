O=gpurun_out; mkdir -p $O; R=$PWD
bash tools/abn_libs.sh 3 tools/ab/libade_pad0.so tools/ab/libade_pad1.so tools/ab/libade_pad2.so tools/ab/libade_pad4.so tools/ab/libade_pad8.so | tee $O/r05_w_plane_pad_ab.txt
L=audio_denoiser_onnx_amd/libade.so; cp $L /tmp/_keep2.so
for P in 0 1 2 4 8; do
  cp tools/ab/libade_pad$P.so $L
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES -f csv -d /tmp/pp$P -- python $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --host-steps 0 --other-steps 0 > /dev/null 2>&1)
  python - /tmp/pp$P $P <<'PY' | tee -a $O/r05_w_plane_pad_ab.txt
import csv,glob,sys,collections,re
f=glob.glob(sys.argv[1]+'/*/*counter_collection.csv')[0]
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    m=re.search(r'(k_gtcrn_chunk|k_front|k_gtblock|k_dpgrnn|k_back)', r['Kernel_Name'])
    if m: agg[m.group(1)][r['Counter_Name']]+=float(r['Counter_Value'])
print('pad', sys.argv[2], ' '.join(f"{k}: conf/active {100*v['SQ_LDS_BANK_CONFLICT']/max(1,v['SQ_LDS_IDX_ACTIVE']):.1f}%" for k,v in agg.items()))
PY
done
cp /tmp/_keep2.so $L
