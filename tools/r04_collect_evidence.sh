#!/bin/bash
# Copy what tools/r04_final_evidence.sh <tag> left under gpurun_out/ into profiles/ (and under the names bench.py looks its committed traffic / matrix-busy figures up by).
TAG=${1:-r04_z}; O=gpurun_out; P=profiles
cp $O/${TAG}_gpu_tests.txt $O/${TAG}_gtcrn_bench.json $O/${TAG}_gtcrn_kernel_stats.csv $O/${TAG}_gtcrn_pmc_summary.txt $P/
cp $O/${TAG}_traffic_pmc.json $P/traffic_pmc.json
for S in zipenhancer_f32 melband_f32 melband_bf16 mossformer_f32; do
  cp $O/${TAG}_${S}_bench.json $O/${TAG}_${S}_kernel_stats.csv $O/${TAG}_${S}_mfma_busy.txt $P/
  W=${S%_*}; D=${S##*_}
  cp $O/${TAG}_${S}_traffic.json $P/r04_${W}_${D}_traffic.json
  if [ "$D" = "f32" ]; then cp $O/${TAG}_${S}_mfma_busy.json $P/r04_${W}_mfma_busy.json; else cp $O/${TAG}_${S}_mfma_busy.json $P/r04_${W}_${D}_mfma_busy.json; fi
done
ls $P | grep -c "r04"
