# round 6, job ai: counters of the ZipEnhancer step (bf16 and f32) on the final tree
O=gpurun_out; mkdir -p $O
for D in bf16 f32; do
timeout 1200 bash tools/pmc_workload.sh gpurun_out/r06_ai_pmc_$D --workload zipenhancer --dtype $D > $O/r06_z_zipenhancer_${D}_pmc_summary.txt 2>&1
rm -rf $O/r06_ai_pmc_$D
done
cut -c1-200 $O/r06_z_zipenhancer_bf16_pmc_summary.txt | head -24
