# round 6, job s: counters of the ZipEnhancer bf16 step after the attention core's second form
O=gpurun_out; mkdir -p $O
timeout 1200 bash tools/pmc_workload.sh gpurun_out/r06_s_pmc --workload zipenhancer --dtype bf16 > $O/r06_s_zip_bf16_pmc_summary.txt 2>&1
rm -rf $O/r06_s_pmc/p1 $O/r06_s_pmc/p2 $O/r06_s_pmc/p3 $O/r06_s_pmc/p4 $O/r06_s_pmc/p5
cut -c1-230 $O/r06_s_zip_bf16_pmc_summary.txt | head -45
