O=gpurun_out; mkdir -p $O
hipcc --offload-arch=gfx950 -O2 tools/ubench/hwid_probe.hip -o /tmp/hwid_probe && /tmp/hwid_probe 4 > $O/r05_a_hwid_probe_grid4.txt 2>&1
for B in 1 64 128 256; do timeout 300 python tools/phase_latency.py $B > $O/r05_a_phase_latency_B$B.txt 2>&1; done
bash tools/clock_sample.sh $O/r05_a_clock_gtcrn.txt timeout 300 python bench.py --steps 20000 --warmup 100 --cpu-seconds 0 --other-steps 0 --host-steps 0
bash tools/clock_sample.sh $O/r05_a_clock_melband_bf16.txt timeout 300 python bench.py --workload melband --dtype bf16 --steps 20 --warmup 2 --cpu-seconds 0 --host-steps 0 --no-deviation
bash tools/clock_sample.sh $O/r05_a_clock_melband_f32.txt timeout 300 python bench.py --workload melband --dtype f32 --steps 5 --warmup 1 --cpu-seconds 0 --host-steps 0
ls /sys/class/drm/ > $O/r05_a_sysfs.txt; ls /sys/class/drm/card*/device/ >> $O/r05_a_sysfs.txt 2>&1; rocm-smi --showclocks --showpower >> $O/r05_a_sysfs.txt 2>&1
head -30 $O/r05_a_clock_gtcrn.txt; tail -3 $O/r05_a_clock_gtcrn.txt.cmd | cut -c1-300
