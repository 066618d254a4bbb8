mkdir -p tests/unit/_build gpurun_out
for V in "-DADE_GEMM_KSUB=1" "-DADE_GEMM_KSUB=2" "-DADE_GEMM_DOUBLE=true"; do
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -Wno-unused-value $V -I include tests/unit/gemm32_unit.hip -o tests/unit/_build/gemm32_unit_gpu || exit 1
echo "$V: $(tests/unit/_build/gemm32_unit_gpu 130 70 37 129 200 64 1000 333 250 257 640 1029 | grep -c OK) OK"
tests/unit/_build/gemm32_unit_gpu -t 1537920 1536 384 1537920 384 1536 511936 2176 512 511936 512 1024
done | tee gpurun_out/r05_r_gemm32_variants.txt
