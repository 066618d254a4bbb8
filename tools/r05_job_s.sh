mkdir -p tests/unit/_build gpurun_out
for W in 3 4; do
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -Wno-unused-value -DADE_GEMM16_WGS=$W -I include tests/unit/gemm16_unit.hip -o tests/unit/_build/gemm16_unit_gpu || exit 1
echo "WGS=$W: $(tests/unit/_build/gemm16_unit_gpu 1000 1544 384 777 384 1536 130 70 72 64 25633 64 4096 512 512 | grep -c OK) OK"
tests/unit/_build/gemm16_unit_gpu -t 1537920 1544 384 1537920 1536 384 1537920 384 1536 1537920 384 512
done | tee gpurun_out/r05_s_gemm16_variants.txt
