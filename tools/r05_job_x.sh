L=audio_denoiser_onnx_amd/libade.so; cp $L /tmp/_keep.so
for r in 1 2; do for V in keep wpe6 wpe8; do
  if [ $V = keep ]; then cp /tmp/_keep.so $L; else cp tools/ab/libade_$V.so $L; fi
  echo "$V: $(timeout 300 python tools/bench_hgtcrn.py --batches 256 --steps 20 2>&1 | tail -1)"
done; done
cp /tmp/_keep.so $L
