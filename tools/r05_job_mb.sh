L=audio_denoiser_onnx_amd/libade.so; cp $L /tmp/_keep.so
cp tools/ab/libade_I.so $L; python tools/debug_melband_rotary.py /tmp/refI.npy 2>&1 | grep -E "^run"
cp tools/ab/libade_old.so $L; python tools/debug_melband_rotary.py /tmp/refI.npy 2>&1 | grep -E "^run|saved"
for r in 1 2 3; do for V in old I; do cp tools/ab/libade_$V.so $L
  timeout 600 python bench.py --workload melband --dtype bf16 --steps 10 --warmup 2 --cpu-seconds 0 --host-steps 0 --no-deviation 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V', d['ms_per_step'])"; done; done
cp /tmp/_keep.so $L
