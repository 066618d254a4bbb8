O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests/test_melband.py tests/test_melband_dynamic.py -m gpu -x -q > $O/r05_v4_tests.txt 2>&1; echo "tests rc $?"; tail -2 $O/r05_v4_tests.txt
L=audio_denoiser_onnx_amd/libade.so; cp $L /tmp/_keep.so
for r in 1 2; do for V in old new; do cp tools/ab/libade_$V.so $L
  timeout 600 python bench.py --workload melband --dtype f32 --steps 5 --warmup 1 --cpu-seconds 0 --host-steps 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V melband f32', d['ms_per_step'], d['roofline']['frac'])"; done; done | tee $O/r05_att_ab.txt
cp /tmp/_keep.so $L
