#!/usr/bin/env python3
"""Print the in-kernel phase clocks (wall_clock64 ticks, 10 ns) of workgroup 0 of each fused stage kernel."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from ade_testlib import make_session
from audio_denoiser_onnx_amd.synth import synth_batch
s = make_session(None)
for B in (1, 256):
    x = synth_batch(B)
    s.process(x); s.profile(1); s.process(x); s.process(x)
    c = s.tap('phase_clock', 64).astype(int)
    print('B', B, 'gtblock[pw1,dw,h1,energy+GI,-,GRU,at,out]:', (c[1:9] - c[0:8]).tolist(), 'total', c[8])
    print('B', B, 'dpgrnn [intra,fcln,inter,fcln]       :', (c[17:21] - c[16:20]).tolist(), 'total', c[20])
    print('B', B, 'front  [mean | last tile: stft..conv0, conv0..conv1, conv1]:', c[33] - c[32], (c[35:37] - c[34:36]).tolist(), 'total', c[36])
    print('B', B, 'back   [last tile: stage+deconv3, +e0, deconv4, istft] :', (c[50:53] - c[49:52]).tolist(), 'total', c[53])
    print('B', B, 'front acc over tiles [stft+feat, conv0, conv1]:', c[40:43].tolist())
    print('B', B, 'back  acc over tiles [stageS, deconv3, +e0, deconv4, mask+irfft+ola, finalize+carry]:', c[56:62].tolist())
    s.profile(0)
