#!/usr/bin/env python3
"""Print the in-kernel phase clocks (wall_clock64 ticks, 10 ns) of workgroup 0.

mode 1: one kernel per stage (first encoder GTConvBlock / first DPGRNN only);
mode 3: the single-launch kernel's clock build -- every stage of the shipped launch, as it runs back to back."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from ade_testlib import make_session
from audio_denoiser_onnx_amd.synth import synth_batch

GT = '[pw1,dw,h1,energy+GI,-,GRU,at,out]'
import os
from audio_denoiser_onnx_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
s = make_session(_lib.AdeLibrary(os.path.abspath(sys.argv[2])) if len(sys.argv) > 2 else None)   # optional: another build to look at
x = synth_batch(B)
s.process(x); s.profile(1); s.process(x); s.process(x)
c = s.tap('phase_clock', 64).astype(int)
print('per-stage kernels, B =', B)
print('  gtblock' + GT + ':', (c[1:9] - c[0:8]).tolist(), 'total', c[8])
print('  dpgrnn [intra,fcln,inter,fcln]       :', (c[17:21] - c[16:20]).tolist(), 'total', c[20])
print('  front  total', c[36], ' acc over tiles [stft+feat, conv0, conv1]:', c[40:43].tolist(), ' mean', c[33] - c[32])
print('  back   total', c[53], ' acc over tiles [top, deconv3, s-issue+deconv4, mask+irfft+ola, commit+finalize, carry]:', c[56:62].tolist())
s.profile(3); s.process(x); s.process(x)
c = s.tap('phase_clock', 640).astype(int).reshape(10, 64)
s.profile(0)
print('single launch (clock build), B =', B)
f = c[0]
print('  front  total', f[36], ' acc [stft+feat, conv0, conv1]:', f[40:43].tolist(), ' mean', f[33] - f[32])
for name, rows in (('enc', (1, 2, 3)), ('dec', (6, 7, 8))):
    for i, r in enumerate(rows):
        g = c[r]
        print(f'  {name}{i} gtblock' + GT + ':', (g[1:9] - g[0:8]).tolist(), 'total', g[8])
for i, r in enumerate((4, 5)):
    d = c[r]
    print(f'  dp{i} [intra,fcln,inter,fcln]:', (d[17:21] - d[16:20]).tolist(), 'total', d[20] - d[16])
b = c[9]
print('  back   total', b[53], ' acc [top, deconv3, s-issue+deconv4, mask+irfft+ola, commit+finalize, carry]:', b[56:62].tolist())
