#!/usr/bin/env python3
"""Phase clocks of the fused stage kernels at a given chunk length (to see how the frame count T shapes each phase)."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from ade_testlib import make_session
for L in [int(a) for a in sys.argv[1:]] or [15700, 16000, 16200]:
    s = make_session(None, seed=0, length=L)
    x = (np.random.default_rng(0).standard_normal((256, L)) * 3000).astype(np.int16)
    s.process(x); s.profile(1); s.process(x); s.process(x)
    c = s.tap('phase_clock', 64).astype(int)
    print('L', L, 'T', s.frames)
    print('  gtblock[pw1,dw,h1,energy+GI,-,GRU,at,out]:', (c[1:9] - c[0:8]).tolist(), 'total', c[8])
    print('  dpgrnn [intra,fcln,inter,fcln]       :', (c[17:21] - c[16:20]).tolist(), 'total', c[20])
    print('  front total', c[36], ' back total', c[53])
    kt = s.kernel_times()
    print('  stage ms:', {k: round(v['ms'], 4) for k, v in kt.items() if v['launches']})
    s.profile(0)
