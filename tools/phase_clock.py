import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from ade_testlib import make_session
from audio_denoiser_onnx_amd.synth import synth_batch
s = make_session(None)
for B in (1, 256):
    x = synth_batch(B)
    s.process(x); s.profile(True); s.process(x); s.process(x)
    c = s.tap('phase_clock', 64)
    print('B', B, 'gtblock phase ticks (10ns):', c[:9].astype(int).tolist())
    print('B', B, 'dpgrnn  phase ticks (10ns):', c[16:21].astype(int).tolist())
    s.profile(False)
