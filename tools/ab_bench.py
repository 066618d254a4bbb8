#!/usr/bin/env python3
"""Same-box A/B of two libade builds: alternating launches, device time of the shipped launch sequence (HIP events).

usage: tools/ab_bench.py <libA.so> <libB.so> [batch]      (boxes differ by +-2 %, so builds are compared on ONE box)"""
import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import torch
from audio_denoiser_onnx_amd import _lib
from ade_testlib import make_session
from audio_denoiser_onnx_amd.synth import synth_batch

paths = sys.argv[1:3]
B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
x = torch.from_numpy(synth_batch(B)).cuda()
sess = []
for p in paths:
    s = make_session(_lib.AdeLibrary(os.path.abspath(p)))
    s.reserve(B)
    sess.append(s)
outs = [torch.empty((B, s.out_len), dtype=torch.int16, device='cuda') for s in sess]
for s, o in zip(sess, outs):
    for _ in range(5):
        s.run_device(x, o)
torch.cuda.synchronize()
acc = [[], []]
for rnd in range(8):
    for k, (s, o) in enumerate(zip(sess, outs)):
        s.profile(2)
        t = 0.0
        for _ in range(10):
            s.run_device(x, o)
            t += sum(v['ms'] for v in s.kernel_times().values())
        s.profile(0)
        acc[k].append(t / 10)
for k, p in enumerate(paths):
    a = np.array(acc[k])
    print(f'{p}: median {np.median(a)*1e3:.1f} us  min {a.min()*1e3:.1f}  max {a.max()*1e3:.1f}')
for k, (s, o) in enumerate(zip(sess, outs)):     # the phase-clock build of the same kernel (mode 3), for reference
    try:
        s.profile(3)
        t = []
        for _ in range(10):
            s.run_device(x, o)
            t.append(sum(v['ms'] for v in s.kernel_times().values()))
        s.profile(0)
        print(f'{paths[k]}: clock build median {np.median(t)*1e3:.1f} us')
    except Exception as e:
        print(paths[k], 'mode 3 unavailable:', e)
print('outputs equal:', bool(torch.equal(outs[0], outs[1])), ' max |diff| LSB:', int((outs[0].int() - outs[1].int()).abs().max()))
