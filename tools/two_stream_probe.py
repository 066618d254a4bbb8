#!/usr/bin/env python3
"""Probe: steps alternating over two sessions (own workspace each) on two streams vs one session on one stream."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
from ade_testlib import make_session
from audio_denoiser_onnx_amd.synth import synth_batch
B = 256
x = torch.from_numpy(synth_batch(B)).cuda()
ss = [make_session(None), make_session(None)]
outs = [torch.empty((B, s.out_len), dtype=torch.int16, device='cuda') for s in ss]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
for s, o, st in zip(ss, outs, streams):
    s.reserve(B)
    for _ in range(5):
        s.run_device(x, o, stream=st.cuda_stream)
torch.cuda.synchronize()
def run(nsess, steps=200):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps):
        k = i % nsess
        ss[k].run_device(x, outs[k], stream=streams[k].cuda_stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
for rep in range(3):
    print('1 stream: %.4f ms/step   2 streams: %.4f ms/step' % (run(1), run(2)))
print('equal outputs:', bool(torch.equal(outs[0], outs[1])))
