#!/usr/bin/env python3
"""Same-box timing of the fused GTCRN path under engine options.   python tools/opt_probe.py B steps key=value[,key=value] [key=value ...]

Every argument after `steps` is one configuration (comma-separated options; `-` = defaults); the configurations are timed round-robin, three rounds."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
from ade_testlib import make_session
from audio_denoiser_onnx_amd.synth import synth_batch

B, steps = int(sys.argv[1]), int(sys.argv[2])
cfgs = sys.argv[3:] or ['-']
x = synth_batch(B)
d_in = torch.from_numpy(x).cuda()
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
sessions = []
for c in cfgs:
    s = make_session()
    if c != '-':
        for kv in c.split(','):
            k, v = kv.split('=')
            s.set_option(k, v)
    s.reserve(B)
    sessions.append(s)
d_out = torch.empty((B, sessions[0].row_out), dtype=torch.int16, device="cuda")
ref = None
for rnd in range(3):
    for c, s in zip(cfgs, sessions):
        for _ in range(30):
            s.run_device(d_in, d_out, stream=stream.cuda_stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            s.run_device(d_in, d_out, stream=stream.cuda_stream)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        o = d_out.cpu().numpy().copy()
        if ref is None:
            ref = o
        print(f"round {rnd} [{c}]: {ms:.4f} ms/step  same-bits {bool((o == ref).all())}", flush=True)
