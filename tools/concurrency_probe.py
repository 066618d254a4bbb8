#!/usr/bin/env python3
"""Do launches of the chunk kernel on different streams run side by side?   python tools/concurrency_probe.py
N sessions (own workspace each) x B chunks, one launch each per round on N streams (torch streams, or the engine's own); ms per ROUND."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
from ade_testlib import make_session
from audio_denoiser_onnx_amd.synth import synth_batch

def probe(N, B, rounds=200):
    x = torch.from_numpy(synth_batch(B)).cuda()
    ss = [make_session(None) for _ in range(N)]
    outs = [torch.empty((B, s.out_len), dtype=torch.int16, device='cuda') for s in ss]
    streams = [torch.cuda.Stream() for _ in range(N)]
    for s, o, st in zip(ss, outs, streams):
        s.reserve(B)
        for _ in range(5):
            s.run_device(x, o, stream=st.cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(rounds):
        for s, o, st in zip(ss, outs, streams):
            s.run_device(x, o, stream=st.cuda_stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / rounds * 1e3

for rep in range(2):
    for N, B in ((1, 256), (1, 128), (2, 128), (1, 64), (2, 64), (4, 64), (1, 32), (8, 32)):
        print(f"rep {rep}: {N} stream(s) x {B} chunks: {probe(N, B):.4f} ms per round", flush=True)
