#!/usr/bin/env python3
"""Same-box A/B of the workgroup de-phasing knob (ade_set_option "stagger_us"): ms/step of the single-launch GTCRN kernel.

    python tools/stagger_probe.py [us,us,...] [batch]
"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
os.chdir(REPO)
import numpy as np
import torch
from ade_testlib import make_session
from audio_denoiser_onnx_amd.synth import synth_batch

B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
sess = make_session(None, seed=0)
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
pcm = torch.from_numpy(synth_batch(B)).to(dev)
out = torch.empty((B, sess.out_len), dtype=torch.int16, device=dev)
sess.reserve(B)
vals = [float(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,15,20,25,27.5,30,35,40,0".split(","))]

ref = None
for rep in range(2):
    for v in vals:
        sess.set_option("stagger_us", str(v))
        with torch.cuda.stream(stream):
            for _ in range(10):
                sess.run_device(pcm, out, stream=stream.cuda_stream)
            stream.synchronize()
            t = time.perf_counter()
            for _ in range(200):
                sess.run_device(pcm, out, stream=stream.cuda_stream)
            stream.synchronize()
            ms = (time.perf_counter() - t) / 200 * 1e3
        if ref is None:
            ref = out.clone()
        assert torch.equal(out, ref)
        print(f"B={B} stagger {v:5.1f} us: {ms:.4f} ms/step", flush=True)
