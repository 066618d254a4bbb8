#!/usr/bin/env python3
"""The pipelined host entry (ade_submit / ade_wait) on 256 x 1 s batches: wall clock per batch at a given depth.  Run under `rocprofv3 --kernel-trace --memory-copy-trace` to get the
timeline tools/pipeline_timeline.py reads.     python tools/pipeline_probe.py [depth] [steps]"""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import torch
from ade_testlib import make_session
from audio_denoiser_onnx_amd.synth import synth_batch

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 3
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
B = 256
s = make_session()
s.reserve(B)
s.set_option("pipe_depth", str(depth))
x = synth_batch(B)
ring_in = [torch.from_numpy(np.roll(x, k, axis=0).copy()).pin_memory() for k in range(depth)]
ring_out = [torch.empty((B, s.row_out), dtype=torch.int16).pin_memory() for _ in range(depth)]
d_in = torch.from_numpy(x).cuda(); d_out = torch.empty((B, s.row_out), dtype=torch.int16, device="cuda")
for _ in range(200):
    s.run_device(d_in, d_out)
def loop(n):
    t = []
    for k in range(n):
        if len(t) >= depth:
            s.wait(t.pop(0))
        t.append(s.submit(ring_in[k % depth].numpy(), ring_out[k % depth].numpy()))
    for q in t:
        s.wait(q)
loop(8)
t0 = time.perf_counter(); loop(steps); dt = (time.perf_counter() - t0) / steps * 1e3
print(f"depth {depth}: {dt:.4f} ms per 256 x 1 s batch over {steps} back-to-back submissions")
