#!/usr/bin/env python3
"""Two engine instances (two workspaces, two sets of streams) fed alternately through ade_submit / ade_wait: does one launch's tail (the last ~60 us of a chunk-kernel launch run
at a quarter of the occupancy) overlap the next launch's head?   python tools/two_engine_probe.py [steps]
Also the device-resident form: K launches alternating between the two engines' streams against K launches on one."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import torch
from ade_testlib import make_session
from audio_denoiser_onnx_amd.synth import synth_batch

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
B, depth = 256, 3
ss = [make_session(), make_session()]
x = synth_batch(B)
for s in ss:
    s.reserve(B)
    s.set_option("pipe_depth", str(depth))
d_in = torch.from_numpy(x).cuda()
d_outs = [torch.empty((B, ss[0].row_out), dtype=torch.int16, device="cuda") for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
for e in range(2):
    for _ in range(200):
        ss[e].run_device(d_in, d_outs[e], stream=streams[e].cuda_stream)
torch.cuda.synchronize()


def resident(n, engines):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(n):
        e = k % engines
        ss[e].run_device(d_in, d_outs[e], stream=streams[e].cuda_stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for eng in (1, 2, 1, 2):
    print(f"device-resident, {eng} engine(s) / stream(s): {resident(steps, eng):.4f} ms per 256 x 1 s batch", flush=True)

ring_in = [[torch.from_numpy(np.roll(x, k + 7 * e, axis=0).copy()).pin_memory() for k in range(depth)] for e in range(2)]
ring_out = [[torch.empty((B, ss[0].row_out), dtype=torch.int16).pin_memory() for _ in range(depth)] for e in range(2)]


def pipelined(n, engines):
    t = []
    for k in range(n):
        e = k % engines
        if len(t) >= depth * engines:
            ee, q = t.pop(0)
            ss[ee].wait(q)
        t.append((e, ss[e].submit(ring_in[e][(k // engines) % depth].numpy(), ring_out[e][(k // engines) % depth].numpy())))
    for ee, q in t:
        ss[ee].wait(q)


for eng in (1, 2, 1, 2):
    pipelined(12, eng)
    t0 = time.perf_counter(); pipelined(steps, eng); dt = (time.perf_counter() - t0) / steps * 1e3
    print(f"host buffers, ade_submit / ade_wait, {eng} engine(s): {dt:.4f} ms per 256 x 1 s batch", flush=True)
# same bits from both engines
a, _ = ss[0].process(x[:8]); b, _ = ss[1].process(x[:8])
print("engines agree:", bool(np.array_equal(a, b)))
