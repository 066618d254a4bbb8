#!/usr/bin/env python3
"""ade_process on page-locked buffers with the batch streamed through one launch (option "host_stream" = row groups) against the sub-batch path (host_stream = 1) and the
device-resident kernel: bit-equality on inputs that CHANGE every call (a stale line of the previous call's PCM would show), then host-inclusive time per call.
python tools/host_stream_probe.py [B] [extra-streams]      extra-streams: streams created first, to move the engine's own onto other hardware queues"""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from ade_testlib import make_session
from audio_denoiser_onnx_amd.synth import synth_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
extra = [torch.cuda.Stream() for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 0)]
s = make_session()
s.reserve(B)
pin_in = torch.empty((B, s.row_in), dtype=torch.int16).pin_memory(); pin_out = torch.empty((B, s.row_out), dtype=torch.int16).pin_memory()
p_in, p_out = pin_in.numpy(), pin_out.numpy()
d_out = torch.empty((B, s.row_out), dtype=torch.int16, device="cuda")
bad = 0
for split in ("4", "8", "2"):
    s.set_option("host_stream", split)
    for it in range(12):
        x = synth_batch(B, first_index=1000 * it + 7)
        p_in[:] = x
        p_out[:] = 0
        s.process_into(p_in, p_out)
        s.run_device(torch.from_numpy(x).cuda(), d_out)
        ref = d_out.cpu().numpy()
        same = bool((p_out == ref).all())
        bad += not same
        if not same:
            print(f"host_stream {split} call {it}: MISMATCH in {int((p_out != ref).any(axis=1).sum())} rows", flush=True)
print("bit-equality over 36 calls on changing inputs:", "OK" if bad == 0 else f"{bad} FAILED", flush=True)
pageable_in, pageable_out = synth_batch(B), np.empty((B, s.row_out), np.int16)
for rnd in range(2):
    for split in ("1", "2", "4", "8"):
        s.set_option("host_stream", split)
        for name, (a, b) in (("page-locked", (p_in, p_out)), ("pageable", (pageable_in, pageable_out))):
            for _ in range(10):
                s.process_into(a, b)
            t0 = time.perf_counter()
            for _ in range(100):
                s.process_into(a, b)
            print(f"round {rnd} host_stream {split} {name:11s}: {(time.perf_counter() - t0) / 100 * 1e3:.4f} ms/call", flush=True)
sys.exit(1 if bad else 0)
