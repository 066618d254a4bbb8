#!/usr/bin/env python3
"""Host-inclusive time of ade_process (page-locked caller buffers, 256 x 1 s) against the number of sub-batches (option "host_split").   python tools/host_split_probe.py [B]"""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from ade_testlib import make_session
from audio_denoiser_onnx_amd.synth import synth_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
x = synth_batch(B)
s = make_session()
s.reserve(B)
pin_in = torch.from_numpy(x.copy()).pin_memory(); pin_out = torch.empty((B, s.row_out), dtype=torch.int16).pin_memory()
p_in, p_out = pin_in.numpy(), pin_out.numpy()
page_out = np.empty((B, s.row_out), np.int16)
ref = None
for rnd in range(3):
    for split in "1", "2", "3", "4", "6", "8":
        s.set_option("host_split", split)
        for name, (a, b) in (("page-locked", (p_in, p_out)), ("pageable", (x, page_out))):
            for _ in range(10):
                s.process_into(a, b)
            t0 = time.perf_counter()
            for _ in range(100):
                s.process_into(a, b)
            ms = (time.perf_counter() - t0) / 100 * 1e3
            if ref is None:
                ref = b.copy()
            print(f"round {rnd} host_split {split} {name:11s}: {ms:.4f} ms/call  same-bits {bool((b == ref).all())}", flush=True)
