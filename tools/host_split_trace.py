#!/usr/bin/env python3
"""A few ade_process calls with host_split = N on page-locked buffers, for a rocprofv3 --kernel-trace --memory-copy-trace timeline.   python tools/host_split_trace.py N"""
import os, sys
os.chdir(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
from ade_testlib import make_session
from audio_denoiser_onnx_amd.synth import synth_batch
B = 256
x = synth_batch(B)
s = make_session(); s.reserve(B)
pin_in = torch.from_numpy(x.copy()).pin_memory(); pin_out = torch.empty((B, s.row_out), dtype=torch.int16).pin_memory()
s.set_option("host_split", sys.argv[1])
for _ in range(8):
    s.process_into(pin_in.numpy(), pin_out.numpy())
