#!/usr/bin/env python3
"""DFSMN throughput on one MI355X (informational; BASELINE.json has no DFSMN configuration): batch x 2 s chunks @ 48 kHz,
int16 PCM resident in HBM, steps enqueued back to back on one stream."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.chdir(REPO)
import numpy as np
import torch
from audio_denoiser_onnx_amd.metadata import build_audio_metadata
from audio_denoiser_onnx_amd.session import InferenceSession

L = 96000
meta = build_audio_metadata(producer="bench_dfsmn", model_name="DFSMN", task="denoise", model_family="dfsmn", input_audio_length=L,
                            in_sample_rate=48000, nfft=1920, window_length=1920, hop_length=960, window_type="hamming",
                            center_pad=False, pad_mode="constant", feature_kind="kaldi_fbank_stft")
with open(os.path.join("tests", "golden", "dfsmn_seed0.adew"), "rb") as f:
    blob = f.read()
sess = InferenceSession(weights=blob, metadata=meta)
for B in (8, 64, 256):
    x = torch.from_numpy((np.random.default_rng(0).standard_normal((B, L)) * 1500).astype(np.int16)).cuda()
    y = torch.empty((B, sess.out_len), dtype=torch.int16, device="cuda")
    st = torch.cuda.Stream()
    for _ in range(3):
        sess.run_device(x, y, stream=st.cuda_stream)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        sess.run_device(x, y, stream=st.cuda_stream)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    N = B * sess.frames
    macs = N * (120 * 1025 + 256 * 120 + 9 * 2 * 256 * 256 + 961 * 256)          # the mask network's matrix products (the transforms are FFTs: ~0.3 MFLOP per frame)
    print(f"B={B:4d}: {dt*1e3:8.3f} ms/step  {B*2.0/dt:10.0f} audio-s/s  RTF {dt/(B*2.0):.2e}  {2*macs/dt/1e12:6.1f} TFLOP/s fp32 in the mask network's GEMMs")
