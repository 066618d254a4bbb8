#!/usr/bin/env python3
"""Generic STFT_Process operator on one MI355X (informational): analysis + synthesis of batch x length audio for the starred folders' transform sizes
(FFT formulation) and, beside each, a neighbouring size with a prime factor above 5, which takes the dense-table MFMA formulation the operator used for
every size before -- the same amount of audio through both."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.chdir(REPO)
import torch
from audio_denoiser_onnx_amd.stft_process import STFT_Process

CASES = [("gtcrn", 512, 256, 16000, 256), ("gtcrn-dense-neighbour", 514, 257, 16000, 256),          # 514 = 2 * 257
         ("zipenhancer", 400, 100, 16000, 256), ("zip-dense-neighbour", 402, 100, 16000, 256),                 # 402 = 2 * 3 * 67
         ("melband", 2048, 441, 66150, 64), ("melband-dense-neighbour", 2044, 441, 66150, 64),                 # 2044 = 4 * 7 * 73
         ("dfsmn", 1920, 960, 96000, 64), ("dfsmn-dense-neighbour", 1918, 959, 96000, 64)]                     # 1918 = 2 * 7 * 137
for name, n_fft, hop, L, B in CASES:
    fwd = STFT_Process("stft_B", n_fft, n_fft, hop, 0, "hann", True, "reflect")
    T = fwd.frames(L)
    inv = STFT_Process("istft_B", n_fft, n_fft, hop, T, "hann", True, "reflect")
    x = torch.randn(B, 1, L, device="cuda") * 0.1
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        y = inv(fwd(x, stream=st), stream=st)
    torch.cuda.synchronize()
    n = 20
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        s = fwd(x, stream=st)
    torch.cuda.synchronize(); ta = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n):
        y = inv(s, stream=st)
    torch.cuda.synchronize(); ts = (time.perf_counter() - t0) / n
    err = float((y.reshape(B, -1) - x.reshape(B, -1)[:, :y.numel() // B]).abs().max())
    sec = B * L / {512: 16000, 514: 16000, 400: 16000, 402: 16000, 2048: 44100, 2044: 44100, 1920: 48000, 1918: 48000}[n_fft]
    print(f"{name:26s} n_fft {n_fft:5d} B {B:4d} T {T:4d}: analysis {ta*1e3:7.3f} ms  synthesis {ts*1e3:7.3f} ms  ({sec/(ta+ts):11.0f} audio-s/s round trip, max |x - istft(stft(x))| {err:.1e})")
