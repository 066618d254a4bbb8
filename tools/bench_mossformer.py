#!/usr/bin/env python3
"""MossFormer2-SS-16K throughput on one MI355X (random-init weights of the architecture, synthetic PCM resident in HBM).

    python tools/bench_mossformer.py [--layers 24] [--batches 1,8,32] [--steps 3] [--window 24000]

One row = one 1.5 s batch-fold window (24000 samples, 2999 frames; Export_MossFormer2_SS_16K.py:44-47).  Reports ms/step,
audio-seconds per second, real-time factor and fp32 matrix TFLOP/s (audio_denoiser_onnx_amd.mossformer.flops_per_window).
"""
import argparse
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.chdir(REPO)

import torch  # noqa: E402

from audio_denoiser_onnx_amd import mossformer  # noqa: E402
from audio_denoiser_onnx_amd.session import InferenceSession  # noqa: E402
from audio_denoiser_onnx_amd.weights import pack_blob  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=24)
    ap.add_argument("--batches", default="1,8,32")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--window", type=int, default=24000)
    a = ap.parse_args()
    t0 = time.time()
    frames = mossformer.frames_of(a.window)
    fused = {n: mossformer.synthetic_tensor(n, s, sc, frames) for n, s, sc in mossformer.synthetic_spec(a.layers)}
    scalars = dict(mossformer.DEFAULT_SCALARS, fs_front_alpha=[0.25] * a.layers)
    sess = InferenceSession(weights=pack_blob(mossformer.model_tensors(fused, scalars, a.window)), metadata=mossformer.metadata(a.window))
    del fused
    print(f"model built in {time.time() - t0:.1f} s: {a.layers} layers, {sess.frames} frames per window", flush=True)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    fl = mossformer.flops_per_window(sess.frames, a.layers)
    for B in [int(x) for x in a.batches.split(",")]:
        rng = np.random.default_rng(B)
        pcm = torch.from_numpy((rng.standard_normal((B, sess.row_in)) * 3000).astype(np.int16)).to(dev)
        out = torch.empty((B, sess.row_out), dtype=torch.int16, device=dev)
        sess.reserve(B)
        with torch.cuda.stream(stream):
            sess.run_device(pcm, out, stream=stream.cuda_stream)
            stream.synchronize()
            t = time.perf_counter()
            for _ in range(a.steps):
                sess.run_device(pcm, out, stream=stream.cuda_stream)
            stream.synchronize()
            ms = (time.perf_counter() - t) / a.steps * 1e3
        secs = B * a.window / 16000.0
        print(f"B={B:4d}: {ms:9.3f} ms/step  {secs / (ms * 1e-3):9.1f} audio-s/s  RTF {ms * 1e-3 / secs:.2e}  {B * fl / (ms * 1e-3) / 1e12:6.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
