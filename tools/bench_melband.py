#!/usr/bin/env python3
"""Mel-Band-Roformer throughput on one MI355X (random-init weights of the architecture, synthetic stereo PCM resident in HBM).

    python tools/bench_melband.py [--depth 6] [--batches 1,4,16] [--steps 5]

One clip = one 1.5 s batch-fold window (66150 samples, 151 frames; Export_MelBandRoformer.py:47-51).  Reports ms/step,
audio-seconds per second, real-time factor and fp32 matrix TFLOP/s (audio_denoiser_onnx_amd.melband.flops_per_clip).
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

from audio_denoiser_onnx_amd import melband, weightgen  # noqa: E402
from audio_denoiser_onnx_amd.session import InferenceSession  # noqa: E402
from audio_denoiser_onnx_amd.weights import pack_blob  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--depth", type=int, default=6)
    ap.add_argument("--batches", default="1,4,16")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--window", type=int, default=66150)
    a = ap.parse_args()
    t0 = time.time()
    w = weightgen.materialise(melband.synthetic_spec(a.depth))
    sess = InferenceSession(weights=pack_blob(melband.model_tensors(w)), metadata=melband.metadata(a.window))
    del w
    print(f"model built in {time.time() - t0:.1f} s: depth {a.depth}, {sess.frames} frames per clip", flush=True)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    fl = melband.flops_per_clip(sess.frames, a.depth)
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
        secs = B * a.window / 44100.0
        print(f"B={B:4d}: {ms:9.3f} ms/step  {secs / (ms * 1e-3):9.1f} audio-s/s  RTF {ms * 1e-3 / secs:.2e}  {B * fl / (ms * 1e-3) / 1e12:6.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
