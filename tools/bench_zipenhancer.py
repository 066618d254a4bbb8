#!/usr/bin/env python3
"""ZipEnhancer throughput on one MI355X (random-init weights of the architecture, synthetic PCM resident in HBM).

    python tools/bench_zipenhancer.py [--batches 1,16,128] [--steps 5] [--length 16000]

One chunk = 1 s (16000 samples, 161 frames x 101 sub-bands): BASELINE.json configs[2].  Reports ms/step, audio-seconds per second, the
real-time factor and fp32 matrix TFLOP/s (audio_denoiser_onnx_amd.zipenhancer.macs_per_window).
"""
import argparse
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.chdir(REPO)

import torch  # noqa: E402

from audio_denoiser_onnx_amd import zipenhancer as zp  # noqa: E402
from audio_denoiser_onnx_amd.session import InferenceSession  # noqa: E402
from audio_denoiser_onnx_amd.synth import synth_batch  # noqa: E402
from audio_denoiser_onnx_amd.weights import pack_blob  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="1,16,128")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--length", type=int, default=16000)
    a = ap.parse_args()
    cfg = zp.ZipConfig()
    sess = InferenceSession(weights=pack_blob(zp.synthetic_tensors(cfg)), metadata=zp.metadata(a.length))
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    fl = 2.0 * zp.macs_per_window(sess.frames, cfg)["total"]
    for B in [int(x) for x in a.batches.split(",")]:
        pcm = torch.from_numpy(synth_batch(B, a.length)).to(dev)
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
        secs = B * a.length / 16000.0
        print(f"B={B:4d}: {ms:9.3f} ms/step  {secs / (ms * 1e-3):9.1f} audio-s/s  RTF {ms * 1e-3 / secs:.2e}  {B * fl / (ms * 1e-3) / 1e12:6.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
