#!/usr/bin/env python3
"""H-GTCRN throughput on one MI355X (seeded reference-architecture weights from the golden fixture, synthetic stereo PCM resident in HBM).

    python tools/bench_hgtcrn.py [--batches 16,64,256] [--length 32000] [--steps 10]
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

from audio_denoiser_onnx_amd import hgtcrn  # noqa: E402
from audio_denoiser_onnx_amd.session import InferenceSession  # noqa: E402
from audio_denoiser_onnx_amd.weights import pack_blob  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="16,64,256")
    ap.add_argument("--length", type=int, default=32000)          # the reference's default export length (2 s, 126 frames)
    ap.add_argument("--steps", type=int, default=10)
    a = ap.parse_args()
    z = np.load(os.path.join(REPO, "tests", "golden", "hgtcrn_seed0.npz"))
    fused = hgtcrn.fold_state_dict({str(k): z["w:" + str(k)] for k in z["keys"]})
    sess = InferenceSession(weights=pack_blob(fused), metadata=hgtcrn.metadata(a.length))
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    secs = a.length / 16000.0
    for B in [int(x) for x in a.batches.split(",")]:
        rng = np.random.default_rng(B)
        src = rng.standard_normal((B, 1, sess.in_len + 7)) * 3000
        pcm = np.concatenate((src[:, :, 7:], 0.7 * src[:, :, :-7]), axis=1) + rng.standard_normal((B, 2, sess.in_len)) * 300   # second microphone: delayed copy + noise
        pcm = torch.from_numpy(np.clip(pcm, -32768, 32767).astype(np.int16).reshape(B, -1)).to(dev)
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
        print(f"B={B:5d} x {secs:g} s stereo: {ms:9.3f} ms/step  {B * secs / (ms * 1e-3):10.0f} audio-s/s  RTF {ms * 1e-3 / (B * secs):.2e}", flush=True)


if __name__ == "__main__":
    main()
