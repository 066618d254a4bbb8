#!/usr/bin/env python3
"""UL-UNAS throughput on one MI355X (seeded reference-architecture weights from the golden fixture, synthetic PCM resident in HBM).

    python tools/bench_ulunas.py [--batches 64,256,1024] [--steps 10]
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

from audio_denoiser_onnx_amd import ulunas  # noqa: E402
from audio_denoiser_onnx_amd.session import InferenceSession  # noqa: E402
from audio_denoiser_onnx_amd.weights import pack_blob  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="64,256,1024")
    ap.add_argument("--steps", type=int, default=10)
    a = ap.parse_args()
    z = np.load(os.path.join(REPO, "tests", "golden", "ulunas_seed0.npz"))
    fused = ulunas.fold_state_dict({str(k): z["w:" + str(k)] for k in z["keys"]})
    sess = InferenceSession(weights=pack_blob(fused), metadata=ulunas.metadata(16000))
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    for B in [int(x) for x in a.batches.split(",")]:
        pcm = torch.from_numpy((np.random.default_rng(B).standard_normal((B, sess.row_in)) * 3000).astype(np.int16)).to(dev)
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
        print(f"B={B:5d} x 1 s: {ms:9.3f} ms/step  {B / (ms * 1e-3):10.0f} audio-s/s  RTF {ms * 1e-3 / B:.2e}", flush=True)


if __name__ == "__main__":
    main()
