#!/usr/bin/env python3
"""Streaming GTCRN throughput / latency on one MI355X: S concurrent streams advanced N frames (N * 16 ms) per push.

    python tools/bench_streaming.py [--streams 256,1024] [--frames 2,4,16,62] [--pushes 50]

Reports the device time of one push (all S streams), the real-time margin (audio advanced per push / push time) and the number of
real-time streams one GPU sustains at that push size.  PCM is resident in HBM (ade_stream_push_device).
"""
import argparse
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
os.chdir(REPO)

import torch  # noqa: E402

from ade_testlib import make_session  # noqa: E402
from audio_denoiser_onnx_amd.session import StreamingSession  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", default="256,1024")
    ap.add_argument("--frames", default="2,4,16,62")
    ap.add_argument("--pushes", type=int, default=50)
    ap.add_argument("--fused", default="1", help="1: a push is ONE launch of the chunk kernel with a carried state (default); 0: the 48-launch multi-kernel push")
    a = ap.parse_args()
    sess = make_session(None, seed=0)
    sess.set_option("fused", a.fused)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    for S in [int(x) for x in a.streams.split(",")]:
        for N in [int(x) for x in a.frames.split(",")]:
            P = N * 256
            pcm = torch.from_numpy((np.random.default_rng(S + N).standard_normal((S, P)) * 3000).astype(np.int16)).to(dev)
            out = torch.empty_like(pcm)
            with StreamingSession(sess, S, N) as st, torch.cuda.stream(stream):
                for _ in range(3):
                    st.push_device(pcm, out, stream=stream.cuda_stream)
                stream.synchronize()
                t = time.perf_counter()
                for _ in range(a.pushes):
                    st.push_device(pcm, out, stream=stream.cuda_stream)
                stream.synchronize()
                ms = (time.perf_counter() - t) / a.pushes * 1e3
            audio_ms = N * 16.0
            print(f"streams={S:5d} frames/push={N:3d} ({audio_ms:6.0f} ms of audio): {ms:8.3f} ms/push  real-time margin {audio_ms / ms:8.1f}x  "
                  f"-> {int(S * audio_ms / ms):8d} real-time streams, {S * audio_ms / ms:10.0f} audio-s/s", flush=True)


if __name__ == "__main__":
    main()
