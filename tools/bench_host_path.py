#!/usr/bin/env python3
"""GTCRN through the HOST-buffer entry point (ade_process: H2D + kernels + D2H, synchronous) beside the device-resident one, for
BASELINE.json's two GTCRN configurations (1 and 256 chunks of 1 s).  The bench.py `value` is the device-resident rate; this tool
measures the PCIe-inclusive figure DESIGN.md quotes.

    python tools/bench_host_path.py [--steps 200]
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

from audio_denoiser_onnx_amd import synth  # noqa: E402
from audio_denoiser_onnx_amd.session import InferenceSession  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    a = ap.parse_args()
    from audio_denoiser_onnx_amd.metadata import build_audio_metadata
    with open(os.path.join(REPO, "tests", "golden", "gtcrn_seed0.adew"), "rb") as f:
        blob = f.read()
    meta = build_audio_metadata(producer="bench_host_path.py", model_name="GTCRN", task="denoise", model_family="gtcrn", input_audio_length=16000)
    sess = InferenceSession(weights=blob, metadata=meta)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    for B in (1, 256):
        pcm = np.ascontiguousarray(synth.synth_batch(B, sess.in_len)).reshape(B, 1, sess.in_len)
        sess.reserve(B)
        name = sess.get_inputs()[0].name
        sess.run(None, {name: pcm})
        t = time.perf_counter()
        for _ in range(a.steps):
            sess.run(None, {name: pcm})
        host_ms = (time.perf_counter() - t) / a.steps * 1e3
        # the same call on page-locked caller buffers (torch's pinned allocator here; hipHostMalloc in a C caller): hipMemcpy then DMAs directly
        pin_in = torch.empty(pcm.shape, dtype=torch.int16).pin_memory()
        pin_in.numpy()[...] = pcm
        pin_out = torch.empty((B, sess.row_out), dtype=torch.int16).pin_memory()
        p_in, p_out = pin_in.numpy().reshape(B, -1), pin_out.numpy()
        sess.process_into(p_in, p_out)
        t = time.perf_counter()
        for _ in range(a.steps):
            sess.process_into(p_in, p_out)
        pin_ms = (time.perf_counter() - t) / a.steps * 1e3
        reuse_in, reuse_out = pcm.reshape(B, -1), np.zeros((B, sess.row_out), np.int16)
        sess.process_into(reuse_in, reuse_out)
        t = time.perf_counter()
        for _ in range(a.steps):
            sess.process_into(reuse_in, reuse_out)
        reuse_ms = (time.perf_counter() - t) / a.steps * 1e3
        assert np.array_equal(reuse_out, p_out)
        d_in = torch.from_numpy(pcm.reshape(B, -1)).to(dev)
        d_out = torch.empty((B, sess.row_out), dtype=torch.int16, device=dev)
        with torch.cuda.stream(stream):
            sess.run_device(d_in, d_out, stream=stream.cuda_stream)
            stream.synchronize()
            t = time.perf_counter()
            for _ in range(a.steps):
                sess.run_device(d_in, d_out, stream=stream.cuda_stream)
            stream.synchronize()
            dev_ms = (time.perf_counter() - t) / a.steps * 1e3
            t = time.perf_counter()
            for _ in range(a.steps):
                sess.run_device(d_in, d_out, stream=stream.cuda_stream)
                stream.synchronize()
            sync_ms = (time.perf_counter() - t) / a.steps * 1e3
        print(f"B={B:4d} x 1 s: session.run (fresh pageable output every call) {host_ms:7.3f} ms = {B / (host_ms * 1e-3):9.0f} audio-s/s | "
              f"re-used pageable buffers {reuse_ms:7.3f} ms | re-used page-locked buffers {pin_ms:7.3f} ms = {B / (pin_ms * 1e-3):9.0f} audio-s/s | device-resident back-to-back {dev_ms:7.3f} ms | device-resident, synchronised every call {sync_ms:7.3f} ms", flush=True)


if __name__ == "__main__":
    main()
