#!/usr/bin/env python3
"""the REFERENCE's own forward, timed on this container's CPU (the "reference-semantics" number).

Runs only in the build container (needs /root/reference; it is read, never copied).  Imports the reference's classes the way the golden-vector generators do
(tools/ref_import.py), builds the seeded GTCRN_CUSTOM module of tools/make_golden_gtcrn.py, and times `GTCRN_CUSTOM.forward`
(GTCRN/Export_GTCRN.py:636-693) in PyTorch eager mode, one 1 s chunk per call exactly as the reference's slice loop calls its session
(GTCRN/Inference_GTCRN_ONNX.py:323-343), on the bench's synthetic chunks (audio_denoiser_onnx_amd.synth) with torch.set_num_threads(8) and (1).
ONNX Runtime -- what the reference actually runs -- is not installable here; eager PyTorch executes the same graph the export traces.

    python tools/ref_eager_cpu_timing.py [--chunks 64] [--out profiles/r06_ref_eager_cpu.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)

from make_golden_gtcrn import build_reference  # noqa: E402
from audio_denoiser_onnx_amd.synth import synth_batch  # noqa: E402


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunks", type=int, default=64)
    ap.add_argument("--out", default=os.path.join(REPO, "profiles", "r06_ref_eager_cpu.json"))
    a = ap.parse_args()
    _, custom, _ = build_reference(0)
    x = torch.from_numpy(synth_batch(a.chunks, 16000)).reshape(a.chunks, 1, 1, 16000)
    res = {"what": "the reference's GTCRN_CUSTOM.forward (GTCRN/Export_GTCRN.py:636-693) in PyTorch eager mode on the BUILD CONTAINER's CPU, one 1 s chunk per call as in "
                   "GTCRN/Inference_GTCRN_ONNX.py:323-343; seeded weights (tools/make_golden_gtcrn.py seed 0); the bench's synthetic chunks; BASELINE.md section 3 item 1",
           "torch": torch.__version__, "cpu": cpu_model(), "hardware_threads": os.cpu_count(), "chunks": a.chunks, "runs": {}}
    with torch.no_grad():
        for threads in (8, 1):
            torch.set_num_threads(threads)
            for i in range(min(10, a.chunks)):
                custom(x[i])
            t0 = time.perf_counter()
            for i in range(a.chunks):
                y = custom(x[i])
            wall = time.perf_counter() - t0
            secs = a.chunks * 15872 / 16000.0
            res["runs"][f"threads_{threads}"] = {"wall_s": round(wall, 3), "ms_per_chunk": round(1e3 * wall / a.chunks, 3), "audio_s_per_s": round(secs / wall, 2),
                                                 "rtf": float(f"{wall / secs:.3e}")}
            print(threads, res["runs"][f"threads_{threads}"], tuple(y.shape), flush=True)
    with open(a.out, "w") as f:
        json.dump(res, f, indent=1)
    print("wrote", a.out)


if __name__ == "__main__":
    main()
