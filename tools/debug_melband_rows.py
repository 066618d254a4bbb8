#!/usr/bin/env python3
"""Debug helper: batch-position independence and run-to-run determinism of the Mel-Band engine at depth D."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
from audio_denoiser_onnx_amd import melband, weightgen
from audio_denoiser_onnx_amd.session import InferenceSession
from audio_denoiser_onnx_amd.weights import pack_blob
D = int(sys.argv[1]) if len(sys.argv) > 1 else 6
L = int(sys.argv[2]) if len(sys.argv) > 2 else 66150
blob = pack_blob(melband.model_tensors(weightgen.materialise(melband.synthetic_spec(D))))
rng = np.random.default_rng(1)
sig = lambda a: np.clip(rng.standard_normal(L) * a, -32768, 32767).astype(np.int16)
rows = np.stack([np.stack((sig(6000.0), sig(4000.0))), np.stack((sig(500.0), np.zeros(L, np.int16)))])
with InferenceSession(weights=blob, metadata=melband.metadata(L)) as sess:
    a = sess.run(None, {"noisy_audio": rows})[0]
    a2 = sess.run(None, {"noisy_audio": rows})[0]
    b = sess.run(None, {"noisy_audio": rows[::-1].copy()})[0]
    one = sess.run(None, {"noisy_audio": rows[1:2]})[0]
d = lambda x, y: (int(np.abs(x.astype(np.int32) - y.astype(np.int32)).max()), float((x != y).mean()))
print("depth", D, "L", L)
print("same input twice:", d(a, a2))
print("reversed batch  :", d(a[::-1], b))
print("solo row 1      :", d(one[0], a[1]))
