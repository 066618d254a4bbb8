#!/usr/bin/env python3
"""Stage-by-stage comparison of the HIP Mel-Band-Roformer engine with the numpy oracle on the golden fixture (GPU box)."""
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))
from audio_denoiser_onnx_amd import melband, weightgen  # noqa: E402
from audio_denoiser_onnx_amd.session import InferenceSession  # noqa: E402
from audio_denoiser_onnx_amd.weights import pack_blob  # noqa: E402
from melband_oracle import MelBandOracle  # noqa: E402

z = np.load(os.path.join(REPO, "tests", "golden", "melband_seed0_io.npz"))
spec = [(n, s, sc) for n, s, sc in json.loads(str(z["spec"]))]
w = weightgen.materialise(spec)
T, L = int(z["frames"]), z["pcm_in"].shape[1]
o = MelBandOracle(w, z["freq_indices"], z["dim_inputs"], T, int(z["depth"]), exact_dft=True)
want = o.process(z["pcm_in"])
sess = InferenceSession(weights=pack_blob(melband.model_tensors(w)), metadata=melband.metadata(L))
got = sess.run(None, {"noisy_audio": z["pcm_in"][None]})[0][0]
sp = sess.tap("spec", 2050 * 2 * T).reshape(2050, 2, T).transpose(0, 2, 1)
print("spec   max err", np.abs(sp - o.taps["spec"]).max(), "ref max", np.abs(o.taps["spec"]).max())
tok = sess.tap("tokens", 60 * T * 384).reshape(60, T, 384)
e = np.abs(tok - o.taps["tf_out"])
print("tokens max err", e.max(), "median", np.median(e), "per band max", e.reshape(60, -1).max(1).round(4)[:12])
mk = sess.tap("mask", 2050 * 2 * T).reshape(2050, 2, T).transpose(0, 2, 1)
e = np.abs(mk - o.taps["mask_avg"])
print("mask   max err", e.max(), "median", np.median(e), "argmax fc", np.unravel_index(e.argmax(), e.shape))
d = got.astype(int) - want.astype(int)
print("pcm max", np.abs(d).max(), "nonzero frac", (d != 0).mean())
