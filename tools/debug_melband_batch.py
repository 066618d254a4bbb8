#!/usr/bin/env python3
"""Which stage of the Mel-Band-Roformer engine depends on the batch size?  Runs the fixture clip alone and as row 0 of a batch of 3 and compares the taps."""
import json, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from audio_denoiser_onnx_amd import melband, weightgen
from audio_denoiser_onnx_amd.session import InferenceSession
from audio_denoiser_onnx_amd.weights import pack_blob
z = np.load(os.path.join(REPO, "tests", "golden", "melband_seed0_io.npz"))
w = weightgen.materialise([(n, s, sc) for n, s, sc in json.loads(str(z["spec"]))])
a = z["pcm_in"]; L = a.shape[1]; T = int(z["frames"])
b = np.ascontiguousarray(a[::-1, ::-1] // 2)
sess = InferenceSession(weights=pack_blob(melband.model_tensors(w)), metadata=melband.metadata(L))
def run(x):
    B = x.shape[0]
    out, f32 = sess.process(x.reshape(B, -1), want_f32=True)
    sp = sess.tap("spec", 2050 * 2 * B * T).reshape(2050, 2, B, T)
    tk = sess.tap("tokens", 60 * B * T * 384).reshape(60, B, T, 384)
    mk = sess.tap("mask", 2050 * 2 * B * T).reshape(2050, 2, B, T)
    return out, f32, sp, tk, mk
o1 = run(a[None]); o3 = run(np.stack((a, b, a)))
for name, i in (("spec", 2), ("tokens", 3), ("mask", 4)):
    x1 = o1[i][:, :, 0] if name != "tokens" else o1[i][:, 0]
    x3 = o3[i][:, :, 0] if name != "tokens" else o3[i][:, 0]
    print(name, "max |d| alone vs batch row 0:", float(np.abs(x1 - x3).max()), "bit-equal:", np.array_equal(x1, x3))
print("f32 wave", float(np.abs(o1[1][0] - o3[1][0]).max()), "pcm", int(np.abs(o1[0][0].astype(int) - o3[0][0].astype(int)).max()))
print("row 0 vs row 2 inside the batch: pcm equal", np.array_equal(o3[0][0], o3[0][2]))
