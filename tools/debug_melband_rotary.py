#!/usr/bin/env python3
"""Debug helper: the bf16 path on the fold fixture, two identical clips in one batch: where do the rows differ, run to run and from a saved reference output?  argv[1]: npy to save / compare."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
from audio_denoiser_onnx_amd import melband
from audio_denoiser_onnx_amd.session import InferenceSession
from audio_denoiser_onnx_amd.weights import pack_blob
import test_melband as tm
from ade_testlib import melband_fixture_weights
w = melband_fixture_weights()[2]
zf = np.load(tm.GOLD_FOLD)
blob = pack_blob(melband.model_tensors(w))
x = np.stack((zf["pcm_in"], zf["pcm_in"])).reshape(2, -1)
meta = melband.metadata(int(zf["input_audio_length"]), use_batch_fold=True, batch_window_seconds=float(zf["batch_window_seconds"]), gemm_dtype="bf16")
with InferenceSession(weights=blob, metadata=meta) as sess:
    if os.environ.get("ADE_ROT_DEBUG"):
        sess.set_option("graph", "0")
    outs = [sess.process(x, want_f32=True)[1].copy() for _ in range(3)]
    tok = sess.tap("tokens", 10 ** 9) if False else None
for k, o in enumerate(outs):
    d = o[0] != o[1]
    print(f"run {k}: rows differ at {int(d.sum())} of {d.size} samples; first {np.flatnonzero(d)[:5]}, max |d| {np.abs(o[0] - o[1]).max():.3e}; vs run 0: {int((o != outs[0]).sum())}")
if len(sys.argv) > 1:
    p = sys.argv[1]
    if os.path.exists(p):
        ref = np.load(p)
        print("vs saved:", int((outs[0] != ref).sum()), "of", ref.size, "max |d|", float(np.abs(outs[0] - ref).max()))
    else:
        np.save(p, outs[0])
