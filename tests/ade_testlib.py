"""Shared helpers for the parity tests (CPU hipsim + real GPU)."""
from __future__ import annotations

import os
import subprocess

import numpy as np

from audio_denoiser_onnx_amd import _lib
from audio_denoiser_onnx_amd.metadata import build_audio_metadata
from audio_denoiser_onnx_amd.session import InferenceSession

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
GOLD = os.path.join(HERE, "golden")
HIPSIM_LIB = os.path.join(HERE, "hipsim", "_build", "libade_hipsim.so")


_MELBAND_CACHE = {}


def melband_fixture_weights():
    """(npz, spec, fused buffers) of tests/golden/melband_seed0_io.npz: 208 M generator-made floats, materialised once per test process (three modules use them)."""
    import json
    from audio_denoiser_onnx_amd import weightgen
    if "w" not in _MELBAND_CACHE:
        z = np.load(os.path.join(GOLD, "melband_seed0_io.npz"))
        spec = [(n, s, sc) for n, s, sc in json.loads(str(z["spec"]))]
        _MELBAND_CACHE["w"] = (z, spec, weightgen.materialise(spec))
    return _MELBAND_CACHE["w"]


def golden_blob(seed: int = 0) -> bytes:
    with open(os.path.join(GOLD, f"gtcrn_seed{seed}.adew"), "rb") as f:
        return f.read()


def golden_inputs():
    return dict(np.load(os.path.join(GOLD, "gtcrn_inputs.npz")))


def default_meta(length: int = 16000, **over):
    meta = build_audio_metadata(producer="tests", model_name="GTCRN", task="denoise", model_family="gtcrn",
                                input_audio_length=length)
    meta.update({k: str(v) for k, v in over.items()})
    return meta


def hipsim_library() -> _lib.AdeLibrary:
    """Build (if stale) and load the host-simulated engine.  TEST-ONLY: same csrc/*.hip, g++ + tests/hipsim shim."""
    import glob
    srcs = sorted(glob.glob(os.path.join(REPO, "audio_denoiser_onnx_amd", "csrc", "*.hip")) + glob.glob(os.path.join(REPO, "audio_denoiser_onnx_amd", "csrc", "*.h")))
    srcs += [os.path.join(HERE, "hipsim", "hipsim.cpp"), os.path.join(HERE, "hipsim", "hip", "hip_runtime.h")]
    if not os.path.exists(HIPSIM_LIB) or any(os.path.getmtime(s) > os.path.getmtime(HIPSIM_LIB) for s in srcs):
        subprocess.run([os.path.join(HERE, "hipsim", "build.sh")], check=True)
    return _lib.AdeLibrary(HIPSIM_LIB)


def make_session(library=None, seed: int = 0, length: int = 16000, **kw) -> InferenceSession:
    return InferenceSession(weights=golden_blob(seed), metadata=default_meta(length), library=library, **kw)


def compare_taps(sess: InferenceSession, oracle, batch: int, row: int = 0):
    """Engine-layout taps of batch row `row` vs the oracle's reference-layout taps of ITS row 0.
    Returns {tap: (max_abs_err, ref_abs_max)}."""
    T = sess.frames

    def tap(name, per_frame):
        return sess.tap(name, batch * T * per_frame).reshape(batch, T, -1)[row]

    def nchw(name, C, F):   # oracle (C,T,F) -> (T,F,C)
        return oracle.tap(name).reshape(C, T, F).transpose(1, 2, 0)

    def gated(name):
        x = tap("x_" + name, 33 * 16).reshape(T, 33, 16).copy()
        at = tap("at_" + name, 8).reshape(T, 8)
        x[:, :, 0::2] *= at[:, None, :]
        return x

    res = {}

    def put(name, got, want):
        res[name] = (float(np.abs(got - want).max()), float(np.abs(want).max()))

    put("spec", tap("spec", 2 * 260).reshape(T, 2, 260)[:, :, :257], oracle.tap("spec").reshape(2, 257, T).transpose(2, 0, 1))
    try:   # on the fused path the ERB features never leave LDS (tap raises FileNotFoundError)
        put("feat_erb", tap("feat", 3 * 132).reshape(T, 3, 132)[:, :, :129], oracle.tap("feat_erb").reshape(3, T, 129).transpose(1, 0, 2))
    except FileNotFoundError:
        pass
    put("e0", tap("e0", 65 * 16).reshape(T, 65, 16), nchw("e0", 16, 65))
    put("e1", tap("e1", 33 * 16).reshape(T, 33, 16), nchw("e1", 16, 33))
    for n in ("e2", "e3", "e4", "d0", "d1", "d2"):
        put(n, gated(n), nchw(n, 16, 33))
    put("dp1", tap("dp1", 33 * 16).reshape(T, 33, 16), oracle.tap("dp1").reshape(T, 33, 16))
    put("dp2", tap("dp2", 33 * 16).reshape(T, 33, 16), oracle.tap("dp2").reshape(T, 33, 16))
    try:   # d3 and the mask stay in LDS on the fused path
        put("d3", tap("d3", 65 * 16).reshape(T, 65, 16), nchw("d3", 16, 65))
        put("d4", tap("mask", 2 * 132).reshape(T, 2, 132)[:, :, :129], oracle.tap("d4").reshape(2, T, 129).transpose(1, 0, 2))
    except FileNotFoundError:
        pass
    return res
