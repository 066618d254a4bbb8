"""Mel-Band-Roformer's DYNAMIC_AXES export: any input length, other input / output sample rates, the dynamic ISTFT trim
(Mel_Band_Roformer/Stereo/Export_MelBandRoformer.py:33, :50-53, :630-644, :660-680; Stereo/STFT_Process.py:296-306).

tests/golden/melband_dynamic_seed0.npz holds the reference's own forward on four such exports (tools/make_golden_melband.py --dynamic); the weights are the
counter-based ones of melband_seed0_io.npz.  The last half window of a dynamic output is the last frame alone divided by its squared Hann window, which falls to
5.5e-12 at the end: fp32 round-off of the inverse DFT is amplified there (the reference's own output saturates), so that stretch is compared relative to the signal."""
import json
import math
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))

from audio_denoiser_onnx_amd import weightgen  # noqa: E402

GOLD = os.path.join(HERE, "golden", "melband_seed0_io.npz")
GOLD_DYN = os.path.join(HERE, "golden", "melband_dynamic_seed0.npz")
SR = 44100


@pytest.fixture(scope="module")
def fixture():
    from ade_testlib import melband_fixture_weights
    z, _, w = melband_fixture_weights()
    d = np.load(GOLD_DYN)
    cases = [(tag, d[tag + "_in"], d[tag + "_out"], int(d[tag + "_rates"][0]), int(d[tag + "_rates"][1])) for tag in json.loads(str(d["cases"]))]
    return z, w, cases


def model_length(n: int, in_rate: int) -> int:
    return n if in_rate == SR else int(math.floor(n * float(SR / in_rate)))


def compare(got: np.ndarray, ref: np.ndarray, out_rate: int, body_lsb: int, tag: str):
    assert got.shape == ref.shape and got.dtype == np.int16, (tag, got.shape, ref.shape)
    d = np.abs(got.astype(np.int32) - ref.astype(np.int32))
    n = ref.shape[1]
    tail = int(math.ceil(1024 * out_rate / SR)) + 2
    assert d[:, :n - tail].max() <= body_lsb and (d[:, :n - tail] != 0).mean() < 0.10, (tag, d[:, :n - tail].max(), (d[:, :n - tail] != 0).mean())
    lim = body_lsb + 5e-3 * np.abs(ref[:, n - tail:].astype(np.float64))
    assert (d[:, n - tail:] <= lim).all(), (tag, d[:, n - tail:].max())


def test_fixture_shapes_follow_the_dynamic_trim(fixture):
    for tag, pcm, ref, sri, sro in fixture[2]:
        L = model_length(pcm.shape[1], sri)
        Lo = 441 * (L // 441) + 1024
        assert ref.shape == (2, Lo if sro == SR else int(math.floor(Lo * float(sro / SR)))), tag


def test_oracle_matches_reference_dynamic_exports(fixture):
    from melband_oracle import MelBandOracle
    z, w, cases = fixture
    for tag, pcm, ref, sri, sro in cases:
        L = model_length(pcm.shape[1], sri)
        o = MelBandOracle(w, z["freq_indices"], z["dim_inputs"], L // 441 + 1, int(z["depth"]), dynamic=True, length=L)
        compare(o.process_rates(pcm, sri, sro), ref, sro, 1, tag)


def test_metadata_refuses_inconsistent_exports():
    from audio_denoiser_onnx_amd import melband
    with pytest.raises(ValueError):
        melband.metadata(13230, in_sample_rate=48000)                       # static + other rates
    with pytest.raises(ValueError):
        melband.metadata(30000, use_batch_fold=True, dynamic_axes=True)     # fold needs a static shape (:46)
    with pytest.raises(ValueError):
        melband.metadata(13000)                                             # static: whole hops
    m = melband.metadata(13000, dynamic_axes=True, in_sample_rate=48000, out_sample_rate=22050)
    assert m["dynamic_axes"] == "1" and m["in_sample_rate"] == "48000" and m["out_sample_rate"] == "22050" and m["model_sample_rate"] == "44100"


# ---- GPU: the HIP engine through the C ABI ------------------------------------------------------------------------------
@pytest.mark.gpu
def test_gpu_matches_reference_dynamic_exports(fixture):
    from audio_denoiser_onnx_amd import melband
    from audio_denoiser_onnx_amd.session import InferenceSession
    from audio_denoiser_onnx_amd.weights import pack_blob
    z, w, cases = fixture
    blob = pack_blob(melband.model_tensors(w))
    for tag, pcm, ref, sri, sro in cases:
        with InferenceSession(weights=blob, metadata=melband.metadata(pcm.shape[1], dynamic_axes=True, in_sample_rate=sri, out_sample_rate=sro)) as sess:
            assert sess.frames == model_length(pcm.shape[1], sri) // 441 + 1
            out = sess.run(None, {"noisy_audio": pcm[None]})[0]
            assert out.shape == (1,) + ref.shape
            compare(out[0], ref, sro, 2, tag)
            # batch rows are independent clips, bit for bit
            other = np.ascontiguousarray(pcm[::-1, ::-1] // 2)
            both = sess.run(None, {"noisy_audio": np.stack((other, pcm))})[0]
            assert np.array_equal(both[1], out[0]), tag


@pytest.mark.gpu
def test_gpu_static_export_still_refuses_other_rates(fixture):
    from audio_denoiser_onnx_amd import melband
    from audio_denoiser_onnx_amd._lib import AdeUnsupportedError
    from audio_denoiser_onnx_amd.session import InferenceSession
    from audio_denoiser_onnx_amd.weights import pack_blob
    z, w, _ = fixture
    meta = melband.metadata(13230)
    meta["in_sample_rate"] = "48000"
    with pytest.raises(AdeUnsupportedError, match="dynamic_axes=1"):
        InferenceSession(weights=pack_blob(melband.model_tensors(w)), metadata=meta)
