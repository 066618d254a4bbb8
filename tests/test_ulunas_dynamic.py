"""UL-UNAS's DYNAMIC_AXES export: any input length, other input / output sample rates, the ISTFT's kept tail sliced to the caller-rate input length
(UL-UNAS/Export_UL_UNAS.py:26, :41-43, :835-845, :851-868, :888-905; UL-UNAS/STFT_Process.py:170-177, 317-326).

tests/golden/ulunas_dynamic_seed0.npz holds the reference's own forward on four such exports (tools/make_golden_ulunas.py --dynamic), over the seeded network of
ulunas_seed0.npz.  Where the kept tail reaches the end of the last frame (the down-sampling case) the samples are that frame alone divided by its squared Hann window,
which falls to 1.4e-9: fp32 round-off of the inverse DFT is amplified there (the reference's own output saturates), so that stretch is compared relative to the signal."""
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

from audio_denoiser_onnx_amd import ulunas  # noqa: E402

GOLD = os.path.join(HERE, "golden", "ulunas_seed0.npz")
GOLD_DYN = os.path.join(HERE, "golden", "ulunas_dynamic_seed0.npz")


@pytest.fixture(scope="module")
def fixture():
    z = np.load(GOLD)
    state = {str(k): z["w:" + str(k)] for k in z["keys"]}
    d = np.load(GOLD_DYN)
    cases = [(tag, d[tag + "_in"], d[tag + "_out"], int(d[tag + "_rates"][0]), int(d[tag + "_rates"][1])) for tag in json.loads(str(d["cases"]))]
    return ulunas.fold_state_dict(state), cases


def model_length(n: int, in_rate: int) -> int:
    return n if in_rate == 16000 else int(math.floor(n * (1.0 / (in_rate / 16000.0))))


def compare(got: np.ndarray, ref: np.ndarray, out_rate: int, body_lsb: int, tag: str):
    assert got.shape == ref.shape and got.dtype == np.int16, (tag, got.shape, ref.shape)
    d = np.abs(got.astype(np.int32) - ref.astype(np.int32))
    n = ref.shape[0]
    tail = int(math.ceil(256 * out_rate / 16000)) + 2
    assert d[:n - tail].max() <= body_lsb and (d[:n - tail] != 0).mean() < 0.05, (tag, d[:n - tail].max(), (d[:n - tail] != 0).mean())
    lim = body_lsb + 5e-3 * np.abs(ref[n - tail:].astype(np.float64))
    assert (d[n - tail:] <= lim).all(), (tag, d[n - tail:].max())


def test_fixture_lengths_follow_the_dynamic_slice(fixture):
    for tag, pcm, ref, sri, sro in fixture[1]:
        Lm = model_length(len(pcm), sri)
        keep = min(256 * (Lm // 256 + 1), len(pcm))                     # audio[..., :audio_len]: the CALLER-rate length (:851, :888-889)
        assert len(ref) == (keep if sro == 16000 else int(math.floor(keep * (sro / 16000.0)))), tag


def test_oracle_matches_reference_dynamic_exports(fixture):
    from ulunas_oracle import UlunasOracle
    fused, cases = fixture
    for tag, pcm, ref, sri, sro in cases:
        o = UlunasOracle(fused, ulunas.block_plan(), model_length(len(pcm), sri), dynamic_keep=len(pcm))
        compare(o.process_rates(pcm[None], sri, sro)[0], ref, sro, 1, tag)


def test_metadata_refuses_inconsistent_exports():
    with pytest.raises(ValueError):
        ulunas.metadata(16000, in_sample_rate=48000)
    with pytest.raises(ValueError):
        ulunas.metadata(16000, use_batch_fold=True, dynamic_axes=True)
    m = ulunas.metadata(7000, dynamic_axes=True, in_sample_rate=48000, out_sample_rate=8000)
    assert m["dynamic_axes"] == "1" and m["in_sample_rate"] == "48000" and m["out_sample_rate"] == "8000" and m["model_sample_rate"] == "16000"


@pytest.mark.hipsim
def test_hipsim_dynamic_export_matches_oracle(fixture):
    """The same csrc/ compiled for the host simulator: a short dynamic clip at 24 kHz in, 8 kHz out against the oracle (exactly reduced DFT angles)."""
    from ade_testlib import hipsim_library
    from ulunas_oracle import UlunasOracle
    from audio_denoiser_onnx_amd.session import InferenceSession
    from audio_denoiser_onnx_amd.weights import pack_blob
    fused, cases = fixture
    pcm = np.ascontiguousarray(cases[0][1][:3000])
    o = UlunasOracle(fused, ulunas.block_plan(), model_length(3000, 24000), exact_dft=True, dynamic_keep=3000)
    want = o.process_rates(pcm[None], 24000, 8000)[0]
    with InferenceSession(weights=pack_blob(fused), metadata=ulunas.metadata(3000, dynamic_axes=True, in_sample_rate=24000, out_sample_rate=8000),
                          library=hipsim_library()) as sess:
        got = sess.run(None, {"noisy_audio": pcm[None, None]})[0][0, 0]
    compare(got, want, 8000, 1, "hipsim")


# ---- GPU: the HIP engine through the C ABI ------------------------------------------------------------------------------
@pytest.mark.gpu
def test_gpu_matches_reference_dynamic_exports(fixture):
    from audio_denoiser_onnx_amd.session import InferenceSession
    from audio_denoiser_onnx_amd.weights import pack_blob
    fused, cases = fixture
    blob = pack_blob(fused)
    for tag, pcm, ref, sri, sro in cases:
        with InferenceSession(weights=blob, metadata=ulunas.metadata(len(pcm), dynamic_axes=True, in_sample_rate=sri, out_sample_rate=sro)) as sess:
            assert sess.frames == model_length(len(pcm), sri) // 256 + 1
            out = sess.run(None, {"noisy_audio": pcm[None, None]})[0]
            assert out.shape == (1, 1, len(ref))
            compare(out[0, 0], ref, sro, 1, tag)
            other = np.ascontiguousarray(pcm[::-1] // 2)
            both = sess.run(None, {"noisy_audio": np.stack((other, pcm))[:, None]})[0]
            assert np.array_equal(both[1, 0], out[0, 0]), tag             # batch rows are independent clips, bit for bit


@pytest.mark.gpu
def test_gpu_static_export_still_refuses_other_rates(fixture):
    from audio_denoiser_onnx_amd._lib import AdeUnsupportedError
    from audio_denoiser_onnx_amd.session import InferenceSession
    from audio_denoiser_onnx_amd.weights import pack_blob
    meta = ulunas.metadata(16000)
    meta["in_sample_rate"] = "48000"
    with pytest.raises(AdeUnsupportedError, match="dynamic_axes=1"):
        InferenceSession(weights=pack_blob(fixture[0]), metadata=meta)
