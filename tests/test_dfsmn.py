"""DFSMN: oracle pinned to the reference-generated fixture; engine vs oracle / fixture.

Fixture: tools/make_golden_dfsmn.py runs the reference's own ``DFSMN.forward`` on a seeded parameter tree (the reference
gets its parameters from modelscope, absent) with this package's restatement of the Kaldi mel bank standing in for
torchaudio's (absent): both are INPUTS of the forward being pinned, not part of it.
"""
import os
import sys

import numpy as np
import pytest

from ade_testlib import GOLD
from audio_denoiser_onnx_amd.weights import load_blob

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
from dfsmn_oracle import DfsmnOracle  # noqa: E402


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "dfsmn_seed0_io.npz"))


@pytest.fixture(scope="module")
def tensors():
    return load_blob(os.path.join(GOLD, "dfsmn_seed0.adew"))


def test_dfsmn_oracle_matches_reference(gold, tensors):
    L = int(gold["input_audio_length"])
    o = DfsmnOracle(tensors, L)
    assert o.frames == int(gold["frames"]) == 24 and o.out_len == L
    names = ["speech0", "speech1", "randn", "zeros"]
    pcm, _ = o.process(np.stack([gold[f"{n}.pcm_in"] for n in names]))
    assert np.abs(o.taps["logmel"] - gold["speech0.logmel"]).max() <= 2e-4            # log of a 1025-term fp32 sum
    assert np.abs(o.taps["mask"] - gold["speech0.mask"]).max() <= 1e-4
    for i, n in enumerate(names):
        d = np.abs(pcm[i].astype(np.int32) - gold[f"{n}.pcm_out"].astype(np.int32)).max()
        assert d <= 1, (n, d)
    assert not pcm[3].any()
    m = gold["speech0.mask"]
    assert float(m.std()) > 0.05 and float(m.min()) < 0.2 and float(m.max()) > 0.8      # a non-degenerate mask


def _dfsmn_meta(length, in_rate=48000, out_rate=48000, **kw):
    from audio_denoiser_onnx_amd.metadata import build_audio_metadata
    return build_audio_metadata(producer="tests", model_name="DFSMN", task="denoise", model_family="dfsmn", input_audio_length=length,
                                in_sample_rate=in_rate, out_sample_rate=out_rate, model_sample_rate=48000, nfft=1920, window_length=1920,
                                hop_length=960, window_type="hamming", center_pad=False, pad_mode="constant", feature_kind="kaldi_fbank_stft", **kw)


def _blob_bytes():
    with open(os.path.join(GOLD, "dfsmn_seed0.adew"), "rb") as f:
        return f.read()


@pytest.mark.hipsim
def test_hipsim_dfsmn_tiny(tensors):
    """The whole DFSMN launch sequence under the host simulator on a 3-frame input (FFT analysis / synthesis, MFMA layers emulated)."""
    import time
    from ade_testlib import hipsim_library
    from audio_denoiser_onnx_amd.session import InferenceSession
    L = 1920 + 2 * 960
    sess = InferenceSession(weights=_blob_bytes(), metadata=_dfsmn_meta(L), library=hipsim_library())
    assert (sess.in_len, sess.out_len, sess.frames, sess.sample_rate) == (L, L, 3, 48000)
    g = np.load(os.path.join(GOLD, "dfsmn_seed0_io.npz"))
    x = g["speech0.pcm_in"][5000:5000 + L][None]
    t0 = time.time()
    pcm, f32 = sess.process(x, want_f32=True)
    o = DfsmnOracle(tensors, L, exact_dft=True)
    opcm, of32 = o.process(x)
    assert np.abs(sess.tap("logmel", 120 * 3).reshape(120, 3) - o.taps["logmel"]).max() <= 2e-4
    assert np.abs(sess.tap("mask", 961 * 3).reshape(3, 961).T - o.taps["mask"]).max() <= 1e-4
    assert np.abs(f32 - of32).max() <= 2e-5
    assert np.abs(pcm.astype(np.int32) - opcm.astype(np.int32)).max() <= 1
    print("sim seconds", time.time() - t0)


@pytest.mark.gpu
def test_gpu_dfsmn_reference_golden(gold, tensors):
    """HIP path through the C ABI vs the reference-generated fixture and vs the oracle with exact DFT tables."""
    from audio_denoiser_onnx_amd.session import InferenceSession
    L = int(gold["input_audio_length"])
    sess = InferenceSession(weights=_blob_bytes(), metadata=_dfsmn_meta(L))
    assert (sess.in_len, sess.out_len, sess.frames, sess.sample_rate) == (L, L, 24, 48000)
    names = ["speech0", "speech1", "randn", "zeros"]
    x = np.stack([gold[f"{n}.pcm_in"] for n in names])
    pcm, f32 = sess.process(x, want_f32=True)
    for i, n in enumerate(names):
        # vs the reference fixture: 1 LSB + the reference's own fp32-angle DFT-table error (amplified where sum(w^2) is
        # small: no centre padding, hamming ends at 0.08) -> a few LSB at the chunk edges for loud inputs
        d = np.abs(pcm[i].astype(np.int32) - gold[f"{n}.pcm_out"].astype(np.int32))
        assert d[1920:-1920].max() <= 2 and d.max() <= 24, (n, int(d.max()))
    assert not pcm[3].any()
    o = DfsmnOracle(tensors, L, exact_dft=True)
    opcm, of32 = o.process(x)
    assert np.abs(sess.tap("logmel", 4 * 120 * 24).reshape(120, 4, 24)[:, 0] - o.taps["logmel"]).max() <= 2e-4
    assert np.abs(sess.tap("mask", 4 * 961 * 24).reshape(4, 24, 961)[0].T - o.taps["mask"]).max() <= 1e-4
    assert np.abs(f32 - of32).max() <= 2e-5
    assert np.abs(pcm.astype(np.int32) - opcm.astype(np.int32)).max() <= 1


@pytest.mark.gpu
def test_gpu_dfsmn_batch_properties():
    """The reference's default 2 s export (96000 samples, 99 frames) at batch 32: rows are independent calls."""
    import torch
    from audio_denoiser_onnx_amd.session import InferenceSession
    sess = InferenceSession(weights=_blob_bytes(), metadata=_dfsmn_meta(96000))
    assert (sess.in_len, sess.out_len, sess.frames) == (96000, 96000, 99)
    rng = np.random.default_rng(3)
    x = (rng.standard_normal((32, 96000)) * 1500).astype(np.int16)
    pcm, _ = sess.process(x)
    sub, _ = sess.process(x[5:9])
    assert np.array_equal(sub, pcm[5:9])
    d_in = torch.from_numpy(x).cuda()
    d_out = torch.empty((32, 96000), dtype=torch.int16, device="cuda")
    sess.run_device(d_in, d_out)
    assert np.array_equal(d_out.cpu().numpy(), pcm)


@pytest.mark.gpu
def test_gpu_dfsmn_file_driver(tensors, tmp_path):
    """inference_dfsmn: model dir + manifest on disk, ragged 48 kHz file, seeded noise tail; vs the oracle per slice."""
    import wave
    from audio_denoiser_onnx_amd import inference_dfsmn
    from audio_denoiser_onnx_amd.inference_gtcrn import cut_slices, read_wav_int16
    from audio_denoiser_onnx_amd.metadata import write_metadata
    L = 24000
    model = tmp_path / "DFSMN.adew"
    model.write_bytes(_blob_bytes())
    write_metadata(model, _dfsmn_meta(L))
    g = np.load(os.path.join(GOLD, "dfsmn_seed0_io.npz"))
    audio = np.concatenate([g["speech0.pcm_in"], g["speech1.pcm_in"], g["randn.pcm_in"]])[:61000]    # 2.54 slices
    noisy = tmp_path / "in.wav"
    with wave.open(str(noisy), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(48000); w.writeframes(audio.astype("<i2").tobytes())
    out_path = tmp_path / "out.wav"
    assert inference_dfsmn.main([str(model), str(noisy), str(out_path), "--seed", "11"]) == 0
    got = read_wav_int16(out_path, 48000)
    assert got.shape == audio.shape
    slices, _ = cut_slices(audio, L, L, "noise", np.random.default_rng(11))
    ref, _ = DfsmnOracle(tensors, L, exact_dft=True).process(slices)
    assert np.abs(got.astype(np.int32) - ref.reshape(-1)[:len(audio)].astype(np.int32)).max() <= 1


def test_dfsmn_oracle_fold_and_resampling_match_reference(tensors):
    """USE_BATCH_FOLD (3 windows of 9600) and the 16 kHz -> 48 kHz -> 24 kHz interpolation edges of the reference against the oracle."""
    z = np.load(os.path.join(GOLD, "dfsmn_seed0_edges.npz"))
    W = int(z["fold_window_length"])
    pcm, _ = DfsmnOracle(tensors, W).process(z["fold_in"].reshape(-1, W))
    d = pcm.reshape(-1).astype(np.int32) - z["fold_out"].astype(np.int32)
    assert np.abs(d).max() <= 1 and (d != 0).mean() < 0.02
    out = DfsmnOracle(tensors, 24000).process_resampled(z["rs_in"], z["rs_out"].shape[0])
    d = out.astype(np.int32) - z["rs_out"].astype(np.int32)
    assert np.abs(d).max() <= 1 and (d != 0).mean() < 0.02


def test_dfsmn_oracle_dynamic_axes_match_reference(tensors):
    """DYNAMIC_AXES = True (tests/golden/dfsmn_dynamic_seed0.npz = two runs of the reference's forward, tools/make_golden_dfsmn.py --dynamic): a free input length at
    48 kHz, and 22.05 kHz -> 48 kHz -> 16 kHz through the scale-factor edges."""
    from dfsmn_oracle import process_dynamic
    z = np.load(os.path.join(GOLD, "dfsmn_dynamic_seed0.npz"))
    for tag, ri, ro in (("eq", 48000, 48000), ("rs", int(z["rs_in_rate"]), int(z["rs_out_rate"]))):
        out = process_dynamic(tensors, z[tag + "_in"], ri, ro)
        assert out.shape == z[tag + "_out"].shape, tag
        d = out.astype(np.int32) - z[tag + "_out"].astype(np.int32)
        assert np.abs(d).max() <= 1 and (d != 0).mean() < 0.02, tag


@pytest.mark.gpu
def test_gpu_dfsmn_dynamic_axes_match_reference():
    from audio_denoiser_onnx_amd.session import InferenceSession
    z = np.load(os.path.join(GOLD, "dfsmn_dynamic_seed0.npz"))
    for tag, ri, ro, frames in (("eq", 48000, 48000, 9), ("rs", int(z["rs_in_rate"]), int(z["rs_out_rate"]), 12)):
        x, want = z[tag + "_in"], z[tag + "_out"]
        with InferenceSession(weights=_blob_bytes(), metadata=_dfsmn_meta(x.shape[0], ri, ro, dynamic_axes=True)) as sess:
            assert (sess.in_len, sess.out_len, sess.frames) == (x.shape[0], want.shape[0], frames), tag
            out = sess.run(None, {"noisy_audio": np.stack((x, x))[:, None]})[0][:, 0]
        d = out[0].astype(np.int32) - want.astype(np.int32)
        assert np.abs(d).max() <= 2 and (d != 0).mean() < 0.05 and np.array_equal(out[0], out[1]), tag


@pytest.mark.gpu
def test_gpu_dfsmn_fold_and_resampling_match_reference():
    from audio_denoiser_onnx_amd.session import InferenceSession
    z = np.load(os.path.join(GOLD, "dfsmn_seed0_edges.npz"))
    meta = _dfsmn_meta(int(z["fold_input_audio_length"]), use_batch_fold=True, batch_window_seconds=float(z["fold_batch_window_seconds"]))
    assert int(meta["fold_window_length"]) == int(z["fold_window_length"]) and int(meta["export_audio_length"]) == z["fold_in"].shape[0]
    with InferenceSession(weights=_blob_bytes(), metadata=meta) as sess:
        assert sess.in_len == 28800 and sess.out_len == 28800 and sess.frames == 9
        out = sess.run(None, {"noisy_audio": np.stack((z["fold_in"], z["fold_in"]))[:, None]})[0][:, 0]
    d = out[0].astype(np.int32) - z["fold_out"].astype(np.int32)
    assert np.abs(d).max() <= 2 and (d != 0).mean() < 0.05 and np.array_equal(out[0], out[1])
    with InferenceSession(weights=_blob_bytes(), metadata=_dfsmn_meta(z["rs_in"].shape[0], int(z["rs_in_rate"]), int(z["rs_out_rate"]))) as sess:
        assert sess.in_len == 8000 and sess.out_len == 12000 and sess.frames == 24
        out = sess.run(None, {"noisy_audio": z["rs_in"][None, None]})[0][0, 0]
    d = out.astype(np.int32) - z["rs_out"].astype(np.int32)
    assert np.abs(d).max() <= 2 and (d != 0).mean() < 0.05
