"""GTCRN_CUSTOM's input / output sandwich (GTCRN/Export_GTCRN.py:636-693): float audio, other input / output sample rates and the dynamic-length export
(frame count from the model-rate waveform, ISTFT trim of STFT_Process.py:337-341).

Fixture: tests/golden/gtcrn_sandwich_seed0.npz -- the reference class itself, run on the seed-0 weights (tools/make_golden_gtcrn_sandwich.py).
The oracle (oracle/gtcrn_sandwich.py around oracle/ade_oracle.c) is pinned to it on the CPU; the engine (multi-kernel sequence behind libade's C ABI) to the oracle
and to the fixture on the host simulator and on the GPU.

Tolerances.  The dynamic export divides its last hop by w^2 of a sqrt-hann window that ends at 3.8e-5: the final samples are amplified by up to 2.6e4 (the int16
fixtures saturate there), so the last 256 model-rate samples are compared RELATIVELY and everything before them like the static path: <= 1 LSB int16, 1e-4 float
(the engine's exact FFT vs the reference's fp32-angle DFT tables, as for the static GTCRN path)."""
import os
import sys

import numpy as np
import pytest

from ade_testlib import GOLD, golden_blob, hipsim_library
from audio_denoiser_onnx_amd.metadata import build_audio_metadata
from audio_denoiser_onnx_amd.session import InferenceSession
from oracle_lib import GtcrnOracle

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import gtcrn_sandwich  # noqa: E402


def _cases():
    z = np.load(os.path.join(GOLD, "gtcrn_sandwich_seed0.npz"))
    return z, [str(n) for n in z["names"]]


Z, NAMES = _cases()


def _cfg(name):
    L, sri, sro, fin, fout, dyn = (int(v) for v in Z[name + ":cfg"])
    return L, sri, sro, bool(fin), bool(fout), bool(dyn)


def _tail_split(name):
    """(samples before the dynamic tail, at the output rate) -- everything for a static export."""
    L, sri, sro, fin, fout, dyn = _cfg(name)
    n = Z[name + ":out"].size
    if not dyn:
        return n
    lm = gtcrn_sandwich.model_length(L, sri)
    head_model = 256 * (lm // 256)                       # 256 (T - 1): the samples two frames overlap on
    return int(np.floor(head_model * (sro / 16000.0))) - 4   # (a few output samples around the seam interpolate into the tail)


def _check(name, got, where):
    want = Z[name + ":out"]
    assert got.shape == want.shape and got.dtype == want.dtype, (where, got.shape, want.shape, got.dtype)
    cut = _tail_split(name)
    g, w = got.astype(np.float64), want.astype(np.float64)
    if want.dtype == np.int16:
        assert np.abs(g[:cut] - w[:cut]).max() <= 1, (where, name, np.abs(g[:cut] - w[:cut]).max())
    else:
        assert np.abs(g[:cut] - w[:cut]).max() <= 1e-4, (where, name, np.abs(g[:cut] - w[:cut]).max())
    if cut < want.size:                                   # the amplified tail: relative, where the reference is not saturated
        t_g, t_w = g[cut:], w[cut:]
        ok = np.abs(t_w) < 32767 if want.dtype == np.int16 else np.ones(t_w.shape, bool)
        # 1 / w^2 grows towards the end, so an error of the un-normalised overlap-add of 1e-6 (this engine's exact FFT against the reference's fp32-angle tables)
        # becomes up to ~3e-2 there: bounded against the sample itself AND the tail's own scale
        tol = 5e-3 * np.abs(t_w) + 1e-3 * np.abs(t_w[ok]).max() + (1.0 if want.dtype == np.int16 else 0.0)
        assert np.all(np.abs(t_g - t_w)[ok] <= tol[ok]), (where, name, float(np.max((np.abs(t_g - t_w) / tol)[ok])))


def _oracle_out(name):
    L, sri, sro, fin, fout, dyn = _cfg(name)
    o = GtcrnOracle(golden_blob(0), gtcrn_sandwich.model_length(L, sri))
    return gtcrn_sandwich.forward(lambda x: o.process_model_f32(x, dyn)[0], Z[name + ":in"], sri, sro, fout)


@pytest.mark.parametrize("name", NAMES)
def test_oracle_matches_reference(name):
    _check(name, _oracle_out(name), "oracle")


def _meta(name):
    L, sri, sro, fin, fout, dyn = _cfg(name)
    return build_audio_metadata(producer="test", model_name="GTCRN", task="denoise", model_family="gtcrn", input_audio_length=L, in_sample_rate=sri,
                                out_sample_rate=sro, model_sample_rate=16000, dynamic_axes=dyn, input_audio_dtype="F32" if fin else "INT16",
                                output_audio_dtype="F32" if fout else "INT16")


def _engine_out(name, library=None):
    with InferenceSession(weights=golden_blob(0), metadata=_meta(name), library=library) as sess:
        assert sess.out_len == Z[name + ":out"].size
        return sess.run(None, {"noisy_audio": Z[name + ":in"][None, None, :]})[0][0, 0]


@pytest.mark.hipsim
@pytest.mark.parametrize("name", ["f32_static", "dyn_48k_to_8k", "dyn_22500_f32_to_44000"])
def test_hipsim_engine_matches_reference_and_oracle(name):
    got = _engine_out(name, hipsim_library())
    _check(name, got, "engine (host simulator)")


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_gpu_engine_matches_reference_and_oracle(name):
    got = _engine_out(name)
    _check(name, got, "engine")
    want = _oracle_out(name).astype(np.float64)
    cut = _tail_split(name)
    assert np.abs(got.astype(np.float64)[:cut] - want[:cut]).max() <= (1 if got.dtype == np.int16 else 1e-4)


@pytest.mark.gpu
def test_gpu_sandwich_batch_rows_are_independent_calls():
    name = "dyn_48k_to_8k"
    L, sri, sro, fin, fout, dyn = _cfg(name)
    rng = np.random.default_rng(4)
    rows = np.stack([Z[name + ":in"]] + [(rng.standard_normal(L) * 3000).astype(np.int16) for _ in range(6)])
    with InferenceSession(weights=golden_blob(0), metadata=_meta(name)) as sess:
        a = sess.run(None, {"noisy_audio": rows[:, None, :]})[0]
        b = sess.run(None, {"noisy_audio": rows[::-1].copy()[:, None, :]})[0]
    assert np.array_equal(a[::-1], b)
    _check(name, a[0, 0], "engine, batch row 0")


def test_static_export_at_other_rates_is_refused():
    """The static export sizes its frame count from the input-rate length (Export_GTCRN.py:45): only dynamic_axes = 1 is self-consistent at other rates."""
    from audio_denoiser_onnx_amd import _lib
    meta = build_audio_metadata(producer="test", model_name="GTCRN", task="denoise", model_family="gtcrn", input_audio_length=48000, in_sample_rate=48000,
                                out_sample_rate=16000, model_sample_rate=16000)
    with pytest.raises(Exception) as ei:
        InferenceSession(weights=golden_blob(0), metadata=meta, library=hipsim_library())
    assert "dynamic_axes" in str(ei.value)


@pytest.mark.hipsim
def test_hipsim_file_driver_with_float_tensors(tmp_path):
    """The driver feeds a float model the int16 samples cast straight to its input dtype and returns / writes its float output (Inference_GTCRN_ONNX.py:133-135,
    336-340); slices, stride and trim as for int16."""
    from audio_denoiser_onnx_amd.inference_gtcrn import denoise, write_wav_float32
    rng = np.random.default_rng(9)
    audio = (rng.standard_normal(20000) * 2000).astype(np.int16)
    with InferenceSession(weights=golden_blob(0), metadata=_meta("f32_static"), library=hipsim_library()) as sess:
        assert sess.get_inputs()[0].type == "tensor(float)" and sess.in_dtype == np.float32 and sess.out_dtype == np.float32
        out = denoise(sess, audio)
        one = sess.run(None, {"noisy_audio": audio[:16000].astype(np.float32)[None, None, :]})[0][0, 0]
    assert out.dtype == np.float32 and out.shape == (20000,) and np.isfinite(out).all()
    assert np.array_equal(out[:15872], one)                     # slice 0 of the file = the same call
    write_wav_float32(tmp_path / "o.wav", out, 16000)
    import scipy.io.wavfile as wavfile
    rate, back = wavfile.read(tmp_path / "o.wav")
    assert rate == 16000 and back.dtype == np.float32 and np.array_equal(back, out)


def test_export_manifest_carries_the_io_switches():
    """`export.py --dynamic --in-rate 48000 --out-rate 8000 --in-dtype F32` stamps what the engine reads (the checkpoint side is tests/test_host_logic.py's)."""
    from audio_denoiser_onnx_amd import export
    meta = export.build_audio_metadata(producer="t", model_name="GTCRN", task="denoise", model_family="gtcrn", input_audio_length=48000, in_sample_rate=48000,
                                       out_sample_rate=8000, model_sample_rate=16000, dynamic_axes=True, input_audio_dtype="F32", output_audio_dtype="INT16")
    assert meta["dynamic_axes"] == "1" and meta["input_audio_dtype"] == "F32" and meta["in_sample_rate"] == "48000" and meta["out_sample_rate"] == "8000"


# ---- float input tensors on a batch-fold export (Export_GTCRN.py:41, :645-660): the whole call is centred, then folded ------------------------------------------------
def _fold_f32():
    g = np.load(os.path.join(GOLD, "gtcrn_seed0_fold_f32.npz"))
    pcm = np.round(g["x_in"].astype(np.float64) * 32768.0).astype(np.int16)        # the fixture's input is int16 / 32768: the same samples, exactly
    assert np.array_equal(pcm.astype(np.float32) / np.float32(32768.0), g["x_in"])
    return g, pcm


def test_oracle_batch_fold_with_float_input_matches_reference():
    from oracle_lib import GtcrnOracle
    g, pcm = _fold_f32()
    o = GtcrnOracle(golden_blob(0), int(g["fold_window_length"]))
    opcm, of32 = o.process_fold(pcm[None], 3, threads=2)
    assert np.abs(opcm[0].astype(np.int32) - g["f32_i16"].astype(np.int32)).max() <= 1
    assert np.abs(of32[0] - g["f32_f32"]).max() <= 2e-5


@pytest.mark.gpu
def test_gpu_batch_fold_with_float_tensors():
    from audio_denoiser_onnx_amd.metadata import build_audio_metadata
    from audio_denoiser_onnx_amd.session import InferenceSession
    g, pcm = _fold_f32()
    for dout, key in (("INT16", "f32_i16"), ("F32", "f32_f32")):
        meta = build_audio_metadata(producer="tests", model_name="GTCRN", task="denoise", model_family="gtcrn", input_audio_length=int(g["input_audio_length"]),
                                    use_batch_fold=True, batch_window_seconds=float(g["batch_window_seconds"]), input_audio_dtype="F32", output_audio_dtype=dout)
        with InferenceSession(weights=golden_blob(0), metadata=meta) as sess:
            assert (sess.in_len, sess.out_len, sess.frames) == (12288, 12288, 17)
            x2 = np.stack((g["x_in"], g["x_in"][::-1] * np.float32(0.5)))
            got = sess.run(None, {"noisy_audio": x2[:, None]})[0][:, 0]
            alone = sess.run(None, {"noisy_audio": x2[:1, None]})[0][0, 0]
        assert np.array_equal(alone, got[0])                                         # calls are independent
        if dout == "INT16":
            assert np.abs(got[0].astype(np.int32) - g[key].astype(np.int32)).max() <= 1
        else:
            assert np.abs(got[0] - g[key]).max() <= 2e-5
    # the int16 fold handle on the same samples: the same PCM, bit for bit (x * 32768 is the int16 sample again... but the float path centres in fp32 on normalised
    # samples exactly as the int16 path does after its 2^-15 scale)
    meta = build_audio_metadata(producer="tests", model_name="GTCRN", task="denoise", model_family="gtcrn", input_audio_length=int(g["input_audio_length"]),
                                use_batch_fold=True, batch_window_seconds=float(g["batch_window_seconds"]))
    with InferenceSession(weights=golden_blob(0), metadata=meta) as sess:
        ref = sess.run(None, {"noisy_audio": pcm[None, None]})[0][0, 0]
    assert np.abs(ref.astype(np.int32) - g["f32_i16"].astype(np.int32)).max() <= 1
