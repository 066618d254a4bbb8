"""Float audio tensors (IN / OUT_AUDIO_DTYPE F32 / F16) for the families behind the sub-engine interface: every export script carries the two switches and only leaves the
int16 scale steps out for float tensors (UL-UNAS/Export_UL_UNAS.py:45-46, 858-859, 897-912; Mel_Band_Roformer/Stereo/Export_MelBandRoformer.py:55-56, 327-328, 667-680;
DFSMN/Export_DFSMN.py:43-44, 178-182, 241-247; MossFormer2_SS_16K/Export_MossFormer2_SS_16K.py:31-32, 563, 649-657; H-GTCRN/Export_H_GTCRN.py:52-53, 965-966, 1042-1063;
ZipEnhancer/Export_ZipEnhancer.py:35-36, 820-821, 913-926).

tests/golden/<family>_float_io_seed0.npz = the reference's own forward with those switches (tools/make_golden_<family>.py --float-io) on the family's seeded weights:
F32 -> F32, F32 -> INT16 and INT16 -> F32.  CPU: each oracle against them; GPU: the engine through the C ABI (ade_process_f32 for float input)."""
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))
GOLD = os.path.join(HERE, "golden")

CASES = (("f32_f32", "F32", "F32"), ("f32_i16", "F32", "INT16"), ("i16_f32", "INT16", "F32"))


def close_f32(got, ref, rel, tag):
    """fp32 waveforms: max error against the largest sample (the graphs are long fp32 chains; the int16 gates of the same paths are 1 - 2 LSB = 3 - 6e-5 of full scale)."""
    assert got.shape == ref.shape and got.dtype == np.float32, (tag, got.shape, ref.shape, got.dtype)
    err, top = float(np.abs(got - ref).max()), float(np.abs(ref).max())
    assert err <= rel * max(top, 1e-3), (tag, err, top)


def close_i16(got, ref, lsb, tag):
    assert got.shape == ref.shape and got.dtype == np.int16, (tag, got.shape, ref.shape, got.dtype)
    d = np.abs(got.astype(np.int32) - ref.astype(np.int32))
    assert d.max() <= lsb and (d != 0).mean() < 0.10, (tag, d.max(), (d != 0).mean())


def with_dtypes(meta, din, dout):
    meta = dict(meta)
    meta["input_audio_dtype"], meta["output_audio_dtype"] = din, dout
    return meta


def run_cases(make_session, z, shape_in, lsb, rel, fam):
    """The three dtype pairs through InferenceSession.run: (1,) + shape_in tensors as the export declares them."""
    for tag, din, dout in CASES:
        with make_session(din, dout) as sess:
            assert sess.in_dtype == (np.int16 if din == "INT16" else np.float32) and sess.out_dtype == (np.int16 if dout == "INT16" else np.float32)
            src = z["pcm_in"] if din == "INT16" else z["x_in"]
            outs = sess.run(None, {sess.get_inputs()[0].name: src.reshape((1,) + shape_in)})
            got = np.stack([o[0] for o in outs]) if len(outs) > 1 else outs[0][0]
            ref = z[tag].reshape(got.shape)
            (close_i16(got, ref, lsb, fam + ":" + tag) if dout == "INT16" else close_f32(got, ref, rel, fam + ":" + tag))
            if din != "INT16":      # batch rows stay independent clips on the float entry
                other = np.ascontiguousarray(src[..., ::-1] * np.float32(0.5)).reshape((1,) + shape_in)
                both = sess.run(None, {sess.get_inputs()[0].name: np.concatenate((other, src.reshape((1,) + shape_in)))})
                again = np.stack([o[1] for o in both]) if len(both) > 1 else both[0][1]
                assert np.array_equal(again, got), fam + ":" + tag


# ---- UL-UNAS -------------------------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def ulu():
    from audio_denoiser_onnx_amd import ulunas
    z = np.load(os.path.join(GOLD, "ulunas_seed0.npz"))
    state = {str(k): z["w:" + str(k)] for k in z["keys"]}
    return ulunas.fold_state_dict(state), np.load(os.path.join(GOLD, "ulunas_float_io_seed0.npz"))


def test_ulunas_oracle_float_tensors(ulu):
    from audio_denoiser_onnx_amd import ulunas
    from ulunas_oracle import UlunasOracle
    fused, z = ulu
    o = UlunasOracle(fused, ulunas.block_plan(), z["pcm_in"].shape[0])
    wave_f = o.process_wave((z["x_in"] * np.float32(32768.0))[None])[0]          # a float input is the int16 path's value x 2^15, exactly (:858-859)
    wave_i = o.process_wave(z["pcm_in"].astype(np.float32)[None])[0]
    close_f32(wave_f, z["f32_f32"], 2e-5, "f32_f32")
    close_f32(wave_i, z["i16_f32"], 2e-5, "i16_f32")
    close_i16(np.clip(wave_f * np.float32(32767.0), -32768.0, 32767.0).astype(np.int16), z["f32_i16"], 1, "f32_i16")


@pytest.mark.gpu
def test_ulunas_gpu_float_tensors(ulu):
    from audio_denoiser_onnx_amd import ulunas
    from audio_denoiser_onnx_amd.session import InferenceSession
    from audio_denoiser_onnx_amd.weights import pack_blob
    fused, z = ulu
    blob, n = pack_blob(fused), z["pcm_in"].shape[0]
    run_cases(lambda din, dout: InferenceSession(weights=blob, metadata=with_dtypes(ulunas.metadata(n), din, dout)), z, (1, n), 1, 5e-5, "ul_unas")


# ---- Mel-Band-Roformer -------------------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def melb():
    from ade_testlib import melband_fixture_weights
    z, _, w = melband_fixture_weights()
    return z, w, np.load(os.path.join(GOLD, "melband_float_io_seed0.npz"))


def test_melband_oracle_float_tensors(melb):
    from melband_oracle import MelBandOracle
    z0, w, z = melb
    o = MelBandOracle(w, z0["freq_indices"], z0["dim_inputs"], int(z0["frames"]), int(z0["depth"]))
    wave_f = o.process_wave(z["x_in"] * np.float32(32768.0))                       # the 2^-15 of an int16 input lives in the STFT kernel (:327-328): exact either way
    close_f32(wave_f, z["f32_f32"], 5e-5, "f32_f32")
    close_f32(o.process_wave(z["pcm_in"].astype(np.float32)), z["i16_f32"], 5e-5, "i16_f32")
    close_i16(np.clip(wave_f * np.float32(32767.0), -32768.0, 32767.0).astype(np.int16), z["f32_i16"], 1, "f32_i16")


@pytest.mark.gpu
def test_melband_gpu_float_tensors(melb):
    from audio_denoiser_onnx_amd import melband
    from audio_denoiser_onnx_amd.session import InferenceSession
    from audio_denoiser_onnx_amd.weights import pack_blob
    z0, w, z = melb
    blob, n = pack_blob(melband.model_tensors(w)), z["pcm_in"].shape[1]
    run_cases(lambda din, dout: InferenceSession(weights=blob, metadata=with_dtypes(melband.metadata(n), din, dout)), z, (2, n), 2, 1e-4, "mel_band_roformer")


# ---- DFSMN -------------------------------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def dfs():
    from audio_denoiser_onnx_amd.weights import load_blob
    return load_blob(os.path.join(GOLD, "dfsmn_seed0.adew")), np.load(os.path.join(GOLD, "dfsmn_float_io_seed0.npz"))


def dfsmn_meta(length):
    from audio_denoiser_onnx_amd.metadata import build_audio_metadata
    return build_audio_metadata(producer="tests", model_name="DFSMN", task="denoise", model_family="dfsmn", input_audio_length=length, in_sample_rate=48000,
                                out_sample_rate=48000, model_sample_rate=48000, nfft=1920, window_length=1920, hop_length=960, window_type="hamming", center_pad=False,
                                pad_mode="constant", feature_kind="kaldi_fbank_stft")


def test_dfsmn_oracle_float_tensors(dfs):
    from dfsmn_oracle import DfsmnOracle
    tensors, z = dfs
    o = DfsmnOracle(tensors, z["pcm_in"].shape[0])
    wave_f = o._one(z["x_in"] * np.float32(32768.0), False)                       # a float input skips the * INV_INT16 (:178-182): the int16 path's value, exactly
    close_f32(wave_f, z["f32_f32"], 5e-5, "f32_f32")
    close_f32(o._one(z["pcm_in"], False), z["i16_f32"], 5e-5, "i16_f32")
    close_i16(np.clip(wave_f * np.float32(32768.0), -32768.0, 32767.0).astype(np.int16), z["f32_i16"], 1, "f32_i16")


@pytest.mark.gpu
def test_dfsmn_gpu_float_tensors(dfs):
    from audio_denoiser_onnx_amd.session import InferenceSession
    _, z = dfs
    with open(os.path.join(GOLD, "dfsmn_seed0.adew"), "rb") as f:
        blob = f.read()
    n = z["pcm_in"].shape[0]
    # the gates of tests/test_dfsmn.py: 2 LSB inside, 24 LSB in the first / last window (no centre padding: sum(w^2) is small there and amplifies the reference's own
    # fp32-angle DFT-table error)
    for tag, din, dout in CASES:
        with InferenceSession(weights=blob, metadata=with_dtypes(dfsmn_meta(n), din, dout)) as sess:
            src = z["pcm_in"] if din == "INT16" else z["x_in"]
            got = sess.run(None, {"noisy_audio": src[None, None]})[0][0, 0]
        assert got.dtype == (np.int16 if dout == "INT16" else np.float32)
        d = np.abs(got.astype(np.float64) - z[tag].astype(np.float64)) * (1.0 if dout == "INT16" else 32768.0)
        assert d[1920:-1920].max() <= 2.0 and d.max() <= 24.0, (tag, d[1920:-1920].max(), d.max())


# ---- MossFormer2-SS ------------------------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def moss():
    from audio_denoiser_onnx_amd import mossformer
    z0 = np.load(os.path.join(GOLD, "mossformer_seed0_io.npz"))
    spec, scalars = json.loads(str(z0["spec"])), json.loads(str(z0["scalars"]))
    W = z0["pcm_in"].shape[0]
    fused = {n: mossformer.synthetic_tensor(n, s, sc, mossformer.frames_of(W), int(scalars["flash_group_size"])) for n, s, sc in spec}
    return z0, fused, scalars, W, np.load(os.path.join(GOLD, "mossformer_float_io_seed0.npz"))


def test_mossformer_oracle_float_tensors(moss):
    """The reference reads a float input AS IT IS (norm_audio multiplies by 2^-15 whatever the dtype, :403-411) and returns the restored waveform * 2^-15 as the float
    output (:655): a normalised input comes back 2^-15 times smaller and its int16 output is all zeros -- the export's behaviour, restated, not repaired."""
    from audio_denoiser_onnx_amd import mossformer
    from mossformer_oracle import MossFormerOracle
    z0, fused, scalars, W, z = moss
    tensors = dict(fused)
    tensors.update(mossformer.position_tables(mossformer.frames_of(W), int(scalars["rot_dim"])))
    o = MossFormerOracle(tensors, scalars, int(z0["layers"]), W)
    pcm_f = o.process(z["x_in"][None])[0]
    close_f32(o.taps["wav"][0] * np.float32(1.0 / 32768.0), z["f32_f32"], 2e-3, "f32_f32")      # eps = 1e-6 against an rms of ~1e-6: the normalisation itself is ill-scaled here
    assert not z["f32_i16"].any() and not pcm_f.any()
    o.process(z["pcm_in"][None])
    close_f32(o.taps["wav"][0] * np.float32(1.0 / 32768.0), z["i16_f32"], 1e-4, "i16_f32")


@pytest.mark.gpu
def test_mossformer_gpu_float_tensors(moss):
    from audio_denoiser_onnx_amd import mossformer
    from audio_denoiser_onnx_amd.session import InferenceSession
    from audio_denoiser_onnx_amd.weights import pack_blob
    _, fused, scalars, W, z = moss
    blob = pack_blob(mossformer.model_tensors(fused, scalars, W))
    for tag, din, dout in CASES:
        with InferenceSession(weights=blob, metadata=with_dtypes(mossformer.metadata(W), din, dout)) as sess:
            src = z["pcm_in"] if din == "INT16" else z["x_in"]
            outs = sess.run(None, {"mix_audio": src[None, None]})
            got = np.stack([o[0, 0] for o in outs])
            if dout == "INT16":
                assert got.dtype == np.int16 and not got.any() and not z[tag].any()
            else:
                close_f32(got, z[tag], 2e-3 if din == "F32" else 1e-4, "mossformer2_ss:" + tag)


# ---- H-GTCRN ---------------------------------------------------------------------------------------------------------------------------------------------
# H-GTCRN's WPE solve is ill-conditioned in a handful of bins (tests/test_hgtcrn.py): end to end the family is compared with the reference at the
# 8 % RMS its own fp32-vs-fp64 spread allows, and exactly (everything downstream of the solve) against the oracle continued from the engine's own WPE output.
@pytest.fixture(scope="module")
def hg():
    from audio_denoiser_onnx_amd import hgtcrn
    z0 = np.load(os.path.join(GOLD, "hgtcrn_seed0.npz"))
    state = {str(k): z0["w:" + str(k)] for k in z0["keys"]}
    return hgtcrn.fold_state_dict(state), np.load(os.path.join(GOLD, "hgtcrn_float_io_seed0.npz"))


def rms_close(got, ref, tag):
    got, ref = got.astype(np.float64), ref.astype(np.float64)
    assert got.shape == ref.shape
    assert np.sqrt(((got - ref) ** 2).mean()) < 0.08 * np.sqrt((ref ** 2).mean()), tag


def test_hgtcrn_oracle_float_tensors(hg):
    from hgtcrn_oracle import HgtcrnOracle
    fused, z = hg
    o = HgtcrnOracle(fused, z["pcm_in"].shape[1], 1, False)
    wave_f = o.process(z["x_in"][None], float_out=True)[0]
    rms_close(wave_f, z["f32_f32"], "f32_f32")
    rms_close(o.process(z["pcm_in"][None], float_out=True)[0], z["i16_f32"], "i16_f32")
    rms_close(o.process(z["x_in"][None])[0], z["f32_i16"], "f32_i16")
    # a float input is the int16 path's value exactly (x 2^15 x 2^-15): up to the fp32 mean of the call, the same waveform
    assert np.array_equal(z["f32_f32"], z["i16_f32"]) or np.abs(z["f32_f32"] - z["i16_f32"]).max() < 0.1 * np.abs(z["i16_f32"]).max()


@pytest.mark.gpu
def test_hgtcrn_gpu_float_tensors(hg):
    from audio_denoiser_onnx_amd import hgtcrn
    from audio_denoiser_onnx_amd.session import InferenceSession
    from audio_denoiser_onnx_amd.weights import pack_blob
    from hgtcrn_oracle import HgtcrnOracle
    fused, z = hg
    blob, n = pack_blob(fused), z["pcm_in"].shape[1]
    o = HgtcrnOracle(fused, n, 1, True)
    for tag, din, dout in CASES:
        with InferenceSession(weights=blob, metadata=with_dtypes(hgtcrn.metadata(n), din, dout)) as sess:
            src = z["pcm_in"] if din == "INT16" else z["x_in"]
            got = sess.run(None, {"noisy_audio": src[None]})[0][0, 0]
            T = sess.frames
            wpe = sess.tap("wpe", 2 * 514 * T).reshape(1, 2, 2, 257, T)
        rms_close(got, z[tag], "h_gtcrn:" + tag)                                   # against the reference: the family's end-to-end bound
        want = o.process(src[None], inject_wpe=(wpe[:, :, 0], wpe[:, :, 1]), float_out=dout != "INT16")[0]      # exactly, downstream of the solve
        if dout == "INT16":
            close_i16(got, want, 3, "h_gtcrn:" + tag)
        else:
            close_f32(got, want, 1e-4, "h_gtcrn:" + tag)


# ---- ZipEnhancer -----------------------------------------------------------------------------------------------------------------------------------------
# The phase feature atan2(im, re + 1e-5) has a branch cut that the two reflect-padded edge frames sit on (tests/test_zipenhancer.py): which side a low bin of those
# frames falls on depends on the summation order of the STFT, so implementations can differ in a few edge-frame bins.  The dtype switches themselves are exact steps:
# a float input is lifted by * 32768 (:820-821) -- the int16 samples again, bit for bit -- and a float output is the same waveform * 2^-15 (:920-922).
@pytest.fixture(scope="module")
def zipf():
    from audio_denoiser_onnx_amd import zipenhancer as zp
    z0 = np.load(os.path.join(GOLD, "zipenhancer_seed0_io.npz"))
    cfg = zp.ZipConfig.from_tensor(z0["config"])
    return zp.fuse_state_dict(zp.synthetic_state_dict(cfg, int(z0["seed"])), cfg), np.load(os.path.join(GOLD, "zipenhancer_float_io_seed0.npz"))


def test_zipenhancer_reference_dtype_switches_are_exact_steps(zipf):
    _, z = zipf
    assert np.array_equal(z["f32_f32"], z["i16_f32"]) and np.array_equal(z["f32_i16"], z["i16_i16"])
    d = np.abs(np.clip(z["i16_f32"] * np.float32(32768.0), -32768, 32767).astype(np.int16).astype(np.int32) - z["i16_i16"].astype(np.int32))
    assert d.max() == 0


def test_zipenhancer_oracle_float_tensors(zipf):
    from zipenhancer_oracle import ZipEnhancerOracle
    t, z = zipf
    o = ZipEnhancerOracle(t, z["pcm_in"].shape[0], 1)
    out, wave = o.process((z["x_in"] * np.float32(32768.0))[None])[:2]
    got = (np.where(np.isnan(wave[0]), np.float32(0.0), wave[0]) * np.float32(1.0 / 32768.0)).astype(np.float32)
    d = np.abs(got - z["f32_f32"])
    # the bulk to fp32 round-off; the first / last 2 frames (200 samples each side) may sit on the other side of the phase branch cut
    assert np.median(d) < 2e-6 and d[400:-400].max() <= 5e-5 * max(1.0, float(np.abs(z["f32_f32"]).max()) * 32768.0 / 100.0), (np.median(d), d[400:-400].max())
    di = np.abs(out[0].astype(np.int32) - z["f32_i16"].astype(np.int32))
    assert np.median(di) == 0 and di[400:-400].max() <= 1


@pytest.mark.gpu
def test_zipenhancer_gpu_float_tensors(zipf):
    from audio_denoiser_onnx_amd import zipenhancer as zp
    from audio_denoiser_onnx_amd.session import InferenceSession
    from audio_denoiser_onnx_amd.weights import pack_blob
    t, z = zipf
    blob, n = pack_blob(t), z["pcm_in"].shape[0]
    with InferenceSession(weights=blob, metadata=zp.metadata(n)) as sess:
        pcm_i, wave_i = sess.process(z["pcm_in"][None], want_f32=True)             # the int16 handle: PCM and the waveform in int16 units
    for tag, din, dout in CASES:
        with InferenceSession(weights=blob, metadata=with_dtypes(zp.metadata(n), din, dout)) as sess:
            src = z["pcm_in"] if din == "INT16" else z["x_in"]
            got = sess.run(None, {"noisy_audio": src[None, None]})[0][0, 0]
        if dout == "INT16":        # the same samples reach the network: the same PCM, bit for bit
            assert got.dtype == np.int16 and np.array_equal(got, pcm_i[0]), tag
            d = np.abs(got.astype(np.int32) - z[tag].astype(np.int32))
        else:
            assert got.dtype == np.float32 and np.array_equal(got, wave_i[0] * np.float32(1.0 / 32768.0)), tag
            d = np.abs(got - z[tag]) * 32768.0
        assert np.median(d) <= 0.05 and d[400:-400].max() <= 1.0, (tag, float(np.median(d)), float(d[400:-400].max()))      # vs the reference, away from the edge frames


# ---- NaN samples on the float entry --------------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_gpu_nan_samples_follow_the_reference_nan_to_num(zipf, ulu):
    """A float input tensor can carry NaN.  ZipEnhancer's per-window RMS turns the whole window into NaN and the reference returns zeros for it (where(isnan, 0) /
    nan_to_num, Export_ZipEnhancer.py:913-920); UL-UNAS is causal, so the frames before the bad sample are untouched, and a NaN waveform sample is
    nan_to_num'ed to 0 (Export_UL_UNAS.py:906-907).  Neither may turn NaN into -32768."""
    from audio_denoiser_onnx_amd import ulunas, zipenhancer as zp
    from audio_denoiser_onnx_amd.session import InferenceSession
    from audio_denoiser_onnx_amd.weights import pack_blob
    t, z = zipf
    x = z["x_in"].copy()
    x[4000] = np.nan
    n = x.shape[0]
    for dout in ("INT16", "F32"):
        with InferenceSession(weights=pack_blob(t), metadata=with_dtypes(zp.metadata(n), "F32", dout)) as sess:
            both = sess.run(None, {"noisy_audio": np.stack((x, z["x_in"]))[:, None]})[0][:, 0]
        assert not both[0].any() and both[1].any() and np.isfinite(both[1].astype(np.float64)).all(), dout      # the bad window is zeros; its neighbour in the batch is untouched
    fused, zu = ulu
    xu = zu["x_in"].copy()
    xu[2048] = np.nan                                                     # first touched by frame 7 (256 * 7 = 1792 .. 2304 holds sample 2048; reflect padding: centre at 256 t)
    with InferenceSession(weights=pack_blob(fused), metadata=with_dtypes(ulunas.metadata(xu.shape[0]), "F32", "INT16")) as sess:
        got = sess.run(None, {"noisy_audio": xu[None, None]})[0][0, 0]
        clean = sess.run(None, {"noisy_audio": zu["x_in"][None, None]})[0][0, 0]
    assert np.array_equal(got[:1024], clean[:1024]) and np.abs(clean[:1024]).max() > 0      # frames well before the bad sample are the clean run's
    # What the NaN reaches: a NaN waveform sample becomes 0, never the clamp limit.  (The reference's graph keeps NaN through torch.clamp and so zeroes every later
    # sample; the engine's fmaxf-style clamps (log-power floor) drop it, so later frames come back finite -- garbage in, unspecified finite samples out)
    assert (got != -32768).all()


@pytest.mark.gpu
def test_gpu_device_resident_float_entry_matches_the_host_entry(ulu):
    """ade_process_device_f32 on device tensors (the reference's io-binding pattern) = ade_process_f32 on host buffers, bit for bit: a sub-engine family (UL-UNAS, F32 in
    and out) and GTCRN's sandwich (F32 in, INT16 out), on the caller's stream."""
    import torch
    from ade_testlib import default_meta, golden_blob, golden_inputs
    from audio_denoiser_onnx_amd import ulunas
    from audio_denoiser_onnx_amd.session import InferenceSession
    from audio_denoiser_onnx_amd.weights import pack_blob
    fused, z = ulu
    x = np.stack((z["x_in"], z["x_in"][::-1] * np.float32(0.5)))
    with InferenceSession(weights=pack_blob(fused), metadata=with_dtypes(ulunas.metadata(x.shape[1]), "F32", "F32")) as sess:
        want = sess.run(None, {"noisy_audio": x[:, None]})[0][:, 0]
        d_in = torch.from_numpy(np.ascontiguousarray(x)).cuda()
        d_f32 = torch.zeros((2, sess.row_out), dtype=torch.float32, device="cuda")
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            sess.run_device(d_in, None, d_f32, stream=stream.cuda_stream)
        stream.synchronize()
        assert np.array_equal(d_f32.cpu().numpy(), want)
        with pytest.raises(ValueError):
            sess.run_device(d_in.to(torch.int16), None, d_f32)                       # a float-input model takes float32 tensors
    pcm = golden_inputs()["wav0"][:16000]
    xg = (pcm.astype(np.float32) / np.float32(32768.0))[None]
    with InferenceSession(weights=golden_blob(0), metadata=default_meta(16000, input_audio_dtype="F32")) as sess:
        want = sess.run(None, {"noisy_audio": xg[:, None]})[0][:, 0]
        d_in = torch.from_numpy(xg).cuda()
        d_out = torch.zeros((1, sess.row_out), dtype=torch.int16, device="cuda")
        sess.run_device(d_in, d_out)
        assert np.array_equal(d_out.cpu().numpy(), want) and np.abs(want).max() > 100


@pytest.mark.gpu
def test_gpu_float_tensors_with_batch_fold(ulu, melb, zipf, dfs):
    """USE_BATCH_FOLD exports with float tensors: the fold is a reshape inside the graph (e.g. Export_UL_UNAS.py:866-871) and the dtype switches are the exact steps pinned
    above, so a folded F32 -> INT16 handle must return the folded INT16 handle's PCM bit for bit (x / 32768 * 32768 is the int16 sample again), and its F32 output the
    same waveform.  The int16 fold paths are pinned to reference runs in each family's own test file."""
    from audio_denoiser_onnx_amd import melband, ulunas, zipenhancer as zp
    from audio_denoiser_onnx_amd.session import InferenceSession
    from audio_denoiser_onnx_amd.weights import pack_blob
    rng = np.random.default_rng(5)

    def check(blob, meta, channels, fam):
        with InferenceSession(weights=blob, metadata=meta) as si:
            n = si.in_len
            pcm = (rng.standard_normal((1, channels, n)) * 3000).astype(np.int16)
            want = si.run(None, {si.get_inputs()[0].name: pcm})[0]
        x = (pcm.astype(np.float32) / np.float32(32768.0)).astype(np.float32)
        with InferenceSession(weights=blob, metadata=with_dtypes(meta, "F32", "INT16")) as sf:
            got = sf.run(None, {sf.get_inputs()[0].name: x})[0]
        assert np.array_equal(got, want) and np.abs(want).max() > 0, fam
        with InferenceSession(weights=blob, metadata=with_dtypes(meta, "F32", "F32")) as sf:
            wave = sf.run(None, {sf.get_inputs()[0].name: x})[0]
        scale = 32768.0 if fam == "dfsmn" else 32767.0
        d = np.abs(np.clip(wave.astype(np.float64) * scale, -32768, 32767).astype(np.int64) - want.astype(np.int64))
        assert d.max() <= 1, (fam, d.max())                       # the float output is the waveform the PCM was cast from (ZipEnhancer: * 2^-15 of int16 units, hence <= 1)

    check(pack_blob(ulu[0]), ulunas.metadata(10000, use_batch_fold=True, batch_window_seconds=0.256), 1, "ul_unas")
    check(pack_blob(melband.model_tensors(melb[1])), melband.metadata(30000, use_batch_fold=True, batch_window_seconds=0.3), 2, "mel_band_roformer")
    check(pack_blob(zipf[0]), zp.metadata(12000, use_batch_fold=True, batch_window_seconds=0.5), 1, "zipenhancer")
    with open(os.path.join(GOLD, "dfsmn_seed0.adew"), "rb") as f:
        blob = f.read()
    meta = dfsmn_meta(20000)
    meta.update({"use_batch_fold": "1", "batch_window_seconds": "0.2", "fold_window_length": "9600", "export_audio_length": "28800"})
    check(blob, meta, 1, "dfsmn")


@pytest.mark.gpu
def test_gpu_hgtcrn_float_tensors_with_batch_fold(hg):
    """H-GTCRN's batch-fold export with a float input: one mean per call, then the same window gather as the int16 entry (Export_H_GTCRN.py:963-981).  The samples that
    reach the network are the int16 entry's, so the two handles must agree bit for bit (the int16 fold path is pinned in tests/test_hgtcrn.py::test_gpu_fold)."""
    from audio_denoiser_onnx_amd import hgtcrn
    from audio_denoiser_onnx_amd.session import InferenceSession
    from audio_denoiser_onnx_amd.weights import pack_blob
    fused, _ = hg
    zf = np.load(os.path.join(GOLD, "hgtcrn_seed0_fold.npz"))
    pcm = zf["pcm_in"][None]
    meta = hgtcrn.metadata(20000, use_batch_fold=True, batch_window_seconds=0.512)
    with InferenceSession(weights=pack_blob(fused), metadata=meta) as si:
        want = si.run(None, {"noisy_audio": pcm})[0]
    x = (pcm.astype(np.float32) / np.float32(32768.0)).astype(np.float32)
    with InferenceSession(weights=pack_blob(fused), metadata=with_dtypes(meta, "F32", "INT16")) as sf:
        got = sf.run(None, {"noisy_audio": x})[0]
    d = np.abs(got.astype(np.int32) - want.astype(np.int32))
    assert d.max() <= 3 and (d != 0).mean() < 0.05, (d.max(), (d != 0).mean())


@pytest.mark.gpu
@pytest.mark.parametrize("in_rate", [8000, 16000, 32000])
def test_gpu_hgtcrn_float_input_equals_int16_input_at_every_input_rate(hg, in_rate):
    """A float input tensor carries the int16 entry's samples (x 2^-15), so both entries must produce the same PCM -- also BELOW the model rate, where the DC mean is
    taken from the caller-rate samples before the interpolation (Export_H_GTCRN.py:953-964): the float entry must take it from the float tensor, not from the
    pointer it was handed in place of PCM."""
    from audio_denoiser_onnx_amd import hgtcrn
    from audio_denoiser_onnx_amd.session import InferenceSession
    from audio_denoiser_onnx_amd.weights import pack_blob
    fused, _ = hg
    L = 20480 * in_rate // 16000
    rng = np.random.default_rng(5)
    pcm = (rng.standard_normal((1, 2, L)) * 3000 + 700).clip(-32768, 32767).astype(np.int16)       # a DC offset, so that a wrong mean shows
    meta = hgtcrn.metadata(L, in_sample_rate=in_rate)
    with InferenceSession(weights=pack_blob(fused), metadata=meta) as si:
        want = si.run(None, {"noisy_audio": pcm})[0]
    x = (pcm.astype(np.float32) / np.float32(32768.0)).astype(np.float32)
    with InferenceSession(weights=pack_blob(fused), metadata=with_dtypes(meta, "F32", "INT16")) as sf:
        got = sf.run(None, {"noisy_audio": x})[0]
    d = np.abs(got.astype(np.int32) - want.astype(np.int32))
    assert d.max() <= 3 and (d != 0).mean() < 0.05, (in_rate, d.max(), (d != 0).mean())
