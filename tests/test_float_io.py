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
    from audio_denoiser_onnx_amd import weightgen
    z = np.load(os.path.join(GOLD, "melband_seed0_io.npz"))
    spec = [(n, s, sc) for n, s, sc in json.loads(str(z["spec"]))]
    return z, weightgen.materialise(spec), np.load(os.path.join(GOLD, "melband_float_io_seed0.npz"))


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
