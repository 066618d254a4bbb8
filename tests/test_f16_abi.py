"""IEEE-half audio tensors at the C ABI (ade_process_f16 / ade_process_device_f16; manifest input_audio_dtype / output_audio_dtype "F16", GTCRN/Export_GTCRN.py:47-48,
645-646, 691-693): the graph computes in fp32, so a half input is widened exactly and the half output is the fp32 output rounded to nearest even -- the native entry
must equal the fp32 entry fed the widened tensor, narrowed with numpy's own rounding, bit for bit."""
import numpy as np
import pytest

from ade_testlib import default_meta, golden_blob, golden_inputs, hipsim_library
from audio_denoiser_onnx_amd.session import InferenceSession


def with_dtypes(din, dout, length=16000):
    meta = dict(default_meta(length))
    meta["input_audio_dtype"], meta["output_audio_dtype"] = din, dout
    return meta


def check(library):
    ins = golden_inputs()
    pcm = np.stack([ins["wav0"], ins["randn"]])[:, None, :]
    x16 = (pcm.astype(np.float32) / np.float32(32768.0)).astype(np.float16)
    # half in -> half out, against the fp32 entry on the widened samples
    with InferenceSession(weights=golden_blob(0), metadata=with_dtypes("F16", "F16"), library=library) as s:
        assert s.get_inputs()[0].type == "tensor(float16)" and s.get_outputs()[0].type == "tensor(float16)"
        got = s.run(None, {"noisy_audio": x16})[0]
        assert got.dtype == np.float16 and got.shape == (2, 1, s.out_len)
        _, f32 = s.process_f32(x16.reshape(2, -1).astype(np.float32), want_pcm=False)
        assert np.array_equal(got.reshape(2, -1).view(np.uint16), f32.astype(np.float16).view(np.uint16))
        assert np.abs(got.astype(np.float32)).max() > 1e-3
    # int16 in -> half out ; half in -> int16 out
    with InferenceSession(weights=golden_blob(0), metadata=with_dtypes("INT16", "F16"), library=library) as s:
        got = s.run(None, {"noisy_audio": pcm})[0]
        _, f32 = s.process(pcm.reshape(2, -1), want_f32=True)
        assert got.dtype == np.float16 and np.array_equal(got.reshape(2, -1).view(np.uint16), f32.astype(np.float16).view(np.uint16))
    with InferenceSession(weights=golden_blob(0), metadata=with_dtypes("F16", "INT16"), library=library) as s:
        got = s.run(None, {"noisy_audio": x16})[0]
        want, _ = s.process_f32(x16.reshape(2, -1).astype(np.float32), want_f32=False)
        assert got.dtype == np.int16 and np.array_equal(got.reshape(2, -1), want)


@pytest.mark.hipsim
def test_hipsim_half_tensors_equal_the_fp32_entry():
    check(hipsim_library())


@pytest.mark.gpu
def test_gpu_half_tensors_equal_the_fp32_entry():
    check(None)


@pytest.mark.gpu
def test_gpu_half_device_entry_and_rounding_edges():
    """ade_process_device_f16 on device tensors; the narrowing on values that sit on rounding boundaries (ties to even, sub-normals, overflow)."""
    import ctypes as C
    import torch
    ins = golden_inputs()
    pcm = np.stack([ins["wav0"]])
    x16 = (pcm.astype(np.float32) / np.float32(32768.0)).astype(np.float16)
    with InferenceSession(weights=golden_blob(0), metadata=with_dtypes("F16", "F16")) as s:
        s.reserve(1)
        d_in = torch.from_numpy(x16.view(np.int16)).cuda()
        d_out = torch.empty((1, s.row_out), dtype=torch.int16, device="cuda")
        st = s._lib.c.ade_process_device_f16(s._h, C.c_void_p(d_in.data_ptr()), 1, None, C.c_void_p(d_out.data_ptr()), None)
        s._lib.check(st, s._h)
        want = s.run(None, {"noisy_audio": x16[:, None, :]})[0].reshape(1, -1)
        assert np.array_equal(d_out.cpu().numpy().view(np.uint16), want.view(np.uint16))
