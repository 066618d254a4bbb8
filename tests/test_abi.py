"""C-ABI checks that need no GPU: the hipcc-built libade.so loads, exports every symbol include/ade.h declares, and
maps manifest / blob errors onto the status codes (= the reference's exception classes) before touching a device."""
import ctypes as C
import json
import os
import re

import pytest

import __graft_entry__ as ge
from ade_testlib import REPO, default_meta, golden_blob
from audio_denoiser_onnx_amd import _lib
from audio_denoiser_onnx_amd.session import InferenceSession


@pytest.fixture(scope="module")
def lib():
    ge.build()
    return _lib.AdeLibrary(ge.LIB)


def test_exports_every_declared_symbol(lib):
    header = open(os.path.join(REPO, "include", "ade.h")).read()
    declared = set(re.findall(r"\b(ade_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)
    for sym in declared:
        assert getattr(lib.c, sym) is not None


def _create(lib, meta, blob):
    h = C.c_void_p()
    st = lib.c.ade_create(json.dumps(meta).encode(), blob, len(blob), 0, C.byref(h))
    msg = lib.c.ade_last_error(None).decode()
    if h:
        lib.c.ade_destroy(h)
    return st, msg


def test_manifest_errors_map_to_reference_exceptions(lib):
    blob = golden_blob(0)
    meta = default_meta()
    bad = dict(meta); del bad["model_family"]
    st, msg = _create(lib, bad, blob)
    assert st == _lib.ADE_ERR_MISSING_KEY and "model_family" in msg           # KeyError in the reference
    bad = dict(meta, dynamic_axes="maybe")
    assert _create(lib, bad, blob)[0] == _lib.ADE_ERR_BAD_VALUE                # ValueError (_parse_bool)
    bad = dict(meta, export_audio_length="32000")
    assert _create(lib, bad, blob)[0] == _lib.ADE_ERR_SHAPE_MISMATCH           # ValueError (length mismatch)
    bad = dict(meta, model_family="sdaec")                                     # a model folder of the reference that is out of scope here
    assert _create(lib, bad, blob)[0] == _lib.ADE_ERR_UNSUPPORTED
    bad = dict(meta, in_sample_rate="48000")
    assert _create(lib, bad, blob)[0] == _lib.ADE_ERR_UNSUPPORTED
    assert _create(lib, meta, b"not a blob at all")[0] == _lib.ADE_ERR_BAD_VALUE
    h = C.c_void_p()
    assert lib.c.ade_create(None, blob, len(blob), 0, C.byref(h)) == _lib.ADE_ERR_NOT_FOUND   # FileNotFoundError


def test_python_session_raises_reference_exception_classes(lib):
    meta = default_meta()
    with pytest.raises(KeyError):
        InferenceSession(weights=golden_blob(0), metadata={k: v for k, v in meta.items() if k != "opset"}, library=lib)
    with pytest.raises(ValueError):
        InferenceSession(weights=golden_blob(0), metadata=dict(meta, dynamic_axes="perhaps"), library=lib)
    with pytest.raises(FileNotFoundError):
        InferenceSession("/nonexistent/dir/GTCRN.adew", library=lib)


def test_no_gpu_fails_loudly_no_cpu_fallback(lib):
    """Without a HIP device the product library refuses to run (ADE_ERR_DEVICE); it never computes on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    st, msg = _create(lib, default_meta(), golden_blob(0))
    assert st == _lib.ADE_ERR_DEVICE and "no CPU" in msg
    with pytest.raises(_lib.AdeDeviceError):
        InferenceSession(weights=golden_blob(0), metadata=default_meta(), library=lib)


def test_missing_weight_tensor_is_a_key_error(lib):
    import torch
    from audio_denoiser_onnx_amd.weights import pack_blob, unpack_blob
    t = unpack_blob(golden_blob(0))
    t.pop("dpgrnn2.inter_fc.bias")
    st, msg = _create(lib, default_meta(), pack_blob(t))
    if not torch.cuda.is_available():
        assert st == _lib.ADE_ERR_DEVICE           # device probe comes before the arena build
    else:
        assert st == _lib.ADE_ERR_MISSING_KEY and "inter_fc.bias" in msg


def test_model_family_manifest_checks_precede_the_device(lib):
    """Family selection and its rate / fold rules are decided from the manifest alone (no GPU needed to be told 'unsupported')."""
    from audio_denoiser_onnx_amd import melband, mossformer
    blob = b"ADEWGT01"                                             # never parsed: the manifest is rejected first
    st, msg = _create(lib, melband.metadata(13230) | {"in_sample_rate": "48000"}, blob)
    assert st == _lib.ADE_ERR_UNSUPPORTED and "no consistent resampling" in msg
    st, msg = _create(lib, mossformer.metadata(4816, use_batch_fold=True, batch_window_seconds=2408 / 16000.0, in_sample_rate=8000), blob)
    assert st == _lib.ADE_ERR_BAD_VALUE and "equal input/model/output sample rates" in msg
    st, msg = _create(lib, mossformer.metadata(2408) | {"model_sample_rate": "8000"}, blob)
    assert st == _lib.ADE_ERR_UNSUPPORTED and "16000" in msg
    st, msg = _create(lib, mossformer.metadata(2408) | {"model_family": "nkf_aec"}, blob)
    assert st == _lib.ADE_ERR_UNSUPPORTED and "nkf_aec" in msg
    st, msg = _create(lib, mossformer.metadata(2408) | {"model_family": "zipenhancer", "use_batch_fold": "1", "in_sample_rate": "8000"}, blob)
    assert st == _lib.ADE_ERR_BAD_VALUE and "equal input/model/output sample rates" in msg       # ZipEnhancer's fold rule (Export_ZipEnhancer.py:80-81)
    st, msg = _create(lib, melband.metadata(13230) | {"ade_dft_tables": "fast"}, golden_blob(0))
    assert st in (_lib.ADE_ERR_BAD_VALUE, _lib.ADE_ERR_DEVICE)     # the table option is validated after the blob parse; without a GPU the device check comes first
