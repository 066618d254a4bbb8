"""ctypes binding of libade (include/ade.h).

The product path is the hipcc-built ``libade.so`` next to this file and nothing else: if it is missing the
import of :func:`get_library` raises — there is no CPU or PyTorch fallback for the hot path.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(PKG_DIR, "libade.so")

ADE_OK = 0
ADE_ERR_NOT_FOUND = 1
ADE_ERR_MISSING_KEY = 2
ADE_ERR_SHAPE_MISMATCH = 3
ADE_ERR_BAD_VALUE = 4
ADE_ERR_DEVICE = 5
ADE_ERR_UNSUPPORTED = 6


class AdeDeviceError(RuntimeError):
    """No usable gfx950 device / HIP failure (ADE_ERR_DEVICE)."""


class AdeUnsupportedError(NotImplementedError):
    """Configuration this build does not implement (ADE_ERR_UNSUPPORTED)."""


# status -> the exception class the reference raises at the same boundary (audio_onnx_metadata.py:251-351)
_EXC = {
    ADE_ERR_NOT_FOUND: FileNotFoundError,
    ADE_ERR_MISSING_KEY: KeyError,
    ADE_ERR_SHAPE_MISMATCH: ValueError,
    ADE_ERR_BAD_VALUE: ValueError,
    ADE_ERR_DEVICE: AdeDeviceError,
    ADE_ERR_UNSUPPORTED: AdeUnsupportedError,
}


class IoDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "abi_version", "in_channels", "out_channels", "n_outputs", "in_len", "out_len", "in_sample_rate",
        "out_sample_rate", "model_sample_rate", "frames", "max_batch", "device")]


class StftConfig(C.Structure):
    """``ade_stft_config`` (include/ade.h)."""
    _fields_ = [("n_fft", C.c_int), ("win_length", C.c_int), ("hop", C.c_int), ("window", C.c_char_p),
                ("synthesis_window", C.c_char_p), ("center_pad", C.c_int), ("pad_mode", C.c_char_p)]


EXPORTED_SYMBOLS = (
    "ade_create", "ade_get_io", "ade_process", "ade_submit", "ade_wait", "ade_process_device", "ade_process_f32", "ade_process_device_f32", "ade_process_f16", "ade_process_device_f16", "ade_stitch_device", "ade_reserve", "ade_set_option", "ade_debug_tap",
    "ade_kernel_count", "ade_kernel_name", "ade_profile_last", "ade_kernel_ms", "ade_last_error", "ade_destroy",
    "ade_stft_forward", "ade_istft_forward",
    "ade_stft_create", "ade_stft_frames", "ade_stft_output_length", "ade_stft_keep_tail", "ade_stft_analyze", "ade_stft_synthesize", "ade_stft_synthesize_polar",
    "ade_stream_create", "ade_stream_push", "ade_stream_push_device", "ade_stream_flush", "ade_stream_reset", "ade_stream_destroy",
    "ade_stft_last_error", "ade_stft_destroy",
)


class AdeLibrary:
    """A loaded libade with typed entry points."""

    def __init__(self, path: str = DEFAULT_LIB):
        if not os.path.exists(path):
            raise ImportError(
                f"{path} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'). "
                "The engine has no CPU fallback.")
        self.path = path
        self.is_simulator = "hipsim" in os.path.basename(path)     # tests/hipsim build: device memory is host memory (TEST-ONLY library)
        # One HIP runtime per process: the PyTorch wheel bundles its own libamdhip64/libhsa-runtime64, and a second
        # copy loaded later reports "No HIP GPUs are available".  Loading torch's first makes libade's
        # DT_NEEDED libamdhip64.so.7 resolve to the copy already in the process (torch is this package's
        # device-memory / stream / torch.distributed plumbing, not its compute path).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(path)
        self.c = L
        L.ade_create.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]
        L.ade_get_io.argtypes = [C.c_void_p, C.POINTER(IoDesc)]
        L.ade_stream_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.ade_stream_push.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ade_stream_push_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ade_stream_flush.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.ade_stream_reset.argtypes = [C.c_void_p]
        L.ade_stream_destroy.argtypes = [C.c_void_p]
        L.ade_stream_destroy.restype = None
        L.ade_process.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.ade_process_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ade_submit.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]
        L.ade_wait.argtypes = [C.c_void_p, C.c_uint64]
        L.ade_process_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.ade_process_device_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ade_stitch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ade_process_f16.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.ade_process_device_f16.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ade_reserve.argtypes = [C.c_void_p, C.c_int]
        L.ade_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
        L.ade_debug_tap.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.ade_kernel_count.argtypes = [C.c_void_p]
        L.ade_kernel_name.argtypes = [C.c_void_p, C.c_int]
        L.ade_kernel_name.restype = C.c_char_p
        L.ade_profile_last.argtypes = [C.c_void_p, C.c_int]
        L.ade_kernel_ms.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int)]
        L.ade_last_error.argtypes = [C.c_void_p]
        L.ade_last_error.restype = C.c_char_p
        L.ade_destroy.argtypes = [C.c_void_p]
        L.ade_destroy.restype = None
        L.ade_stft_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.ade_istft_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.ade_stft_create.argtypes = [C.POINTER(StftConfig), C.c_int, C.POINTER(C.c_void_p)]
        L.ade_stft_frames.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.ade_stft_output_length.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.ade_stft_keep_tail.argtypes = [C.c_void_p, C.c_int]
        L.ade_stft_analyze.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.ade_stft_synthesize.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.ade_stft_synthesize_polar.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.ade_stft_synthesize_polar.restype = C.c_int
        L.ade_stft_last_error.argtypes = [C.c_void_p]
        L.ade_stft_last_error.restype = C.c_char_p
        L.ade_stft_destroy.argtypes = [C.c_void_p]
        L.ade_stft_destroy.restype = None

    def check(self, status: int, handle: Optional[C.c_void_p]) -> None:
        if status == ADE_OK:
            return
        msg = self.c.ade_last_error(handle if handle else None)
        text = msg.decode("utf-8", "replace") if msg else f"libade status {status}"
        raise _EXC.get(status, RuntimeError)(text)


def raise_for_status(status: int, text: str) -> None:
    """Raise the reference-equivalent exception class for a libade status (same mapping as ``AdeLibrary.check``)."""
    raise _EXC.get(status, RuntimeError)(text)


_default: Optional[AdeLibrary] = None


def get_library() -> AdeLibrary:
    """The product library (in-tree ``libade.so`` built for gfx950). Raises ImportError when absent."""
    global _default
    if _default is None:
        _default = AdeLibrary(DEFAULT_LIB)
    return _default
