"""ctypes binding of the CPU oracle (oracle/libade_oracle.so).  TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module;
the product package never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(REPO, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "libade_oracle.so")


def build_oracle(force: bool = False) -> str:
    src = os.path.join(ORACLE_DIR, "ade_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", ORACLE_DIR, "-B", "libade_oracle.so"], check=True, capture_output=True)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build_oracle()
        L = C.CDLL(LIB_PATH)
        L.ade_oracle_create.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]
        L.ade_oracle_destroy.argtypes = [C.c_void_p]
        L.ade_oracle_destroy.restype = None
        L.ade_oracle_set_exact_dft.argtypes = [C.c_void_p, C.c_int]
        L.ade_oracle_set_exact_dft.restype = None
        L.ade_oracle_in_len.argtypes = [C.c_void_p]
        L.ade_oracle_out_len.argtypes = [C.c_void_p]
        L.ade_oracle_last_error.restype = C.c_char_p
        L.ade_oracle_process.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.ade_oracle_process_fold.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.ade_oracle_process_model_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.ade_oracle_set_generic_exact_dft.argtypes = [C.c_int]
        L.ade_oracle_set_generic_exact_dft.restype = None
        L.ade_oracle_tap.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_size_t)]
        L.ade_oracle_stft.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int,
                                      C.c_char_p, C.c_void_p, C.POINTER(C.c_int)]
        L.ade_oracle_istft.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int,
                                       C.c_void_p, C.POINTER(C.c_int)]
        _lib = L
    return _lib


class OracleError(RuntimeError):
    pass


class GtcrnOracle:
    """The reference's chunk call, restated in C: int16 [B, in_len] -> int16 [B, out_len]."""

    def __init__(self, blob: bytes, in_len: int = 16000):
        self._h = C.c_void_p()
        self._blob = bytes(blob)
        if lib().ade_oracle_create(self._blob, len(self._blob), in_len, C.byref(self._h)) != 0:
            raise OracleError(lib().ade_oracle_last_error().decode())
        self.in_len = lib().ade_oracle_in_len(self._h)
        self.out_len = lib().ade_oracle_out_len(self._h)
        self.T = self.in_len // 256 + 1

    def set_exact_dft(self, exact: bool):
        """Test knob: exact DFT tables (NOT the reference's arithmetic) to isolate kernel error from table error."""
        lib().ade_oracle_set_exact_dft(self._h, int(bool(exact)))

    def close(self):
        if self._h:
            lib().ade_oracle_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def process(self, pcm: np.ndarray, threads: int = 1):
        pcm = np.ascontiguousarray(pcm, dtype=np.int16).reshape(-1, self.in_len)
        B = pcm.shape[0]
        out_pcm = np.empty((B, self.out_len), np.int16)
        out_f32 = np.empty((B, self.out_len), np.float32)
        rc = lib().ade_oracle_process(self._h, pcm.ctypes.data, B, out_pcm.ctypes.data, out_f32.ctypes.data, threads)
        if rc != 0:
            raise OracleError(lib().ade_oracle_last_error().decode())
        return out_pcm, out_f32

    def process_model_f32(self, x: np.ndarray, dynamic_tail: bool = False) -> np.ndarray:
        """The network between the halves of GTCRN_CUSTOM's sandwich: final fp32 model-rate waveforms (B, in_len) -> normalised waveforms
        (B, out_len), or (B, 256 T) with the dynamic-length export's ISTFT trim."""
        x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, self.in_len)
        keep = 256 * self.T if dynamic_tail else self.out_len
        out = np.empty((x.shape[0], keep), np.float32)
        if lib().ade_oracle_process_model_f32(self._h, x.ctypes.data, x.shape[0], out.ctypes.data, int(bool(dynamic_tail))) != 0:
            raise OracleError(lib().ade_oracle_last_error().decode())
        return out

    def process_fold(self, pcm: np.ndarray, n_win: int, threads: int = 1):
        """USE_BATCH_FOLD calls: pcm (n_calls, n_win * in_len) -> (n_calls, n_win * out_len); mean over the whole call."""
        pcm = np.ascontiguousarray(pcm, dtype=np.int16).reshape(-1, n_win * self.in_len)
        n_calls = pcm.shape[0]
        out_pcm = np.empty((n_calls, n_win * self.out_len), np.int16)
        out_f32 = np.empty((n_calls, n_win * self.out_len), np.float32)
        rc = lib().ade_oracle_process_fold(self._h, pcm.ctypes.data, n_calls, n_win, out_pcm.ctypes.data, out_f32.ctypes.data, threads)
        if rc != 0:
            raise OracleError(lib().ade_oracle_last_error().decode())
        return out_pcm, out_f32

    def tap(self, name: str) -> np.ndarray:
        p = C.POINTER(C.c_float)()
        n = C.c_size_t()
        if lib().ade_oracle_tap(self._h, name.encode(), C.byref(p), C.byref(n)) != 0:
            raise OracleError(lib().ade_oracle_last_error().decode())
        return np.ctypeslib.as_array(p, shape=(n.value,)).copy()


def oracle_set_generic_exact_dft(exact: bool) -> None:
    """Exact (double-angle) DFT tables in oracle_stft / oracle_istft instead of the reference's fp32-angle tables."""
    lib().ade_oracle_set_generic_exact_dft(int(bool(exact)))


def oracle_stft(x: np.ndarray, n_fft, win_length, hop, window, center=True, pad_mode="reflect"):
    x = np.ascontiguousarray(x, np.float32)
    B, L = x.shape
    F = n_fft // 2 + 1
    T = ((L + n_fft if center else L) - n_fft) // hop + 1
    out = np.empty((B, 2 * F, T), np.float32)
    t = C.c_int()
    rc = lib().ade_oracle_stft(x.ctypes.data, B, L, n_fft, win_length, hop, window.encode(), int(center),
                               pad_mode.encode(), out.ctypes.data, C.byref(t))
    if rc != 0 or t.value != T:
        raise OracleError(lib().ade_oracle_last_error().decode())
    return out


def oracle_istft(spec: np.ndarray, n_fft, win_length, hop, window, center=True):
    spec = np.ascontiguousarray(spec, np.float32)
    B, _, T = spec.shape
    raw = n_fft + hop * (T - 1)
    out_len = raw - n_fft if center else raw
    out = np.empty((B, out_len), np.float32)
    n = C.c_int()
    rc = lib().ade_oracle_istft(spec.ctypes.data, B, T, n_fft, win_length, hop, window.encode(), int(center),
                                out.ctypes.data, C.byref(n))
    if rc != 0 or n.value != out_len:
        raise OracleError(lib().ade_oracle_last_error().decode())
    return out
