"""``STFT_Process`` — host-side mirror of the reference's STFT module over libade's generic operator.

The reference builds one ``STFT_Process(model_type, n_fft, win_length, hop_len, max_frames, window_type, center_pad,
pad_mode, static_norm=...)`` per direction (GTCRN/STFT_Process.py:129-341; e.g. GTCRN/Export_GTCRN.py:719-741,
ZipEnhancer/Export_ZipEnhancer.py:947-948, DFSMN/Export_DFSMN.py:273-274) and calls it with ``(B, 1, L)`` audio or
``(B, 2F, T)`` packed spectra.  This class keeps those constructor arguments and call shapes; the compute is
``csrc/ade_stft.hip`` (an LDS FFT when n_fft is 5-smooth, as in every starred folder; the dense windowed DFT as fp32 MFMA
GEMMs otherwise) — device tensors in, device tensors out, no CPU path.

Window names: every model folder carries its own registry; the periodic / symmetric choice is explicit here
(``"hann"``, ``"hann_sqrt"``, ``"hamming"`` = torch ``periodic=True``; suffix ``"_sym"`` = ``periodic=False``;
``"hamming_periodic"`` = ``"hamming"``).  ``WINDOW_ALIASES`` maps a folder's names to these.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

from . import _lib

# (folder registry name -> canonical name) where the folder's binding differs from the GTCRN registry
WINDOW_ALIASES = {
    "DFSMN": {"hamming": "hamming_sym", "hann_sqrt": "hann_sqrt_sym"},          # DFSMN/STFT_Process.py:92-95
    "ZipEnhancer": {"hann_sqrt": "hann_sqrt_sym"},                                # ZipEnhancer/STFT_Process.py:94
    "Mel_Band_Roformer": {"hann_sqrt": "hann_sqrt_sym"},
}


class STFT_Process:
    def __init__(self, model_type: str, n_fft: int = 512, win_length: int = 512, hop_len: int = 256, max_frames: int = 0,
                 window_type: str = "hann_sqrt", center_pad: bool = True, pad_mode: str = "reflect", static_norm: bool = True,
                 device_id: int = 0, library: Optional[_lib.AdeLibrary] = None):
        if model_type not in ("stft_A", "stft_B", "istft_A", "istft_B"):
            raise ValueError(f"Unknown model_type: {model_type}")
        # static_norm only chooses between a precomputed and a per-call sum of squared windows in the reference (:253-273 vs :355-361);
        # the operator always divides by the exact per-sample sum, which both forms equal.  What static_norm = False changes is the LENGTH when the
        # ISTFT is fed fewer frames than max_frames (a DYNAMIC_AXES export, e.g. UL-UNAS/Export_UL_UNAS.py:43, 957): the slice
        # [out_start : out_end(max_frames)] then keeps the second half of the last frame (UL-UNAS/STFT_Process.py:170-177, 317-326).
        self._dynamic = model_type in ("istft_A", "istft_B") and not static_norm and max_frames > 0
        self._out_start = n_fft // 2 if center_pad else 0
        self._out_end = n_fft + hop_len * (max_frames - 1) - self._out_start
        self._lib = library or _lib.get_library()
        self.model_type, self.n_fft, self.hop_len, self.n_frames = model_type, n_fft, hop_len, max_frames
        self.half_n_fft = n_fft // 2
        self._h = C.c_void_p()
        cfg = _lib.StftConfig(n_fft, win_length, hop_len, window_type.encode(), None, int(bool(center_pad)), pad_mode.encode())
        st = self._lib.c.ade_stft_create(C.byref(cfg), int(device_id), C.byref(self._h))
        if st != _lib.ADE_OK:
            msg = self._lib.c.ade_stft_last_error(None)
            _lib.raise_for_status(st, msg.decode() if msg else f"ade_stft_create status {st}")
        if self._dynamic:
            self._check(self._lib.c.ade_stft_keep_tail(self._h, 1))

    def _kept(self, frames: int) -> int:
        """Samples the reference's slice [out_start : out_end(max_frames)] keeps of a T-frame overlap-add."""
        return max(0, min(self.n_fft + self.hop_len * (frames - 1), self._out_end) - self._out_start)

    def _check(self, st: int) -> None:
        if st != _lib.ADE_OK:
            msg = self._lib.c.ade_stft_last_error(self._h)
            _lib.raise_for_status(st, msg.decode() if msg else f"libade status {st}")

    def frames(self, length: int) -> int:
        t = C.c_int()
        self._check(self._lib.c.ade_stft_frames(self._h, int(length), C.byref(t)))
        return t.value

    def output_length(self, frames: int) -> int:
        n = C.c_int()
        self._check(self._lib.c.ade_stft_output_length(self._h, int(frames), C.byref(n)))
        return n.value

    def __call__(self, x, *more, stream: Optional[int] = None):
        return self.forward_polar(x, more[0], stream) if self.model_type == "istft_A" else self.forward(x, stream)

    def forward(self, x, stream: Optional[int] = None):
        """stft_B: (B, 1, L) or (B, L) float32 CUDA tensor -> packed (B, 2F, T) (``split()`` gives the reference's ``_stft_B_forward`` pair
        :298-301); stft_A: the real rows only (B, F, T) (:285-296).  istft_B: (B, 2F, T) -> (B, 1, L_out); istft_A: ``forward(magnitude,
        phase)``, each (B, F, T) (:343-361)."""
        import torch
        if self.model_type == "istft_A":
            raise TypeError("istft_A takes (magnitude, phase): call forward_polar")
        if not x.is_cuda or x.dtype != torch.float32:
            raise ValueError("STFT_Process takes float32 device tensors")
        x = x.contiguous()
        sp = C.c_void_p(stream) if stream else None
        if self.model_type in ("stft_A", "stft_B"):
            B, L = int(x.shape[0]), int(x.shape[-1])
            T = self.frames(L)
            out = torch.empty((B, self.n_fft + 2, T), dtype=torch.float32, device=x.device)
            self._check(self._lib.c.ade_stft_analyze(self._h, C.c_void_p(x.data_ptr()), B, L, C.c_void_p(out.data_ptr()), sp))
            return out if self.model_type == "stft_B" else out[:, :self.half_n_fft + 1]
        B, T = int(x.shape[0]), int(x.shape[2])
        if int(x.shape[1]) != self.n_fft + 2:
            raise ValueError(f"packed spectrum must have {self.n_fft + 2} rows, got {tuple(x.shape)}")
        if self.n_frames and T != self.n_frames and not self._dynamic:
            raise ValueError(f"static ISTFT was built for {self.n_frames} frames, got {T}")
        out = torch.empty((B, 1, self.output_length(T)), dtype=torch.float32, device=x.device)
        self._check(self._lib.c.ade_stft_synthesize(self._h, C.c_void_p(x.data_ptr()), B, T, C.c_void_p(out.data_ptr()), sp))
        return out[..., :self._kept(T)] if self._dynamic else out

    @staticmethod
    def split(packed):
        """(B, 2F, T) -> (real, imag), each (B, F, T): the two outputs of the reference's un-packed ``_stft_B_forward`` (:298-301)."""
        F = packed.shape[1] // 2
        return packed[:, :F], packed[:, F:]

    def forward_polar(self, magnitude, phase, stream: Optional[int] = None):
        """istft_A (:343-361): magnitude, phase (B, F, T) float32 device tensors -> (B, 1, L_out); the polar -> rectangular step runs inside
        the synthesis GEMM's operand loader."""
        import torch
        if self.model_type != "istft_A":
            raise TypeError("forward_polar belongs to model_type 'istft_A'")
        if magnitude.shape != phase.shape or magnitude.dim() != 3 or int(magnitude.shape[1]) != self.half_n_fft + 1:
            raise ValueError(f"magnitude / phase must both be (B, {self.half_n_fft + 1}, T)")
        if not (magnitude.is_cuda and phase.is_cuda) or magnitude.dtype != torch.float32 or phase.dtype != torch.float32:
            raise ValueError("STFT_Process takes float32 device tensors")
        magnitude, phase = magnitude.contiguous(), phase.contiguous()
        B, T = int(magnitude.shape[0]), int(magnitude.shape[2])
        if self.n_frames and T != self.n_frames and not self._dynamic:
            raise ValueError(f"static ISTFT was built for {self.n_frames} frames, got {T}")
        out = torch.empty((B, 1, self.output_length(T)), dtype=torch.float32, device=magnitude.device)
        self._check(self._lib.c.ade_stft_synthesize_polar(self._h, C.c_void_p(magnitude.data_ptr()), C.c_void_p(phase.data_ptr()), B, T,
                                                          C.c_void_p(out.data_ptr()), C.c_void_p(stream) if stream else None))
        return out[..., :self._kept(T)] if self._dynamic else out

    def close(self) -> None:
        if getattr(self, "_h", None) and self._h:
            self._lib.c.ade_stft_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
