"""Kaldi triangular mel filterbank (the table ``torchaudio.compliance.kaldi.get_mel_banks`` returns).

DFSMN's feature extractor multiplies the 2048-point power spectrum by this bank (DFSMN/Export_DFSMN.py:134-137,
``get_mel_banks(120, 2048, 48000.0, 20.0, 0.0, 100.0, -500.0, 1.0)``).  torchaudio is not a dependency of this package
(and is absent from the build image), so the published Kaldi algorithm (kaldi/src/feat/mel-computations.cc,
``MelBanks::MelBanks``) is restated here for vtln_warp = 1: ``num_bins`` triangles, equally spaced on the mel scale
``1127 ln(1 + f/700)`` between ``low_freq`` and ``high_freq`` (<= 0: offset from Nyquist), evaluated at the centre
frequencies of the first ``padded_window_size / 2`` FFT bins.  The bank travels inside the weight blob (like GTCRN's ERB
matrices), so the engine, the oracle and the reference module used for the fixtures all consume the same numbers.
"""
from __future__ import annotations

import numpy as np


def mel_scale(freq):
    return 1127.0 * np.log(1.0 + np.asarray(freq, np.float64) / 700.0)


def get_mel_banks(num_bins: int, window_length_padded: int, sample_freq: float, low_freq: float, high_freq: float,
                  vtln_low: float = 100.0, vtln_high: float = -500.0, vtln_warp_factor: float = 1.0) -> np.ndarray:
    """(num_bins, window_length_padded // 2) float32."""
    if vtln_warp_factor != 1.0:
        raise NotImplementedError("VTLN warping is not used by the reference")
    num_fft_bins = window_length_padded // 2
    nyquist = 0.5 * sample_freq
    if high_freq <= 0.0:
        high_freq += nyquist
    fft_bin_width = sample_freq / window_length_padded
    mel_low, mel_high = float(mel_scale(low_freq)), float(mel_scale(high_freq))
    delta = (mel_high - mel_low) / (num_bins + 1)
    b = np.arange(num_bins, dtype=np.float64)[:, None]
    left, center, right = mel_low + b * delta, mel_low + (b + 1.0) * delta, mel_low + (b + 2.0) * delta
    mel = mel_scale(fft_bin_width * np.arange(num_fft_bins, dtype=np.float64))[None, :]
    up = (mel - left) / (center - left)
    down = (right - mel) / (right - center)
    return np.maximum(0.0, np.minimum(up, down)).astype(np.float32)
