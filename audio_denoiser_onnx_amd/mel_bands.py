"""Band tables of Mel-Band-Roformer: which STFT bins each of the ``num_bands`` overlapping mel bands reads and writes.

Restates the table construction of Mel_Band_Roformer/Stereo/Export_MelBandRoformer.py:359-378 (and the Slaney mel filter
bank it calls, :67-143): a band owns every rfft bin where its triangular mel filter is positive, the first band is forced to
own bin 0 and the last band the Nyquist bin; for stereo each owned bin expands to the channel-interleaved pair
(2 f, 2 f + 1).  Only the SUPPORT of the filters matters -- their values are never used -- so the Slaney area
normalisation is left out.  Pinned against the tables the reference itself builds (tests/golden/melband_seed0_io.npz).
"""
from __future__ import annotations

import numpy as np


def _hz_to_mel(f):
    f = np.asarray(f, np.float64)
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-300) / min_log_hz) / logstep, f / f_sp)


def _mel_to_hz(m):
    m = np.asarray(m, np.float64)
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def band_support(sample_rate: int = 44100, n_fft: int = 2048, num_bands: int = 60) -> np.ndarray:
    """bool (num_bands, n_fft/2 + 1): bin f belongs to band i."""
    bins = n_fft // 2 + 1
    fft_f = np.linspace(0.0, sample_rate / 2.0, bins)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(0.0), _hz_to_mel(sample_rate / 2.0), num_bands + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fft_f[None, :]
    tri = np.empty((num_bands, bins), np.float32)              # the reference stores the filters in fp32 before testing > 0
    for i in range(num_bands):
        tri[i] = np.maximum(0.0, np.minimum(-ramps[i] / fdiff[i], ramps[i + 2] / fdiff[i + 1]))
    own = tri > 0
    own[0, 0] = True
    own[-1, -1] = True
    return own


def band_tables(sample_rate: int = 44100, n_fft: int = 2048, num_bands: int = 60, channels: int = 2):
    """(freq_indices int32 (S,), dim_inputs int32 (num_bands,)): band-major list of channel-interleaved bin indices
    (f * channels + ch) and the per-band feature widths 2 * bins_in_band * channels (re/im pairs)."""
    own = band_support(sample_rate, n_fft, num_bands)
    f_idx = np.broadcast_to(np.arange(own.shape[1]), own.shape)[own]                   # band-major, ascending bin
    idx = (f_idx[:, None] * channels + np.arange(channels)[None, :]).reshape(-1)
    return idx.astype(np.int32), (2 * own.sum(axis=1) * channels).astype(np.int32)
