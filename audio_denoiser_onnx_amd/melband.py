"""Mel-Band-Roformer host side: the tensor set libade expects for ``model_family = "mel_band_roformer"`` and its manifest.

The engine (csrc/ade_melband.hip) consumes the FUSED buffers the reference's export constructor registers
(Mel_Band_Roformer/Stereo/Export_MelBandRoformer.py:455-531: ``bs_w_i / bs_b_i``, ``time{i}_* / freq{i}_*``,
``me_w1t / me_b1 / me_w2t / me_b2``, ``me_w3_i / me_b3_i``) under their registered names, plus the two band tables
(``freq_indices``, ``dim_inputs``; :359-378), which this package rebuilds from the STFT geometry (mel_bands.py).
A converter from the upstream checkpoint would produce exactly this dict; it is not part of this package because no
checkpoint is available offline to pin it against.
"""
from __future__ import annotations

from typing import Dict, Mapping

import numpy as np

from .mel_bands import band_tables
from .metadata import build_audio_metadata

SAMPLE_RATE, NFFT, HOP, NUM_BANDS = 44100, 2048, 441, 60      # Export_MelBandRoformer.py:35-45, config num_bands


def model_tensors(fused: Mapping[str, np.ndarray], num_bands: int = NUM_BANDS) -> Dict[str, np.ndarray]:
    """``fused`` (registered buffer name -> array) + the band tables, all float32 (indices are exact in fp32)."""
    fi, di = band_tables(SAMPLE_RATE, NFFT, num_bands, 2)
    out = {k: np.ascontiguousarray(v, np.float32) for k, v in fused.items()}
    out["freq_indices"] = fi.astype(np.float32)
    out["dim_inputs"] = di.astype(np.float32)
    return out


def metadata(input_audio_length: int, dft_tables: str = "reference") -> Dict[str, str]:
    """Manifest of a static, un-folded stereo export (``USE_BATCH_FOLD = False``): fold windows are passed as batch rows.
    ``dft_tables``: "reference" = the reference's fp32-angle DFT matrices (bit-compatible behaviour, default);
    "exact" = exactly reduced angles (see csrc/ade_melband.hip)."""
    if input_audio_length % HOP:
        raise ValueError(f"input_audio_length must be a multiple of the hop ({HOP})")
    return build_audio_metadata(producer="audio_denoiser_onnx_amd", model_name="MelBandRoformer", task="denoise",
                                model_family="mel_band_roformer", input_audio_length=input_audio_length, in_sample_rate=SAMPLE_RATE,
                                nfft=NFFT, window_length=NFFT, hop_length=HOP, window_type="hann", center_pad=True, pad_mode="reflect",
                                use_batch_fold=False, input_channels=2, output_channels=2, extra={"ade_dft_tables": dft_tables})
