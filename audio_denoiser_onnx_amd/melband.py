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


def metadata(input_audio_length: int, dft_tables: str = "reference", use_batch_fold: bool = False,
             batch_window_seconds: float = 1.5) -> Dict[str, str]:
    """Manifest of a static stereo export.  ``use_batch_fold`` as in the reference (Export_MelBandRoformer.py:47-51): the graph
    input is ``input_audio_length`` rounded up to whole windows of ``batch_window_seconds`` (itself rounded up to the hop),
    each window an independent stereo clip; without it the clip is ``input_audio_length`` long (a multiple of the hop) and
    windows can still be passed as batch rows.
    ``dft_tables``: "reference" = the reference's fp32-angle DFT matrices (bit-compatible behaviour, default);
    "exact" = exactly reduced angles (see csrc/ade_melband.hip)."""
    if not use_batch_fold and input_audio_length % HOP:
        raise ValueError(f"input_audio_length must be a multiple of the hop ({HOP})")
    return build_audio_metadata(producer="audio_denoiser_onnx_amd", model_name="MelBandRoformer", task="denoise",
                                model_family="mel_band_roformer", input_audio_length=input_audio_length, in_sample_rate=SAMPLE_RATE,
                                nfft=NFFT, window_length=NFFT, hop_length=HOP, window_type="hann", center_pad=True, pad_mode="reflect",
                                use_batch_fold=use_batch_fold, batch_window_seconds=batch_window_seconds, input_channels=2,
                                output_channels=2, extra={"ade_dft_tables": dft_tables})


def synthetic_spec(depth: int, dim: int = 384, heads: int = 8, dim_head: int = 64, ff_mult: int = 4, me_hidden: int = 1536,
                   num_bands: int = NUM_BANDS):
    """(name, shape, scale) of every fused buffer for random-init weights of the architecture (``weightgen.materialise``):
    what tools/bench_melband.py runs on, since no checkpoint is available offline.  Scales keep activations O(1)."""
    _, dims = band_tables(SAMPLE_RATE, NFFT, num_bands, 2)
    di = heads * dim_head
    spec = []
    for i, d in enumerate(int(x) for x in dims):
        spec += [(f"bs_w_{i}", [dim, d], 1.5), (f"bs_b_{i}", [dim], 0.05)]
    for i in range(depth):
        for axis in ("time", "freq"):
            p = f"{axis}{i}_"
            spec += [(p + "in_w", [3 * di + heads, dim], 0.75), (p + "in_b", [3 * di + heads], 0.05), (p + "out_w", [dim, di], 0.03),
                     (p + "ff1_w", [ff_mult * dim, dim], 2.0), (p + "ff1_b", [ff_mult * dim], 0.05), (p + "ff2_w", [dim, ff_mult * dim], 0.02),
                     (p + "ff2_b", [dim], 0.05), (p + "out_g", [dim], float(dim) ** 0.5)]
    for i, d in enumerate(int(x) for x in dims):
        spec += [(f"me_w3_{i}", [2 * d, me_hidden], 0.05), (f"me_b3_{i}", [2 * d], 0.05)]
    spec += [("me_w1t", [num_bands, dim, me_hidden], 0.1), ("me_b1", [num_bands, 1, me_hidden], 0.05),
             ("me_w2t", [num_bands, me_hidden, me_hidden], 0.05), ("me_b2", [num_bands, 1, me_hidden], 0.05)]
    return spec


def flops_per_clip(frames: int, depth: int, dim: int = 384, heads: int = 8, dim_head: int = 64, ff_mult: int = 4, me_hidden: int = 1536,
                   num_bands: int = NUM_BANDS) -> float:
    """Multiply-add flops (2 per MAC) of one stereo clip of ``frames`` STFT frames: the dense-DFT STFT / ISTFT, band split,
    transformers (projections, FFN, attention scores + values), mask estimator."""
    _, dims = band_tables(SAMPLE_RATE, NFFT, num_bands, 2)
    di, s2 = heads * dim_head, int(dims.sum())
    rows = num_bands * frames
    stft = 2 * 2 * (2 * 1025) * NFFT * (2 * frames)
    split = 2 * frames * s2 * dim
    proj = 2 * rows * dim * ((3 * di + heads) + di + 2 * ff_mult * dim)
    attn = 2 * 2 * heads * dim_head * (num_bands * frames * frames + frames * num_bands * num_bands)
    me = 2 * rows * (dim * me_hidden + me_hidden * me_hidden) + 2 * frames * 2 * s2 * me_hidden
    return float(stft + split + depth * (2 * proj + attn) + me)
