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
             batch_window_seconds: float = 1.5, gemm_dtype: str = "f32", dynamic_axes: bool = False,
             in_sample_rate: int = SAMPLE_RATE, out_sample_rate: int = SAMPLE_RATE) -> Dict[str, str]:
    """Manifest of a static stereo export.  ``use_batch_fold`` as in the reference (Export_MelBandRoformer.py:47-51): the graph
    input is ``input_audio_length`` rounded up to whole windows of ``batch_window_seconds`` (itself rounded up to the hop),
    each window an independent stereo clip; without it the clip is ``input_audio_length`` long (a multiple of the hop) and
    windows can still be passed as batch rows.
    ``dft_tables``: "reference" = the reference's fp32-angle DFT matrices (bit-compatible behaviour, default);
    "exact" = exactly reduced angles (see csrc/ade_melband.hip).
    ``gemm_dtype``: "f32" (default, the parity path) | "bf16" = the transformer stack and the mask estimator on bf16 activations and weights stored in HBM
    (csrc/ade_gemm16.h; fp32 residual stream, norms, softmax statistics, STFT, band split, mask, ISTFT): BASELINE.json's dtype for this model, ~4.4 x the f32 path,
    ~41 dB from it.
    ``dynamic_axes`` = a DYNAMIC_AXES export (Export_MelBandRoformer.py:33, :50): any ``input_audio_length`` whose model-rate length reaches one window,
    other input / output sample rates (:52-53, :630-644, :660-680), and an output that keeps the tail of the last frame (Stereo/STFT_Process.py:296-306).
    The engine serves one input length per handle."""
    if dynamic_axes and use_batch_fold:
        raise ValueError("Batch folding requires a static shape (dynamic_axes=False)")
    if not dynamic_axes and (in_sample_rate != SAMPLE_RATE or out_sample_rate != SAMPLE_RATE):
        raise ValueError("other sample rates need dynamic_axes=True: the static export sizes its frames from the input-rate length (:50)")
    if not use_batch_fold and not dynamic_axes and input_audio_length % HOP:
        raise ValueError(f"input_audio_length must be a multiple of the hop ({HOP})")
    return build_audio_metadata(producer="audio_denoiser_onnx_amd", model_name="MelBandRoformer", task="denoise",
                                model_family="mel_band_roformer", input_audio_length=input_audio_length, in_sample_rate=in_sample_rate,
                                out_sample_rate=out_sample_rate, model_sample_rate=SAMPLE_RATE, dynamic_axes=dynamic_axes,
                                nfft=NFFT, window_length=NFFT, hop_length=HOP, window_type="hann", center_pad=True, pad_mode="reflect",
                                use_batch_fold=use_batch_fold, batch_window_seconds=batch_window_seconds, input_channels=2,
                                output_channels=2, extra={"ade_dft_tables": dft_tables, "ade_gemm_dtype": gemm_dtype})


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


def fuse_checkpoint(state: Mapping[str, np.ndarray], heads: int = 8, dim_head: int = 64, num_bands: int = NUM_BANDS) -> Dict[str, np.ndarray]:
    """Checkpoint ``state_dict`` (upstream Mel-Band-Roformer key names, e.g. ``layers.0.0.layers.0.0.to_qkv.weight``) -> the fused
    buffers the engine consumes.  Restates the fold algebra of the reference's export constructor
    (Export_MelBandRoformer.py:455-538) in float64 with one rounding to fp32 at the end, as the reference does:
      * every RMSNorm gain (scale * gamma, scale = sqrt(dim)) folds into the Linear that consumes it (band split :456-461,
        attention input :505-511, feed-forward :514-516, transformer output norm :519);
      * q | k | v | gates stack into one projection, the attention scale dim_head^-1/2 folds into the q rows (:507-511);
      * the mask-estimator MLP's two uniform Linears stack across bands, transposed for bmm (:498-501); the scatter-average
        1 / (bands owning a bin) folds into the GLU value rows of the last Linear (:471-492).
    Pinned against the reference's own constructor by tests/test_melband.py::test_checkpoint_fusion_matches_reference."""
    g = {k: np.asarray(v, np.float64) for k, v in state.items()}
    di = heads * dim_head
    fi, dims = band_tables(SAMPLE_RATE, NFFT, num_bands, 2)
    out: Dict[str, np.ndarray] = {}

    def rms_gain(gamma):
        return gamma * float(gamma.shape[0]) ** 0.5

    for i in range(num_bands):
        p = f"band_split.to_features.{i}."
        out[f"bs_w_{i}"] = g[p + "1.weight"] * rms_gain(g[p + "0.gamma"])[None, :]
        out[f"bs_b_{i}"] = g[p + "1.bias"]
    depth = 0
    while f"layers.{depth}.0.layers.0.0.to_qkv.weight" in g:
        depth += 1
    for i in range(depth):
        for axis, name in ((0, "time"), (1, "freq")):
            p = f"layers.{i}.{axis}."
            a, f = p + "layers.0.0.", p + "layers.0.1."
            wqkv = g[a + "to_qkv.weight"]
            stacked = np.concatenate((wqkv[:di] * dim_head ** -0.5, wqkv[di:2 * di], wqkv[2 * di:], g[a + "to_gates.weight"]), axis=0)
            out[f"{name}{i}_in_w"] = stacked * rms_gain(g[a + "norm.gamma"])[None, :]
            out[f"{name}{i}_in_b"] = np.concatenate((np.zeros(3 * di), g[a + "to_gates.bias"]))
            out[f"{name}{i}_out_w"] = g[a + "to_out.0.weight"]
            out[f"{name}{i}_ff1_w"] = g[f + "net.1.weight"] * rms_gain(g[f + "net.0.gamma"])[None, :]
            out[f"{name}{i}_ff1_b"] = g[f + "net.1.bias"]
            out[f"{name}{i}_ff2_w"] = g[f + "net.4.weight"]
            out[f"{name}{i}_ff2_b"] = g[f + "net.4.bias"]
            out[f"{name}{i}_out_g"] = rms_gain(g[p + "norm.gamma"])
    # bands owning each channel-interleaved bin -> per gathered entry, repeated for (re, im)
    owners = np.bincount(fi, minlength=2 * (NFFT // 2 + 1)).astype(np.float64)
    denom = np.repeat(1.0 / np.maximum(owners, 1e-8)[fi], 2)
    w1, b1, w2, b2, off = [], [], [], [], 0
    for i in range(num_bands):
        p = f"mask_estimators.0.to_freqs.{i}.0."
        d = int(dims[i])
        w1.append(g[p + "0.weight"]); b1.append(g[p + "0.bias"]); w2.append(g[p + "2.weight"]); b2.append(g[p + "2.bias"])
        w3, b3 = g[p + "4.weight"].copy(), g[p + "4.bias"].copy()
        w3[:d] *= denom[off:off + d, None]
        b3[:d] *= denom[off:off + d]
        off += d
        out[f"me_w3_{i}"], out[f"me_b3_{i}"] = w3, b3
    out["me_w1t"] = np.stack(w1).transpose(0, 2, 1)
    out["me_b1"] = np.stack(b1)[:, None, :]
    out["me_w2t"] = np.stack(w2).transpose(0, 2, 1)
    out["me_b2"] = np.stack(b2)[:, None, :]
    return {k: np.ascontiguousarray(v, np.float32) for k, v in out.items()}
