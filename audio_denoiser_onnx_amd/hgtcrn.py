"""H-GTCRN host side: checkpoint ``state_dict`` -> the tensor set libade expects for ``model_family = "h_gtcrn"``, and its manifest.

Restates ``fuse_bn_`` of the reference's blocks (H-GTCRN/Export_H_GTCRN.py:193-217, 254-258): every BatchNorm folded into the convolution
before it (ConvTranspose2d weights are (Cin, Cout / groups, 1, 5)).  The output uses GTCRN's folded names (``point_conv1.weight``,
``point_act.weight`` ...), so the engine shares GTCRN's weight packer; unlike GTCRN, the decoder's GTConvBlocks are ordinary Conv2d
(:405-407).  Pinned through the oracle against the reference's own forward (tests/test_hgtcrn.py).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Mapping

import numpy as np

from .metadata import build_audio_metadata

BN_EPS = 1e-5
NFFT, HOP, ERB_LOW, ERB_BANDS = 512, 256, 65, 64
WPE_RT60, WPE_DELAY, WPE_ITER, IVA_ITER, CG_SOLVE_ITER = 0.3, 2, 1, 10, 6          # (:50-54)

_CONVBLOCKS = [("encoder.en_convs.0.", False, 1), ("encoder.en_convs.1.", False, 2), ("decoder.de_convs.3.", True, 2), ("decoder.de_convs.4.", True, 1)]
_GTBLOCKS = [f"encoder.en_convs.{i}." for i in (2, 3, 4)] + [f"decoder.de_convs.{i}." for i in (0, 1, 2)]


def erb_filters() -> np.ndarray:
    """(64, 192) triangular ERB filters over bins 65..256 (``ERB.erb_filter_banks`` :98-123)."""
    fs, nfft = 16000, NFFT
    hz2erb = lambda f: 24.7 * np.log10(0.00437 * f + 1)                # noqa: E731
    erb2hz = lambda e: (10 ** (e / 24.7) - 1) / 0.00437                # noqa: E731
    pts = np.linspace(hz2erb(ERB_LOW / nfft * fs), hz2erb(8000), ERB_BANDS)
    bins = np.round(erb2hz(pts) / fs * nfft).astype(np.int32)
    f = np.zeros((ERB_BANDS, nfft // 2 + 1), np.float32)
    f[0, bins[0]:bins[1]] = (bins[1] - np.arange(bins[0], bins[1]) + 1e-12) / (bins[1] - bins[0] + 1e-12)
    for i in range(ERB_BANDS - 2):
        f[i + 1, bins[i]:bins[i + 1]] = (np.arange(bins[i], bins[i + 1]) - bins[i] + 1e-12) / (bins[i + 1] - bins[i] + 1e-12)
        f[i + 1, bins[i + 1]:bins[i + 2]] = (bins[i + 2] - np.arange(bins[i + 1], bins[i + 2]) + 1e-12) / (bins[i + 2] - bins[i + 1] + 1e-12)
    f[-1, bins[-2]:bins[-1] + 1] = 1 - f[-2, bins[-2]:bins[-1] + 1]
    return np.abs(f[:, ERB_LOW:]).astype(np.float32)


def _fold(w, b, gamma, beta, mean, var, transposed, groups):
    scale = (gamma / np.sqrt(var + BN_EPS)).astype(np.float32)
    if transposed:
        cin, og = w.shape[0], w.shape[1]
        w2 = (w.reshape(groups, cin // groups, og, w.shape[2], w.shape[3]) * scale.reshape(groups, 1, og, 1, 1)).reshape(w.shape)
    else:
        w2 = w * scale.reshape(-1, 1, 1, 1)
    b2 = beta - mean * scale if b is None else (b - mean) * scale + beta
    return w2.astype(np.float32), b2.astype(np.float32)


def fold_state_dict(sd: Mapping[str, np.ndarray]) -> "OrderedDict[str, np.ndarray]":
    """Checkpoint-format ``GTCRN_IVA`` state_dict (numpy arrays) -> the BN-folded tensor set libade loads."""
    sd = {k: np.asarray(v, dtype=np.float32) for k, v in sd.items() if "num_batches_tracked" not in k}
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()

    def bn(prefix):
        return sd[prefix + "weight"], sd[prefix + "bias"], sd[prefix + "running_mean"], sd[prefix + "running_var"]

    for p, transposed, groups in _CONVBLOCKS:
        out[p + "conv.weight"], out[p + "conv.bias"] = _fold(sd[p + "conv.weight"], sd.get(p + "conv.bias"), *bn(p + "bn."), transposed, groups)
        if p + "act.weight" in sd:
            out[p + "act.weight"] = sd[p + "act.weight"]
    for p in _GTBLOCKS:
        for conv, groups in (("point_conv1", 1), ("depth_conv", 16), ("point_conv2", 1)):
            q = p + conv + "."
            out[p + conv + ".weight"], out[p + conv + ".bias"] = _fold(sd[q + "conv.weight"], sd.get(q + "conv.bias"), *bn(q + "bn."), False, groups)
        out[p + "point_act.weight"] = sd[p + "point_conv1.act.weight"]
        out[p + "depth_act.weight"] = sd[p + "depth_conv.act.weight"]
        for leaf in ("tra.att_gru.weight_ih_l0", "tra.att_gru.weight_hh_l0", "tra.att_gru.bias_ih_l0", "tra.att_gru.bias_hh_l0", "tra.att_fc.weight", "tra.att_fc.bias"):
            out[p + leaf] = sd[p + leaf]
    for k, v in sd.items():
        if k.startswith("dpgrnn"):
            out[k] = v
    erb = sd["erb.erb_fc.weight"] if "erb.erb_fc.weight" in sd else erb_filters()          # (64, 192); a fixed table (:87-88)
    out["erb.erb_weight_t"] = np.ascontiguousarray(erb.T)                                   # (192, 64)
    out["erb.ierb_weight_t"] = np.ascontiguousarray(sd["erb.ierb_fc.weight"].T if "erb.ierb_fc.weight" in sd else erb)   # (64, 192)
    return out


def metadata(input_audio_length: int = 32000, use_batch_fold: bool = False, batch_window_seconds: float = 1.5, in_sample_rate: int = 16000,
             out_sample_rate: int = 16000, dynamic_axes: bool = False) -> Dict[str, str]:
    """Manifest keys the reference stamps for this model (:1180-1186): 16 kHz, two microphones in, one channel out, 512 / 256 'hann' STFT;
    the MODEL-rate length int(L * 16000 / in_rate) must be whole hops (:33, :45); optionally batch-fold (WPE / AuxIVA then run per window,
    :43-47; equal rates only); other input / output rates go through the export's linear-interpolation edges (:953-970, :1036-1052).
    ``dynamic_axes``: the DYNAMIC_AXES export (:27, :1075, :1097, :1110): the same arithmetic at the call's frame count with the ISTFT's dynamic trim -- the output is
    half a window (256 model-rate samples) longer than the input; not with batch folding (:41)."""
    if dynamic_axes and use_batch_fold:
        raise ValueError("Batch folding requires a static shape (DYNAMIC_AXES = False)")
    return build_audio_metadata(producer="audio_denoiser_onnx_amd", model_name="H_GTCRN", task="denoise", model_family="h_gtcrn",
                                input_audio_length=input_audio_length, in_sample_rate=in_sample_rate, out_sample_rate=out_sample_rate, model_sample_rate=16000, nfft=NFFT, window_length=NFFT, hop_length=HOP,
                                window_type="hann", center_pad=True, pad_mode="reflect", use_batch_fold=use_batch_fold, dynamic_axes=dynamic_axes,
                                batch_window_seconds=batch_window_seconds, input_channels=2, output_channels=1, feature_kind="stft_wpe_auxiva",
                                extra={"wpe_rt60": WPE_RT60, "wpe_delay": WPE_DELAY, "wpe_iter": WPE_ITER,
                                       "iva_iter": IVA_ITER, "cg_solve_iter": CG_SOLVE_ITER})
