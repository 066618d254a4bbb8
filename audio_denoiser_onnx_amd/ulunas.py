"""UL-UNAS host side: checkpoint ``state_dict`` -> the tensor set libade expects for ``model_family = "ul_unas"``, and its manifest.

Restates the folds of the reference's ``prepare_for_export_`` (UL-UNAS/Export_UL_UNAS.py:697-717): every BatchNorm into the
convolution before it (``fuse_bn_`` :243-262, 319-339, 411-431; ConvTranspose2d weights are (Cin, Cout / groups, kt, kf)), every
AffinePReLU into a positive / negative slope pair + bias (``fuse_for_export_`` :122-126), and the ``0.5 / ln 10`` of the log-power
feature into the first convolution (:707-710).  The grouped GRUs stay as their two half-width GRUs (the reference's block-diagonal
fusion :470-513 and the matching reorder of ``intra_fc`` :555-559 are an ONNX-export device with identical arithmetic).
Input keys are the optimised model's (``encoder.en_convs.0.conv.weight`` ...; ``convert_state_dict`` :742-822 maps the upstream
checkpoint onto them).  Pinned through the oracle against the reference's own export path (tests/test_ulunas.py).
"""
from __future__ import annotations

from typing import Dict, Mapping

import numpy as np

from .metadata import build_audio_metadata

# ULUNAS() defaults (:655-668)
TYPES, STRIDES, GROUPS = [0, 2, 1, 2, 1], [2, 2, 1, 1, 1], [1, 2, 2, 2, 2]
CHANNELS, KERNELS, WIDTHS = [12, 24, 24, 32, 16], [(3, 3), (2, 3), (2, 3), (1, 5), (1, 5)], [65, 33, 33, 33, 33]
ERB_LOW, ERB_HIGH, NFFT, HOP = 65, 64, 512, 256
BN_EPS = 1e-5


def block_plan():
    """[(prefix, type, cin, cout, width, (kt, kf), stride, groups, deconv, is_last)] for the 5 encoder and 5 decoder blocks (:577-652)."""
    plan, cin = [], 1
    for i in range(5):
        plan.append((f"encoder.en_convs.{i}.", TYPES[i], cin, CHANNELS[i], WIDTHS[i], KERNELS[i], STRIDES[i], GROUPS[i], False, False))
        cin = CHANNELS[i]
    j = 0
    for i in range(4, 0, -1):
        plan.append((f"decoder.de_convs.{j}.", TYPES[i], cin, CHANNELS[i - 1], WIDTHS[i - 1], KERNELS[i], STRIDES[i], GROUPS[i], True, False))
        cin = CHANNELS[i - 1]
        j += 1
    plan.append((f"decoder.de_convs.{j}.", TYPES[0], cin, 1, ERB_LOW + ERB_HIGH, KERNELS[0], STRIDES[0], GROUPS[0], True, True))
    return plan


def erb_matrix() -> np.ndarray:
    """(64, 192) triangular ERB filters over bins 65..256 (``ERB.erb_filter_banks`` :73-95)."""
    fs, nfft, n2, low = 16000, NFFT, ERB_HIGH, ERB_LOW
    hz2erb = lambda f: 21.4 * np.log10(0.00437 * f + 1)            # noqa: E731
    erb2hz = lambda e: (10 ** (e * 0.046728972) - 1) * 228.832951945   # noqa: E731
    pts = np.linspace(hz2erb(low / nfft * fs), hz2erb(8000), n2)
    bins = np.round(erb2hz(pts) / fs * nfft).astype(np.int32)
    f = np.zeros((n2, nfft // 2 + 1), np.float32)
    f[0, bins[0]:bins[1]] = (bins[1] - np.arange(bins[0], bins[1]) + 1e-12) / (bins[1] - bins[0] + 1e-12)
    for i in range(n2 - 2):
        f[i + 1, bins[i]:bins[i + 1]] = (np.arange(bins[i], bins[i + 1]) - bins[i] + 1e-12) / (bins[i + 1] - bins[i] + 1e-12)
        f[i + 1, bins[i + 1]:bins[i + 2]] = (bins[i + 2] - np.arange(bins[i + 1], bins[i + 2]) + 1e-12) / (bins[i + 2] - bins[i + 1] + 1e-12)
    f[-1, bins[-2]:bins[-1] + 1] = 1 - f[-2, bins[-2]:bins[-1] + 1]
    return np.abs(f[:, low:]).astype(np.float32)


def convert_state_dict(original: Mapping[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """Upstream UL-UNAS checkpoint keys (blocks as ``nn.Sequential``: ``ops.N`` / ``pconv.N`` / ``dconv.N`` / ``pconv1.N`` / ``pconv2.N``) ->
    the optimised model's flat names, with the AffinePReLU tables reshaped to (1, C, 1, W) and the slopes to (1, C, 1, 1): the mapping of
    the reference's ``convert_state_dict`` (:742-822)."""
    maps = {0: [("ops.1.", "conv."), ("ops.2.", "bn."), ("ops.3.", "act."), ("ops.4.", "ctfa.")],
            1: [("pconv.0.", "pconv_conv."), ("pconv.1.", "pconv_bn."), ("pconv.2.", "pconv_act."), ("dconv.1.", "dconv_conv."), ("dconv.2.", "dconv_bn."),
                ("dconv.3.", "dconv_act."), ("dconv.4.", "dconv_ctfa.")],
            2: [("pconv1.0.", "pconv1_conv."), ("pconv1.1.", "pconv1_bn."), ("pconv1.2.", "pconv1_act."), ("dconv.1.", "dconv_conv."), ("dconv.2.", "dconv_bn."),
                ("dconv.3.", "dconv_act."), ("pconv2.0.", "pconv2_conv."), ("pconv2.1.", "pconv2_bn."), ("pconv2.2.", "pconv2_ctfa.")]}
    dec_types = [TYPES[i] for i in range(4, 0, -1)] + [TYPES[0]]
    out: Dict[str, np.ndarray] = {}
    for key, value in original.items():
        new_key, value = key, np.asarray(value)
        for head, types in (("encoder.en_convs.", TYPES), ("decoder.de_convs.", dec_types)):
            if key.startswith(head):
                idx, rest = key[len(head):].split(".", 1)
                for old, new in maps[types[int(idx)]]:
                    if rest.startswith(old):
                        new_key = f"{head}{idx}.{new}{rest[len(old):]}"
                        break
        if new_key.endswith(("affine_weight", "affine_bias")) and value.ndim == 2:
            value = value.reshape(1, value.shape[0], 1, value.shape[1])
        elif new_key.endswith("slope_weight") and value.ndim <= 2:
            value = value.reshape(1, value.shape[0], 1, 1)
        out[new_key] = value
    return out


def _fold_conv(sd, conv, bn, transposed, groups):
    w = np.asarray(sd[conv + "weight"], np.float64)
    b = np.asarray(sd[conv + "bias"], np.float64) if conv + "bias" in sd else None
    scale = np.asarray(sd[bn + "weight"], np.float64) / np.sqrt(np.asarray(sd[bn + "running_var"], np.float64) + BN_EPS)
    mean, beta = np.asarray(sd[bn + "running_mean"], np.float64), np.asarray(sd[bn + "bias"], np.float64)
    if transposed:
        cin, og = w.shape[0], w.shape[1]
        w = (w.reshape(groups, cin // groups, og, w.shape[2], w.shape[3]) * scale.reshape(groups, 1, og, 1, 1)).reshape(w.shape)
    else:
        w = w * scale.reshape(-1, 1, 1, 1)
    b = beta - mean * scale if b is None else (b - mean) * scale + beta
    return w, b


def fold_state_dict(sd: Mapping[str, np.ndarray]) -> Dict[str, np.ndarray]:
    out: Dict[str, np.ndarray] = {}

    def act(dst, src):           # AffinePReLU -> positive / negative slope tables (C, W) + bias (C, W)
        aw = np.asarray(sd[src + "affine_weight"], np.float64)[0, :, 0, :]
        out[dst + "pos"] = aw + 1.0
        out[dst + "neg"] = aw + np.asarray(sd[src + "slope_weight"], np.float64)[0, :, 0, :]
        out[dst + "bias"] = np.asarray(sd[src + "affine_bias"], np.float64)[0, :, 0, :]

    def ctfa(dst, src):
        for k in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"):
            out[dst + "ta_" + k] = sd[src + "ta_gru." + k]
            out[dst + "fa_" + k] = sd[src + "fa.gru." + k]
            out[dst + "fa_" + k + "_reverse"] = sd[src + "fa.gru." + k + "_reverse"]
        out[dst + "ta_fc_w"], out[dst + "ta_fc_b"] = sd[src + "ta_fc.weight"], sd[src + "ta_fc.bias"]
        out[dst + "fa_fc_w"], out[dst + "fa_fc_b"] = sd[src + "fa.fc.weight"], sd[src + "fa.fc.bias"]

    for (p, typ, cin, cout, width, k, stride, groups, deconv, last) in block_plan():
        if typ == 0:
            w, b = _fold_conv(sd, p + "conv.", p + "bn.", deconv, groups)
            if p == "encoder.en_convs.0.":
                w = w * (0.5 / np.log(10.0))                       # log10(sqrt(power)) = 0.5 / ln 10 * log(power) (:707-710)
            out[p + "conv_w"], out[p + "conv_b"] = w, b
            if not last:
                act(p + "act_", p + "act.")
            ctfa(p + "ctfa_", p + "ctfa.")
        elif typ == 1:
            out[p + "pconv_w"], out[p + "pconv_b"] = _fold_conv(sd, p + "pconv_conv.", p + "pconv_bn.", False, groups)
            act(p + "pconv_act_", p + "pconv_act.")
            out[p + "dconv_w"], out[p + "dconv_b"] = _fold_conv(sd, p + "dconv_conv.", p + "dconv_bn.", deconv, cout)
            if not last:
                act(p + "dconv_act_", p + "dconv_act.")
            ctfa(p + "ctfa_", p + "dconv_ctfa.")
        else:
            out[p + "pconv1_w"], out[p + "pconv1_b"] = _fold_conv(sd, p + "pconv1_conv.", p + "pconv1_bn.", False, groups)
            act(p + "pconv1_act_", p + "pconv1_act.")
            out[p + "dconv_w"], out[p + "dconv_b"] = _fold_conv(sd, p + "dconv_conv.", p + "dconv_bn.", deconv, cout)
            act(p + "dconv_act_", p + "dconv_act.")
            out[p + "pconv2_w"], out[p + "pconv2_b"] = _fold_conv(sd, p + "pconv2_conv.", p + "pconv2_bn.", False, groups)
            ctfa(p + "ctfa_", p + "pconv2_ctfa.")
    for i in range(2):
        p = f"dpgrnn.{i}."
        for rnn in ("intra_rnn.rnn1", "intra_rnn.rnn2", "inter_rnn.rnn1", "inter_rnn.rnn2"):
            for k in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"):
                out[f"{p}{rnn}.{k}"] = sd[f"{p}{rnn}.{k}"]
                if rnn.startswith("intra"):
                    out[f"{p}{rnn}.{k}_reverse"] = sd[f"{p}{rnn}.{k}_reverse"]
        for k in ("intra_fc.weight", "intra_fc.bias", "intra_ln.weight", "intra_ln.bias", "inter_fc.weight", "inter_fc.bias", "inter_ln.weight",
                  "inter_ln.bias"):
            out[p + k] = sd[p + k]
    out["erb_filters"] = erb_matrix()
    return {k: np.ascontiguousarray(v, np.float32) for k, v in out.items()}


def metadata(input_audio_length: int = 16000, use_batch_fold: bool = False, batch_window_seconds: float = 1.5, dynamic_axes: bool = False,
             in_sample_rate: int = 16000, out_sample_rate: int = 16000) -> Dict[str, str]:
    """Manifest keys the reference stamps for this model (:1005-1012): 16 kHz, 512 / 256 'hann' STFT, no DC removal; optionally batch-fold
    (the graph input is the length rounded up to whole windows of ``batch_window_seconds``, each an independent clip, :41-44).
    ``dynamic_axes`` = a DYNAMIC_AXES export (:26, :41-43): other input / output sample rates become consistent (:835-845, :851-868, :890-905) and the output is the
    ISTFT's kept tail sliced to the caller-rate input length (:851, :888-889).  The engine serves one input length per handle."""
    if dynamic_axes and use_batch_fold:
        raise ValueError("Batch folding requires a static shape (dynamic_axes=False)")
    if not dynamic_axes and (in_sample_rate != 16000 or out_sample_rate != 16000):
        raise ValueError("other sample rates need dynamic_axes=True: the static export sizes its frames from the input-rate length (:42)")
    return build_audio_metadata(producer="audio_denoiser_onnx_amd", model_name="UL_UNAS", task="denoise", model_family="ul_unas",
                                input_audio_length=input_audio_length, in_sample_rate=in_sample_rate, out_sample_rate=out_sample_rate, model_sample_rate=16000,
                                dynamic_axes=dynamic_axes, nfft=NFFT, window_length=NFFT, hop_length=HOP,
                                window_type="hann", center_pad=True, pad_mode="reflect", use_batch_fold=use_batch_fold,
                                batch_window_seconds=batch_window_seconds, extra={"n_mels": 100, "remove_dc_offset": 0})
