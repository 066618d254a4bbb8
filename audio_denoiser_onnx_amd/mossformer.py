"""MossFormer2-SS-16K host side: the tensor set libade expects for ``model_family = "mossformer2_ss"`` and its manifest.

The engine (csrc/ade_mossformer.hip) consumes the FUSED buffers the reference's export wrapper registers
(MossFormer2_SS_16K/Export_MossFormer2_SS_16K.py:130-395: ``encoder_w``, ``front_w/b``, per layer ``fl_in_{w,b,c}_i``,
``qkos_{gamma,beta}_i``, ``fl_out_{w,b,c}_i``, ``fs_*_i`` ..., ``mm_norm_*``, ``intra_norm_*``, ``tail_gate_{w,b}``,
``mask_decoder_w``, ``decoder_w``) under their registered names, the position tables it derives from the frame count
(``emb_pos`` :156-162, ``rot_cos`` / ``rot_signed_sin`` :196-205; rebuilt here), and the scalar attributes of the wrapper
(eps values, slopes, FLASH / FSMN geometry) as one small tensor ``hyper`` plus ``fs_front_alpha``.
"""
from __future__ import annotations

from typing import Dict, Mapping

import numpy as np

from . import weightgen
from .metadata import build_audio_metadata

SAMPLE_RATE, ENC_KERNEL, ENC_STRIDE, MODEL_DIM = 16000, 16, 8, 512
# order of the ``hyper`` tensor (csrc/ade_mossformer.hip reads it by index)
HYPER_KEYS = ("norm_factor", "flash_group_size", "rot_dim", "dw_pad", "fl_norm_eps", "fl_out_norm_eps", "front_norm_eps", "mm_norm_eps",
              "intra_norm_eps", "fs_ln_eps", "fs_n1_eps", "fs_n2_eps", "fs_mem_depth", "tail_prelu_alpha", "fs_mem_norm_eps0", "fs_mem_lorder")


def frames_of(window: int) -> int:
    return (window - ENC_KERNEL) // ENC_STRIDE + 1


def position_tables(frames: int, rot_dim: int = 32, pos_scale: float = 1.0) -> Dict[str, np.ndarray]:
    """``emb_pos`` (1, 512, n) = ScaledSinuEmbedding (:156-162); ``rot_cos`` / ``rot_signed_sin`` (1, n, 1, rot_dim) (:196-205)."""
    t = np.arange(frames, dtype=np.float32)
    inv_freq = (np.float32(1.0) / (np.float32(10000.0) ** (np.arange(0, MODEL_DIM, 2, dtype=np.float32) / np.float32(MODEL_DIM)))).astype(np.float32)
    sinu = (t[:, None] * inv_freq[None, :]).astype(np.float32)
    emb = np.concatenate((np.sin(sinu), np.cos(sinu)), axis=1).astype(np.float32) * np.float32(pos_scale)
    freqs = (np.float32(1.0) / (np.float32(10000.0) ** (np.arange(0, rot_dim, 2, dtype=np.float32) / np.float32(rot_dim)))).astype(np.float32)
    ang = np.repeat((t[:, None] * freqs[None, :]).astype(np.float32), 2, axis=1)
    sign = np.tile(np.array([-1.0, 1.0], np.float32), rot_dim // 2)
    return {"emb_pos": np.ascontiguousarray(emb.T[None]), "rot_cos": np.cos(ang).astype(np.float32)[None, :, None, :],
            "rot_signed_sin": (np.sin(ang).astype(np.float32) * sign)[None, :, None, :]}


def model_tensors(fused: Mapping[str, np.ndarray], scalars: Mapping, window: int) -> Dict[str, np.ndarray]:
    """``fused`` (registered buffer name -> array) + position tables for ``window`` samples + the scalar attributes."""
    out = {k: np.ascontiguousarray(v, np.float32) for k, v in fused.items()}
    out.update(position_tables(frames_of(window), int(scalars["rot_dim"]), float(scalars.get("pos_scale", 1.0))))
    pads, dils = list(scalars["fs_mem_paddings"]), list(scalars["fs_mem_dilations"])
    lorder = pads[0] + 1                                           # padding_j = lorder + (dil_j - 1)(lorder - 1) - 1, dil_0 = 1 (:283-287)
    for j, (p, d) in enumerate(zip(pads, dils)):
        if d != 2 ** j or p != lorder + (d - 1) * (lorder - 1) - 1:
            raise ValueError("dilated memory geometry must be dilation 2^j with the reference's symmetric padding")
    if len(set(scalars["fs_mem_norm_eps"])) != 1:
        raise ValueError("all memory InstanceNorms must share one eps")
    hyper = dict(scalars, fs_mem_norm_eps0=scalars["fs_mem_norm_eps"][0], fs_mem_lorder=lorder)
    out["hyper"] = np.array([float(hyper[k]) for k in HYPER_KEYS], np.float32)
    out["fs_front_alpha"] = np.asarray(scalars["fs_front_alpha"], np.float32)
    return out


def metadata(input_audio_length: int, use_batch_fold: bool = False, batch_window_seconds: float = 1.5, in_sample_rate: int = SAMPLE_RATE,
             out_sample_rate: int = SAMPLE_RATE, gemm_dtype: str = "f32", dynamic_axes: bool = False) -> Dict[str, str]:
    """Manifest keys the reference stamps for this model (:712-718): two output sources, conv encoder/decoder features.
    ``input_audio_length`` counts INPUT-rate samples; with in / out rates other than 16 kHz the engine interpolates linearly on both
    edges like the export (:562-571, :625-640) and the model sees round(length * 16000 / in_rate) samples.
    ``gemm_dtype``: "f32" only (the parity path and BASELINE.json's dtype for this model); the engine refuses other values.
    ``dynamic_axes``: the DYNAMIC_AXES export (:24): the edges interpolate by scale factor (floor(length * 16000 / in_rate) model-rate samples), and the weights carry the
    linear keys' OffsetScale row WITHOUT the 1 / frames factor, which the graph applies at run time (:183, :430, :500-501) -- ``synthetic_tensor(..., fold_inv_n=False)``."""
    if dynamic_axes and use_batch_fold:
        raise ValueError("Batch folding requires a static shape (DYNAMIC_AXES = False)")                   # (:97)
    meta = build_audio_metadata(producer="audio_denoiser_onnx_amd", model_name="MossFormer2_SS_16K", task="source_separation",
                                model_family="mossformer2_ss", input_audio_length=input_audio_length, in_sample_rate=in_sample_rate,
                                out_sample_rate=out_sample_rate, model_sample_rate=SAMPLE_RATE,
                                nfft=ENC_KERNEL, window_length=ENC_KERNEL, hop_length=ENC_STRIDE, window_type="none", center_pad=False,
                                pad_mode="none", use_batch_fold=use_batch_fold, batch_window_seconds=batch_window_seconds, dynamic_axes=dynamic_axes,
                                max_dynamic_audio_seconds=6,
                                feature_kind="conv_encoder_decoder", extra={"pad_head": 8000, "enc_stride": ENC_STRIDE, "output_sources": 2,
                                                                             "ade_gemm_dtype": gemm_dtype})
    return meta


def synthetic_tensor(name: str, shape, scale: float, frames: int, group_size: int = 256, fold_inv_n: bool = True) -> np.ndarray:
    """Random-init value of one fused buffer (``weightgen.tensor`` + the constraints that keep the stack well conditioned):
    norm gains positive and away from zero, PReLU slopes positive, and the two factors the reference folds into OffsetScale
    -- 1 / group_size into the quadratic-query row, 1 / frames into the linear-key row (:236-241) -- applied to those rows."""
    v = weightgen.tensor(name, shape, scale)
    if name.startswith(("fs_mem_norm_w_", "fs_n1_w_", "fs_n2_w_")) or name in ("mm_norm_w", "intra_norm_w"):
        v = np.abs(v) + np.float32(0.3)
    if name.startswith("fs_mem_prelu_"):
        v = np.abs(v)
    if name.startswith(("qkos_gamma_", "qkos_beta_")):
        v = v.copy()
        v[0] *= np.float32(1.0 / group_size)
        if fold_inv_n:                                  # (the DYNAMIC_AXES export leaves this row alone and scales the reduced product instead, :183)
            v[3] *= np.float32(1.0 / frames)
    return v


# stand-in scalar attributes of the published MossFormer2_SS_16K geometry (what MOSSFORMER_SS.__init__ derives from the clearvoice network)
DEFAULT_SCALARS = {
    "norm_factor": 10.0 ** (-25.0 / 20.0), "flash_group_size": 256, "rot_dim": 32, "dw_pad": 8, "fl_norm_eps": 1e-5 / 512 ** -0.5,
    "fl_out_norm_eps": 1e-5 / 1024 ** -0.5, "front_norm_eps": 1e-8, "mm_norm_eps": 1e-8, "intra_norm_eps": 1e-8, "fs_ln_eps": 1e-5, "fs_n1_eps": 1e-8,
    "fs_n2_eps": 1e-8, "fs_mem_depth": 2, "tail_prelu_alpha": 0.25, "fs_mem_paddings": [19, 38], "fs_mem_dilations": [1, 2],
    "fs_mem_norm_eps": [1e-5, 1e-5],
}


def synthetic_spec(layers: int):
    """(name, shape, scale) of every fused buffer for random-init weights of the architecture (tools/bench_mossformer.py)."""
    def lin(fan_in):
        return 1.7 / fan_in ** 0.5
    spec = [("encoder_w", [512, 1, 16], 0.8), ("decoder_w", [512, 1, 16], 0.3), ("mask_decoder_w", [512, 512, 1], lin(512)),
            ("mm_norm_w", [512], 1.2), ("mm_norm_b", [512], 0.05), ("intra_norm_w", [512], 1.2), ("intra_norm_b", [512], 0.05),
            ("front_w", [512, 512, 1], lin(512)), ("front_b", [512], 0.05), ("tail_gate_w", [2048, 512, 1], lin(512)), ("tail_gate_b", [2048], 0.05)]
    for i in range(layers):
        spec += [(f"fl_in_w_{i}", [2176, 512], 2.0 * 1.7 * 22.6 / 512 ** 0.5), (f"fl_in_b_{i}", [2176], 0.05), (f"fl_in_c_{i}", [2176, 1, 17], 0.15),
                 (f"fl_out_w_{i}", [512, 1024], 1.7 * 32.0 / 1024 ** 0.5), (f"fl_out_b_{i}", [512], 0.05), (f"fl_out_c_{i}", [512, 1, 17], 0.15),
                 (f"qkos_gamma_{i}", [4, 128], 0.6), (f"qkos_beta_{i}", [4, 128], 0.05),
                 (f"fs_uv_w_{i}", [512, 256], lin(256)), (f"fs_uv_b_{i}", [512], 0.05), (f"fs_uv_c_{i}", [512, 1, 17], 0.15),
                 (f"fs_mem_w_{i}_0", [256, 1, 39], 0.15), (f"fs_mem_norm_w_{i}_0", [256], 1.2), (f"fs_mem_norm_b_{i}_0", [256], 0.05), (f"fs_mem_prelu_{i}_0", [256], 0.3),
                 (f"fs_mem_w_{i}_1", [256, 2, 39], 0.15), (f"fs_mem_norm_w_{i}_1", [256], 1.2), (f"fs_mem_norm_b_{i}_1", [256], 0.05), (f"fs_mem_prelu_{i}_1", [256], 0.3),
                 (f"fs_mem_linear_w_{i}", [256, 256], lin(256)), (f"fs_mem_linear_b_{i}", [256], 0.05), (f"fs_mem_project_w_{i}", [256, 256], lin(256)),
                 (f"fs_front_w_{i}", [256, 512], lin(512)), (f"fs_front_b_{i}", [256], 0.05), (f"fs_back_w_{i}", [512, 256], lin(256)), (f"fs_back_b_{i}", [512], 0.05),
                 (f"fs_n1_w_{i}", [256], 1.2), (f"fs_n1_b_{i}", [256], 0.05), (f"fs_n2_w_{i}", [256], 1.2), (f"fs_n2_b_{i}", [256], 0.05)]
    return spec


def flops_per_window(frames: int, layers: int, group: int = 256) -> float:
    """Multiply-add flops (2 per MAC) of the matrix products of one window: encoder / decoder, per layer the FLASH projections, the
    quadratic attention inside groups, the linear attention, the FSMN Linears; the speaker tail."""
    padded = -(-frames // group) * group
    per_layer = 2 * frames * (512 * 2176 + 1024 * 512 + 512 * 256 + 256 * 512 + 2 * 256 * 256 + 256 * 512)
    per_layer += 2 * (padded // group) * (group * group * 128 + group * group * 2048) + 2 * 128 * 2048 * frames * 2
    tail = 2 * frames * (16 * 512 + 512 * 512 + 512 * 2048 + 2 * 512 * 512 + 2 * 512 * 16)
    return float(layers * per_layer + tail)


def fuse_checkpoint(state: Mapping[str, np.ndarray], frames: int, scalars: Mapping = None, fold_inv_n: bool = True):
    """clearvoice ``MossFormer2_SS_16K`` ``state_dict`` -> (fused buffers, scalar attributes) for ``frames`` encoder frames per window.
    Restates the fold algebra of the reference's export wrapper (Export_MossFormer2_SS_16K.py:130-395) in float64 with one rounding
    to fp32, as the reference does:
      * the front GroupNorm affine folds into the 1x1 conv that consumes it (:222-228);
      * to_hidden | to_qk share one ScaleNorm: their Linears stack, the scalar gains g / scale fold into the weights, their
        depthwise kernels stack (:236-262); the to_out ScaleNorm gain folds into its Linear (:250);
      * 1 / group_size folds into the quadratic-query OffsetScale row and 1 / frames into the linear-key row (:251-256);
      * to_u | to_v share one affine-free LayerNorm: each branch's LayerNorm affine folds into its Linear, then they stack (:300-310);
      * conv1d_out folds into output | output_gate per speaker (:367-389).
    ``scalars``: the attributes that are not parameters (eps values, group size, memory order); default = the published geometry
    (DEFAULT_SCALARS).  Pinned against the reference's own constructor by tests/test_mossformer.py::test_checkpoint_fusion_matches_reference."""
    g = {k: np.asarray(v, np.float64) for k, v in state.items()}
    sc = dict(DEFAULT_SCALARS if scalars is None else scalars)
    group = int(sc["flash_group_size"])
    out: Dict[str, np.ndarray] = {}
    mn = "mask_net."
    out["encoder_w"], out["decoder_w"], out["mask_decoder_w"] = g["enc.conv1d.weight"], g["dec.weight"], g[mn + "conv1_decoder.weight"]
    out["mm_norm_w"], out["mm_norm_b"] = g[mn + "mdl.intra_mdl.norm.weight"], g[mn + "mdl.intra_mdl.norm.bias"]
    out["intra_norm_w"], out["intra_norm_b"] = g[mn + "mdl.intra_norm.weight"], g[mn + "mdl.intra_norm.bias"]
    fw = g[mn + "conv1d_encoder.weight"]
    out["front_w"] = fw * g[mn + "norm.weight"][None, :, None]
    out["front_b"] = fw[:, :, 0] @ g[mn + "norm.bias"] + (g[mn + "conv1d_encoder.bias"] if mn + "conv1d_encoder.bias" in g else 0.0)
    lp = mn + "mdl.intra_mdl.mossformerM."
    layers = 0
    while f"{lp}layers.{layers}.to_hidden.mdl.1.weight" in g:
        layers += 1
    alphas = []
    for i in range(layers):
        fl = f"{lp}layers.{i}."
        dim = g[fl + "to_hidden.mdl.1.weight"].shape[1]
        in_fold, out_fold = float(dim) ** 0.5, float(g[fl + "to_out.mdl.1.weight"].shape[1]) ** 0.5          # 1 / ScaleNorm.scale, scale = dim^-1/2
        out[f"fl_in_w_{i}"] = np.concatenate((g[fl + "to_hidden.mdl.1.weight"] * g[fl + "to_hidden.mdl.0.g"] * in_fold,
                                              g[fl + "to_qk.mdl.1.weight"] * g[fl + "to_qk.mdl.0.g"] * in_fold), axis=0)
        out[f"fl_in_b_{i}"] = np.concatenate((g[fl + "to_hidden.mdl.1.bias"], g[fl + "to_qk.mdl.1.bias"]))
        out[f"fl_in_c_{i}"] = np.concatenate((g[fl + "to_hidden.mdl.3.sequential.1.conv.weight"], g[fl + "to_qk.mdl.3.sequential.1.conv.weight"]), axis=0)
        out[f"fl_out_w_{i}"] = g[fl + "to_out.mdl.1.weight"] * g[fl + "to_out.mdl.0.g"] * out_fold
        out[f"fl_out_b_{i}"] = g[fl + "to_out.mdl.1.bias"]
        out[f"fl_out_c_{i}"] = g[fl + "to_out.mdl.3.sequential.1.conv.weight"]
        row = np.array([1.0 / group, 1.0, 1.0, 1.0 / frames if fold_inv_n else 1.0])[:, None]          # (fold_lin_inv_n = not DYNAMIC_AXES, :183, :252-253)
        out[f"qkos_gamma_{i}"] = g[fl + "qk_offset_scale.gamma"] * row
        out[f"qkos_beta_{i}"] = g[fl + "qk_offset_scale.beta"] * row
        fb = f"{lp}fsmn.{i}."
        gf = fb + "gated_fsmn."
        ws, bs, cs = [], [], []
        for br in ("to_u", "to_v"):
            lw, lb = g[f"{gf}{br}.mdl.1.weight"], g[f"{gf}{br}.mdl.1.bias"]
            ws.append(lw * g[f"{gf}{br}.mdl.0.weight"][None, :])
            bs.append(lw @ g[f"{gf}{br}.mdl.0.bias"] + lb)
            cs.append(g[f"{gf}{br}.mdl.3.sequential.1.conv.weight"])
        out[f"fs_uv_w_{i}"], out[f"fs_uv_b_{i}"], out[f"fs_uv_c_{i}"] = np.concatenate(ws), np.concatenate(bs), np.concatenate(cs)
        for j in range(int(sc["fs_mem_depth"])):
            out[f"fs_mem_w_{i}_{j}"] = g[f"{gf}fsmn.conv.conv{j + 1}.weight"][..., 0]
            out[f"fs_mem_norm_w_{i}_{j}"], out[f"fs_mem_norm_b_{i}_{j}"] = g[f"{gf}fsmn.conv.norm{j + 1}.weight"], g[f"{gf}fsmn.conv.norm{j + 1}.bias"]
            out[f"fs_mem_prelu_{i}_{j}"] = g[f"{gf}fsmn.conv.prelu{j + 1}.weight"]
        out[f"fs_mem_linear_w_{i}"], out[f"fs_mem_linear_b_{i}"] = g[gf + "fsmn.linear.weight"], g[gf + "fsmn.linear.bias"]
        out[f"fs_mem_project_w_{i}"] = g[gf + "fsmn.project.weight"]
        out[f"fs_front_w_{i}"], out[f"fs_front_b_{i}"] = g[fb + "conv1.0.weight"][..., 0], g[fb + "conv1.0.bias"]
        out[f"fs_back_w_{i}"], out[f"fs_back_b_{i}"] = g[fb + "conv2.weight"][..., 0], g[fb + "conv2.bias"]
        out[f"fs_n1_w_{i}"], out[f"fs_n1_b_{i}"] = g[fb + "norm1.weight"], g[fb + "norm1.bias"]
        out[f"fs_n2_w_{i}"], out[f"fs_n2_b_{i}"] = g[fb + "norm2.weight"], g[fb + "norm2.bias"]
        alphas.append(float(g[fb + "conv1.1.weight"].reshape(-1)[0]))
    gate_w = np.concatenate((g[mn + "output.0.weight"], g[mn + "output_gate.0.weight"]), axis=0)[..., 0]
    gate_b = np.concatenate((g[mn + "output.0.bias"], g[mn + "output_gate.0.bias"]))
    cw, cb = g[mn + "conv1d_out.weight"][..., 0], g[mn + "conv1d_out.bias"]
    ch = gate_w.shape[1]
    tw, tb = [], []
    for spk in range(cw.shape[0] // ch):
        tw.append(gate_w @ cw[spk * ch:(spk + 1) * ch])
        tb.append(gate_w @ cb[spk * ch:(spk + 1) * ch] + gate_b)
    out["tail_gate_w"], out["tail_gate_b"] = np.concatenate(tw)[..., None], np.concatenate(tb)
    sc["tail_prelu_alpha"] = float(g[mn + "prelu.weight"].reshape(-1)[0])
    sc["fs_front_alpha"] = alphas
    sc["pos_scale"] = float(g[mn + "pos_enc.scale"].reshape(-1)[0]) if mn + "pos_enc.scale" in g else 1.0
    return {k: np.ascontiguousarray(v, np.float32) for k, v in out.items()}, sc
