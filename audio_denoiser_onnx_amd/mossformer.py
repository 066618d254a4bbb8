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
    out.update(position_tables(frames_of(window), int(scalars["rot_dim"])))
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


def metadata(input_audio_length: int, use_batch_fold: bool = False, batch_window_seconds: float = 1.5) -> Dict[str, str]:
    """Manifest keys the reference stamps for this model (:712-718): two output sources, conv encoder/decoder features."""
    meta = build_audio_metadata(producer="audio_denoiser_onnx_amd", model_name="MossFormer2_SS_16K", task="source_separation",
                                model_family="mossformer2_ss", input_audio_length=input_audio_length, in_sample_rate=SAMPLE_RATE,
                                nfft=ENC_KERNEL, window_length=ENC_KERNEL, hop_length=ENC_STRIDE, window_type="none", center_pad=False,
                                pad_mode="none", use_batch_fold=use_batch_fold, batch_window_seconds=batch_window_seconds,
                                feature_kind="conv_encoder_decoder", extra={"pad_head": 8000, "enc_stride": ENC_STRIDE, "output_sources": 2})
    return meta


def synthetic_tensor(name: str, shape, scale: float, frames: int, group_size: int = 256) -> np.ndarray:
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
        v[3] *= np.float32(1.0 / frames)
    return v
