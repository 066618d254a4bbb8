"""ZipEnhancer host side: geometry, the tensor set libade expects for ``model_family = "zipenhancer"``, the checkpoint fold
and the manifest.

The reference wraps ModelScope's ``speech_zipenhancer_ans_multiloss_16k_base`` network (ZipEnhancer/Export_ZipEnhancer.py:949)
and overrides ten of its forwards (:118-355).  The engine (csrc/ade_zipenhancer.hip) consumes the FUSED tensors that wrapper's
constructor registers (:437-664): the grouped mask | phase decoder (:473-550), the fused real | imaginary phase head (:570-575),
per-head [q | k | p] attention rows concatenated with feed_forward1's in-projection (:606-644), Swoosh offsets folded into the
following bias (:446-455), BiasNorm x the two bypasses folded into one scale pair (:649-664), softmaxed down-sampling weights
(:456-460) and the out-combiner's residual scale (:587-592).  ``fuse_state_dict`` performs those folds on a checkpoint-format
state dict whose names follow the attribute paths the reference's constructor reads; it is pinned to that constructor in
tests/test_zipenhancer.py (tools/make_golden_zipenhancer.py).

The LEAF geometry (channel counts, head sizes, kernel sizes) is not in the reference: it lives in the modelscope package and
its configuration.json, both absent here.  ``ZipConfig``'s defaults are the published "base" model as far as the reference
constrains it -- 4 dual-path encoders, the middle two down-sampled by 2 (pad <= 1 frame, :203-205), 4 heads with
query 16 + value 12 = hidden_size 112 (ZipEnhancer/Optimize_ONNX.py:70-71), causal 2 x 3 dilated dense blocks of depth 4
(:408-424), sub-pixel (1, 3) up-sampling and (1, 2) heads (:504-575) -- and they reproduce the model card's 2.04 M parameters.
Every one of them travels in the blob (tensor ``zip_config``), so a different checkpoint geometry needs no code change.
PARITY OF THE LEAF GEOMETRY IS UNPINNED; everything the reference's own code computes is pinned.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, fields
from typing import Dict, List, Mapping, Sequence, Tuple

import numpy as np

from .metadata import build_audio_metadata

SAMPLE_RATE, NFFT, HOP = 16000, 400, 100                     # Export_ZipEnhancer.py:39,47-49
SWOOSH_L_OFFSET, SWOOSH_R_OFFSET = 0.035, 0.313261687         # :450


@dataclass(frozen=True)
class ZipConfig:
    channels: int = 64            # dense_channel == encoder_dim
    heads: int = 4
    query_head_dim: int = 16
    pos_head_dim: int = 4
    value_head_dim: int = 12
    pos_dim: int = 48
    ff_dim: int = 256             # feed_forward2; feed_forward1 = 3/4, feed_forward3 = 5/4 of it (Zipformer2EncoderLayer)
    conv_kernel: int = 15         # depthwise kernel of the two convolution modules (odd)
    down_t1: int = 1
    down_f1: int = 1
    down_t2: int = 2              # encoders[1], encoders[2]
    down_f2: int = 2
    upscale: int = 2              # decoder sub-pixel width factor
    dense_depth: int = 4

    @property
    def ff1(self) -> int:
        return self.ff_dim * 3 // 4

    @property
    def ff3(self) -> int:
        return self.ff_dim * 5 // 4

    @property
    def hidden(self) -> int:      # NonlinAttention hidden_channels
        return self.channels * 3 // 4

    @property
    def attn_dim(self) -> int:
        return self.heads * (2 * self.query_head_dim + self.pos_head_dim)

    @property
    def value_dim(self) -> int:
        return self.heads * self.value_head_dim

    def as_tensor(self) -> np.ndarray:
        return np.array([getattr(self, f.name) for f in fields(self)], np.float32)

    @staticmethod
    def from_tensor(t) -> "ZipConfig":
        v = [int(round(float(x))) for x in np.asarray(t).reshape(-1)]
        return ZipConfig(*v[:len(fields(ZipConfig))])


ENCODERS = 4
DOWNSAMPLED = (1, 2)               # forward order: plain, down, down, plain (Export_ZipEnhancer.py:860-863)


def frames_of(length: int) -> int:
    return length // HOP + 1


def freq_len() -> int:
    """Sub-band count after the stride-2 (1, 3) convolution with padding (0, 1): (201 + 2 - 3) // 2 + 1 = 101 (:673-679)."""
    return (NFFT // 2 + 1 + 2 - 3) // 2 + 1


def layer_tensors(cfg: ZipConfig) -> List[Tuple[str, List[int]]]:
    """(suffix, shape) of one fused Zipformer2 encoder layer, in the order the forward uses them (:143-187)."""
    C, K = cfg.channels, cfg.conv_kernel
    out = [("attn_ff1_w", [cfg.attn_dim + cfg.ff1, C]), ("attn_ff1_b", [cfg.attn_dim + cfg.ff1]),
           ("pos_w", [cfg.heads * cfg.pos_head_dim, cfg.pos_dim]),
           ("ff1_out_w", [C, cfg.ff1]), ("ff1_out_b", [C]),
           ("nonlin_in_w", [3 * cfg.hidden, C]), ("nonlin_in_b", [3 * cfg.hidden]), ("nonlin_out_w", [C, cfg.hidden]), ("nonlin_out_b", [C])]
    for i in (1, 2):
        out += [(f"sa{i}_in_w", [cfg.value_dim, C]), (f"sa{i}_in_b", [cfg.value_dim]), (f"sa{i}_out_w", [C, cfg.value_dim]), (f"sa{i}_out_b", [C])]
    for i in (1, 2):
        out += [(f"conv{i}_in_w", [2 * C, C]), (f"conv{i}_in_b", [2 * C]), (f"conv{i}_dw_w", [C, K]), (f"conv{i}_dw_b", [C]),
                (f"conv{i}_out_w", [C, C]), (f"conv{i}_out_b", [C])]
    for i, d in ((2, cfg.ff_dim), (3, cfg.ff3)):
        out += [(f"ff{i}_in_w", [d, C]), (f"ff{i}_in_b", [d]), (f"ff{i}_out_w", [C, d]), (f"ff{i}_out_b", [C])]
    out += [("bypass_mid", [C]), ("norm_bias", [C]), ("final_norm_scale", [C]), ("final_residual_scale", [C])]
    return out


def blob_tensors(cfg: ZipConfig) -> List[Tuple[str, List[int]]]:
    """(name, shape) of every tensor of a ``zipenhancer`` weight blob."""
    C, r = cfg.channels, cfg.upscale
    out = [("zip_config", [len(fields(ZipConfig))]),
           ("enc_conv1_w", [C, 2]), ("enc_conv1_b", [C]), ("enc_norm1_w", [C]), ("enc_norm1_b", [C]), ("enc_prelu1", [C])]
    for i in range(cfg.dense_depth):
        out += [(f"enc_dense{i}_w", [C, C * (i + 1), 2, 3]), (f"enc_dense{i}_b", [C]), (f"enc_dense{i}_nw", [C]), (f"enc_dense{i}_nb", [C]),
                (f"enc_dense{i}_pr", [C])]
    out += [("enc_conv2_w", [C, C, 1, 3]), ("enc_conv2_b", [C]), ("enc_norm2_w", [C]), ("enc_norm2_b", [C]), ("enc_prelu2", [C])]
    for e in range(ENCODERS):
        for p in ("f", "t"):
            out += [(f"enc{e}_{p}_{s}", shape) for s, shape in layer_tensors(cfg)]
        if e in DOWNSAMPLED:
            out += [(f"enc{e}_down_t_w", [cfg.down_t2]), (f"enc{e}_down_f_w", [cfg.down_f2]), (f"enc{e}_out_scale", [C]), (f"enc{e}_res_scale", [C])]
    for i in range(cfg.dense_depth):
        out += [(f"dec_dense{i}_w", [2 * C, C * (i + 1), 2, 3]), (f"dec_dense{i}_b", [2 * C]), (f"dec_dense{i}_nw", [2 * C]),
                (f"dec_dense{i}_nb", [2 * C]), (f"dec_dense{i}_pr", [2 * C])]
    out += [("dec_up_w", [2 * C * r, C, 1, 3]), ("dec_up_b", [2 * C * r]), ("dec_up_nw", [2 * C]), ("dec_up_nb", [2 * C]), ("dec_up_pr", [2 * C]),
            ("mask_out_w", [1, C, 1, 2]), ("mask_out_b", [1]), ("phase_out_w", [2, C, 1, 2]), ("phase_out_b", [2])]
    return out


# ---- checkpoint-format names: the attribute paths ZipEnhancer.__init__ / forward read (Export_ZipEnhancer.py:437-664, 847-877) ----
def _layer_state_names(prefix: str) -> Dict[str, str]:
    """fused-piece key -> state-dict name under one Zipformer2EncoderLayer."""
    n = {"attn_in_w": "self_attn_weights.in_proj.weight", "attn_in_b": "self_attn_weights.in_proj.bias",
         "pos_w": "self_attn_weights.linear_pos.weight",
         "nonlin_in_w": "nonlin_attention.in_proj.weight", "nonlin_in_b": "nonlin_attention.in_proj.bias",
         "nonlin_out_w": "nonlin_attention.out_proj.weight", "nonlin_out_b": "nonlin_attention.out_proj.bias",
         "bypass_mid": "bypass_mid.bypass_scale", "bypass": "bypass.bypass_scale", "norm_bias": "norm.bias", "norm_log_scale": "norm.log_scale"}
    for i in (1, 2, 3):
        n.update({f"ff{i}_in_w": f"feed_forward{i}.in_proj.weight", f"ff{i}_in_b": f"feed_forward{i}.in_proj.bias",
                  f"ff{i}_out_w": f"feed_forward{i}.out_proj.weight", f"ff{i}_out_b": f"feed_forward{i}.out_proj.bias"})
    for i in (1, 2):
        n.update({f"sa{i}_in_w": f"self_attn{i}.in_proj.weight", f"sa{i}_in_b": f"self_attn{i}.in_proj.bias",
                  f"sa{i}_out_w": f"self_attn{i}.out_proj.weight", f"sa{i}_out_b": f"self_attn{i}.out_proj.bias",
                  f"conv{i}_in_w": f"conv_module{i}.in_proj.weight", f"conv{i}_in_b": f"conv_module{i}.in_proj.bias",
                  f"conv{i}_dw_w": f"conv_module{i}.depthwise_conv.weight", f"conv{i}_dw_b": f"conv_module{i}.depthwise_conv.bias",
                  f"conv{i}_out_w": f"conv_module{i}.out_proj.weight", f"conv{i}_out_b": f"conv_module{i}.out_proj.bias"})
    return {k: prefix + v for k, v in n.items()}


def fuse_state_dict(sd: Mapping[str, np.ndarray], cfg: ZipConfig) -> Dict[str, np.ndarray]:
    """Checkpoint-format state dict (names = modelscope attribute paths as read by the reference) -> blob tensors.
    Every fold cites the constructor line it restates."""
    g = lambda k: np.asarray(sd[k], np.float32)
    C, H, q, p = cfg.channels, cfg.heads, cfg.query_head_dim, cfg.pos_head_dim
    out: Dict[str, np.ndarray] = {"zip_config": cfg.as_tensor()}

    def seq3(dst, src):            # Sequential(Conv2d, InstanceNorm2d, PReLU)   (dense_conv_1 / dense_conv_2, :851-853)
        w = g(src + ".0.weight")
        out[dst[0] + "_w"] = w.reshape(C, 2) if w.shape[1] == 2 and w.shape[2:] == (1, 1) else w
        out[dst[0] + "_b"] = g(src + ".0.bias")
        out[dst[1] + "_w"], out[dst[1] + "_b"], out[dst[2]] = g(src + ".1.weight"), g(src + ".1.bias"), g(src + ".2.weight")
    seq3(("enc_conv1", "enc_norm1", "enc_prelu1"), "dense_encoder.dense_conv_1")
    seq3(("enc_conv2", "enc_norm2", "enc_prelu2"), "dense_encoder.dense_conv_2")
    for i in range(cfg.dense_depth):                                        # layer = [pad, conv, norm, prelu]  (:411-412)
        b = f"dense_encoder.dense_block.dense_block.{i}."
        out[f"enc_dense{i}_w"], out[f"enc_dense{i}_b"] = g(b + "1.weight"), g(b + "1.bias")
        out[f"enc_dense{i}_nw"], out[f"enc_dense{i}_nb"], out[f"enc_dense{i}_pr"] = g(b + "2.weight"), g(b + "2.bias"), g(b + "3.weight")
        m, ph = f"mask_decoder.dense_block.dense_block.{i}.", f"phase_decoder.dense_block.dense_block.{i}."
        for dst, k in (("w", "1.weight"), ("b", "1.bias"), ("nw", "2.weight"), ("nb", "2.bias"), ("pr", "3.weight")):
            out[f"dec_dense{i}_{dst}"] = np.concatenate((g(m + k), g(ph + k)), axis=0)                                  # :486-500
    for dst, k in (("w", "0.conv1.weight"), ("b", "0.conv1.bias"), ("nw", "1.weight"), ("nb", "1.bias"), ("pr", "2.weight")):
        out[f"dec_up_{dst}"] = np.concatenate((g("mask_decoder.mask_conv." + k), g("phase_decoder.phase_conv." + k)), axis=0)   # :536-550
    out["mask_out_w"], out["mask_out_b"] = g("mask_decoder.mask_conv.3.weight"), g("mask_decoder.mask_conv.3.bias")          # :868
    out["phase_out_w"] = np.concatenate((g("phase_decoder.phase_conv_r.weight"), g("phase_decoder.phase_conv_i.weight")), axis=0)   # :570-575
    out["phase_out_b"] = np.concatenate((g("phase_decoder.phase_conv_r.bias"), g("phase_decoder.phase_conv_i.bias")), axis=0)

    for e in range(ENCODERS):
        base = f"TSConformer.encoders.{e}."
        dual = base + ("encoder." if e in DOWNSAMPLED else "")
        for pi, path in enumerate(("f", "t")):
            names = _layer_state_names(f"{dual}{path}_layers.0.")
            t = {k: g(v) for k, v in names.items()}
            pre = f"enc{e}_{path}_"
            # per-head [q | k | p] row blocks (:606-626), then feed_forward1's in-projection underneath (:633-644)
            qw, kw, pw = np.split(t["attn_in_w"], [H * q, 2 * H * q], axis=0)
            qb, kb, pb = np.split(t["attn_in_b"], [H * q, 2 * H * q], axis=0)
            w = np.concatenate((qw.reshape(H, q, C), kw.reshape(H, q, C), pw.reshape(H, p, C)), axis=1).reshape(-1, C)
            b = np.concatenate((qb.reshape(H, q), kb.reshape(H, q), pb.reshape(H, p)), axis=1).reshape(-1)
            out[pre + "attn_ff1_w"] = np.concatenate((w, t["ff1_in_w"]), axis=0)
            out[pre + "attn_ff1_b"] = np.concatenate((b, t["ff1_in_b"]), axis=0)
            out[pre + "pos_w"] = t["pos_w"]
            for i, off in ((1, SWOOSH_L_OFFSET), (2, SWOOSH_L_OFFSET), (3, SWOOSH_L_OFFSET)):      # SwooshL offset folded (:450-455)
                out[pre + f"ff{i}_out_w"] = t[f"ff{i}_out_w"]
                out[pre + f"ff{i}_out_b"] = (t[f"ff{i}_out_b"].astype(np.float64) - off * t[f"ff{i}_out_w"].astype(np.float64).sum(axis=1)).astype(np.float32)
                if i > 1:
                    out[pre + f"ff{i}_in_w"], out[pre + f"ff{i}_in_b"] = t[f"ff{i}_in_w"], t[f"ff{i}_in_b"]
            for k in ("nonlin_in_w", "nonlin_in_b", "nonlin_out_w", "nonlin_out_b", "bypass_mid", "norm_bias"):
                out[pre + k] = t[k]
            for i in (1, 2):
                for k in (f"sa{i}_in_w", f"sa{i}_in_b", f"sa{i}_out_w", f"sa{i}_out_b", f"conv{i}_in_w", f"conv{i}_in_b", f"conv{i}_dw_b"):
                    out[pre + k] = t[k]
                out[pre + f"conv{i}_dw_w"] = t[f"conv{i}_dw_w"].reshape(C, cfg.conv_kernel)
                out[pre + f"conv{i}_out_w"] = t[f"conv{i}_out_w"]                                     # SwooshR offset folded (:450-455)
                out[pre + f"conv{i}_out_b"] = (t[f"conv{i}_out_b"].astype(np.float64)
                                                - SWOOSH_R_OFFSET * t[f"conv{i}_out_w"].astype(np.float64).sum(axis=1)).astype(np.float32)
            outer = g(f"{dual}bypass_layers.{pi}.bypass_scale").astype(np.float64)                    # bypass_layers[2 i + path], one layer per path (:581-586)
            comb = t["bypass"].astype(np.float64) * outer                                             # :649-652
            l2 = np.exp(t["norm_log_scale"].astype(np.float64)) * math.sqrt(C)                        # :653-656
            out[pre + "final_norm_scale"] = (comb * l2).astype(np.float32)                            # :657-660
            out[pre + "final_residual_scale"] = (1.0 - comb).astype(np.float32)                       # :661-664
        if e in DOWNSAMPLED:
            for axis in ("t", "f"):
                bias = g(f"{base}downsample_{axis}.bias")
                ex = np.exp(bias - bias.max())
                out[f"enc{e}_down_{axis}_w"] = (ex / ex.sum(dtype=np.float32)).astype(np.float32)    # bias.softmax(dim=0) (:456-460)
            sc = g(base + "out_combiner.bypass_scale")
            out[f"enc{e}_out_scale"] = sc                                                             # :812
            out[f"enc{e}_res_scale"] = (1.0 - sc.astype(np.float64)).astype(np.float32)               # :587-592
    want = dict(blob_tensors(cfg))
    for k, shape in want.items():
        if tuple(out[k].shape) != tuple(shape):
            raise ValueError(f"fuse_state_dict: {k} has shape {out[k].shape}, expected {tuple(shape)}")
    return {k: np.ascontiguousarray(out[k], np.float32) for k in want}


def config_from_state_dict(sd: Mapping[str, np.ndarray], heads: int = 4, query_head_dim: int = 0) -> ZipConfig:
    """Geometry of a checkpoint-format state dict, read from its tensor shapes.  ``heads`` cannot be inferred from shapes alone (it is 4 in the
    published model, ZipEnhancer/Optimize_ONNX.py:70); ``query_head_dim`` defaults to the split in_proj = heads * (2 q + p), p = linear_pos / heads."""
    L = "TSConformer.encoders.0.f_layers.0."
    C = int(sd["dense_encoder.dense_conv_1.0.weight"].shape[0])
    attn = int(sd[L + "self_attn_weights.in_proj.weight"].shape[0])
    pos_out, pos_dim = (int(x) for x in sd[L + "self_attn_weights.linear_pos.weight"].shape)
    p = pos_out // heads
    q = query_head_dim or (attn // heads - p) // 2
    depth = sum(1 for k in sd if k.startswith("dense_encoder.dense_block.dense_block.") and k.endswith(".1.weight"))
    return ZipConfig(channels=C, heads=heads, query_head_dim=q, pos_head_dim=p,
                     value_head_dim=int(sd[L + "self_attn1.in_proj.weight"].shape[0]) // heads, pos_dim=pos_dim,
                     ff_dim=int(sd[L + "feed_forward2.in_proj.weight"].shape[0]),
                     conv_kernel=int(sd[L + "conv_module1.depthwise_conv.weight"].shape[-1]),
                     down_t2=int(sd["TSConformer.encoders.1.downsample_t.bias"].shape[0]), down_f2=int(sd["TSConformer.encoders.1.downsample_f.bias"].shape[0]),
                     upscale=int(sd["mask_decoder.mask_conv.0.conv1.weight"].shape[0]) // C, dense_depth=depth)


def state_dict_spec(cfg: ZipConfig) -> List[Tuple[str, List[int], float]]:
    """(name, shape, scale) of a checkpoint-format state dict for ``weightgen.materialise``: random-init weights of the
    architecture (no checkpoint is available offline).  Scales keep every activation O(1) through the residual stack;
    ``post_state_dict`` turns the affine scales / softmax logits into sensible positive values."""
    C, K = cfg.channels, cfg.conv_kernel
    spec: List[Tuple[str, List[int], float]] = []

    def seq3(name, cin, kh, kw):
        spec.extend([(name + ".0.weight", [C, cin, kh, kw], 1.2 / math.sqrt(cin * kh * kw)), (name + ".0.bias", [C], 0.1),
                     (name + ".1.weight", [C], 0.4), (name + ".1.bias", [C], 0.2), (name + ".2.weight", [C], 0.25)])
    seq3("dense_encoder.dense_conv_1", 2, 1, 1)
    seq3("dense_encoder.dense_conv_2", C, 1, 3)
    for blk in ("dense_encoder", "mask_decoder", "phase_decoder"):
        for i in range(cfg.dense_depth):
            b = f"{blk}.dense_block.dense_block.{i}."
            cin = C * (i + 1)
            spec += [(b + "1.weight", [C, cin, 2, 3], 1.2 / math.sqrt(cin * 6)), (b + "1.bias", [C], 0.1), (b + "2.weight", [C], 0.4),
                     (b + "2.bias", [C], 0.2), (b + "3.weight", [C], 0.25)]
    for dec, seq in (("mask_decoder", "mask_conv"), ("phase_decoder", "phase_conv")):
        spec += [(f"{dec}.{seq}.0.conv1.weight", [C * cfg.upscale, C, 1, 3], 1.2 / math.sqrt(3 * C)), (f"{dec}.{seq}.0.conv1.bias", [C * cfg.upscale], 0.1),
                 (f"{dec}.{seq}.1.weight", [C], 0.4), (f"{dec}.{seq}.1.bias", [C], 0.2), (f"{dec}.{seq}.2.weight", [C], 0.25)]
    spec += [("mask_decoder.mask_conv.3.weight", [1, C, 1, 2], 1.5 / math.sqrt(2 * C)), ("mask_decoder.mask_conv.3.bias", [1], 0.3),
             ("phase_decoder.phase_conv_r.weight", [1, C, 1, 2], 1.5 / math.sqrt(2 * C)), ("phase_decoder.phase_conv_r.bias", [1], 0.1),
             ("phase_decoder.phase_conv_i.weight", [1, C, 1, 2], 1.5 / math.sqrt(2 * C)), ("phase_decoder.phase_conv_i.bias", [1], 0.1)]
    lin = lambda name, o, i, gain=1.0: [(name + ".weight", [o, i], gain * math.sqrt(3.0 / i)), (name + ".bias", [o], 0.05)]
    for e in range(ENCODERS):
        base = f"TSConformer.encoders.{e}."
        dual = base + ("encoder." if e in DOWNSAMPLED else "")
        for pi, path in enumerate(("f", "t")):
            L = f"{dual}{path}_layers.0."
            spec += lin(L + "self_attn_weights.in_proj", cfg.attn_dim, C, 0.9)
            spec += [(L + "self_attn_weights.linear_pos.weight", [cfg.heads * cfg.pos_head_dim, cfg.pos_dim], 0.6)]
            for i, d in ((1, cfg.ff1), (2, cfg.ff_dim), (3, cfg.ff3)):
                spec += lin(L + f"feed_forward{i}.in_proj", d, C, 2.0) + lin(L + f"feed_forward{i}.out_proj", C, d, 0.5)
            spec += lin(L + "nonlin_attention.in_proj", 3 * cfg.hidden, C, 1.5) + lin(L + "nonlin_attention.out_proj", C, cfg.hidden, 0.6)
            for i in (1, 2):
                spec += lin(L + f"self_attn{i}.in_proj", cfg.value_dim, C) + lin(L + f"self_attn{i}.out_proj", C, cfg.value_dim, 0.5)
                spec += lin(L + f"conv_module{i}.in_proj", 2 * C, C, 1.5)
                spec += [(L + f"conv_module{i}.depthwise_conv.weight", [C, 1, K], 1.5 / math.sqrt(K)), (L + f"conv_module{i}.depthwise_conv.bias", [C], 0.05)]
                spec += lin(L + f"conv_module{i}.out_proj", C, C, 0.5)
            spec += [(L + "bypass_mid.bypass_scale", [C], 0.2), (L + "bypass.bypass_scale", [C], 0.2), (L + "norm.bias", [C], 0.1),
                     (L + "norm.log_scale", [], 0.2), (f"{dual}bypass_layers.{pi}.bypass_scale", [C], 0.2)]
        if e in DOWNSAMPLED:
            spec += [(base + "downsample_t.bias", [cfg.down_t2], 0.5), (base + "downsample_f.bias", [cfg.down_f2], 0.5),
                     (base + "out_combiner.bypass_scale", [C], 0.2)]
    return spec


def post_state_dict(sd: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """Shift the generator's symmetric values where the architecture expects positive ones: bypass scales 0.5 +- 0.2 (their training
    range is [0.2, 1]), instance-norm gains 1 +- 0.4, PReLU slopes 0.25 +- 0.25, the magnitude head's bias 1.5 +- 0.3."""
    out = dict(sd)
    for k, v in sd.items():
        if k.endswith("mask_conv.3.bias"):               # compressed-magnitude head: centre it on 1.5 so that relu(.)^(1/0.3) is O(1..10) like a spectrum
            out[k] = (v + np.float32(1.5)).astype(np.float32)
        if v.ndim != 1:
            continue
        in_dense = "dense_block.dense_block" in k        # [pad, conv, norm, prelu]; elsewhere Sequential(conv, norm, prelu[, conv])
        if k.endswith("bypass_scale"):
            out[k] = (v + np.float32(0.5)).astype(np.float32)
        elif k.endswith(".2.weight" if in_dense else ".1.weight"):
            out[k] = (v + np.float32(1.0)).astype(np.float32)
        elif k.endswith(".3.weight" if in_dense else ".2.weight"):
            out[k] = (v + np.float32(0.25)).astype(np.float32)
    return out


def synthetic_state_dict(cfg: ZipConfig, seed: int = 0) -> Dict[str, np.ndarray]:
    from . import weightgen
    return post_state_dict(weightgen.materialise(state_dict_spec(cfg), seed))


def synthetic_tensors(cfg: ZipConfig, seed: int = 0) -> Dict[str, np.ndarray]:
    """Blob tensors of a random-init model of the architecture (what bench.py --workload zipenhancer runs on)."""
    return fuse_state_dict(synthetic_state_dict(cfg, seed), cfg)


def metadata(input_audio_length: int, use_batch_fold: bool = False, batch_window_seconds: float = 1.5, in_sample_rate: int = SAMPLE_RATE,
             out_sample_rate: int = SAMPLE_RATE, gemm_dtype: str = "f32", dft_tables: str = "reference", dynamic_axes: bool = False) -> Dict[str, str]:
    """Manifest of a static export (Export_ZipEnhancer.py:977-981).  Without batch-fold ``input_audio_length`` must be whole hops (the
    reference's STFT -> ISTFT pair reconstructs (T - 1) * 100 samples; its default export always folds, :58-60).
    ``gemm_dtype``: "f32" (default: exact fp32 matrix-core products, the parity path) | "bf16" (BASELINE.json configs[2]'s dtype: bf16 weights and activations stored in
    HBM for the dense blocks, the feed-forward modules, the projections and the attention / convolution-module operands, fp32 residual stream, statistics, softmax and
    front / back ends -- csrc/ade_zip16.h; gated on its distance from the f32 path and from the reference-run fixture).  ``dft_tables``: "reference" (its fp32-angle tables) | "exact"."""
    # ``dynamic_axes``: the DYNAMIC_AXES export (:31, :61, :828-829, :898-899, :907-908): any input length, scale-factor interpolation on the edges, the ISTFT divides by
    # the overlap-add denominator of the actual frame count; the handle still serves ONE input length.
    if dynamic_axes and use_batch_fold:
        raise ValueError("Batch folding requires static axes.")
    if not dynamic_axes and not use_batch_fold and input_audio_length % HOP and in_sample_rate == SAMPLE_RATE:
        raise ValueError(f"input_audio_length must be a multiple of the hop ({HOP}) without batch-fold")
    return build_audio_metadata(producer="audio_denoiser_onnx_amd", model_name="ZipEnhancer", task="denoise", model_family="zipenhancer",
                                input_audio_length=input_audio_length, in_sample_rate=in_sample_rate, out_sample_rate=out_sample_rate,
                                model_sample_rate=SAMPLE_RATE, nfft=NFFT, window_length=NFFT, hop_length=HOP,
                                window_type="hann", center_pad=True, pad_mode="reflect", use_batch_fold=use_batch_fold,
                                batch_window_seconds=batch_window_seconds, max_dynamic_audio_seconds=2, feature_kind="stft_zipformer", dynamic_axes=dynamic_axes,
                                extra={"n_mels": 100, "ade_gemm_dtype": gemm_dtype, "ade_dft_tables": dft_tables})


def macs_per_window(frames: int, cfg: ZipConfig = ZipConfig()) -> Dict[str, float]:
    """Multiply-accumulates of one window of ``frames`` STFT frames by part (FFT-form STFT excluded: < 0.1 %)."""
    C, F0, F = cfg.channels, NFFT // 2 + 1, freq_len()
    T = frames
    dense_layer = sum(6 * C * C * (i + 1) for i in range(cfg.dense_depth))
    enc = T * F0 * (2 * C + dense_layer) + T * F * 3 * C * C
    dec = T * F * (2 * dense_layer + 2 * 3 * C * C * cfg.upscale) + T * (F * cfg.upscale - 1) * 2 * C * 3
    per_tok = C * (cfg.attn_dim + cfg.ff1) + cfg.ff1 * C + C * 3 * cfg.hidden + cfg.hidden * C + 2 * 2 * C * cfg.value_dim + \
        2 * (2 * C * C + C * cfg.conv_kernel + C * C) + 2 * C * cfg.ff_dim + 2 * C * cfg.ff3

    def layer(nseq, n):            # projections + scores (q k, p pos) + three weighted sums (head-0 nonlin, two value paths)
        attn = nseq * cfg.heads * n * (n * cfg.query_head_dim + (2 * n - 1) * cfg.pos_head_dim) + nseq * n * n * (cfg.hidden + 2 * cfg.value_dim)
        return nseq * n * per_tok + attn
    dt, df = -(-T // cfg.down_t2), -(-F // cfg.down_f2)
    tf = 2 * (layer(T, F) + layer(F, T)) + 2 * (layer(dt, df) + layer(df, dt))
    return {"dense_encoder": float(enc), "encoders": float(tf), "decoders": float(dec), "total": float(enc + tf + dec)}
