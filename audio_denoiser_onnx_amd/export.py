"""Checkpoint -> engine model files (``<name>.adew`` + ``<name>_Metadata.json``).

Plays the role of the reference's ``GTCRN/Export_GTCRN.py:742-789`` minus the ONNX step: take the
checkpoint-format ``state_dict`` (``torch.load(..)['model']``), fold every BatchNorm into its conv exactly as
``ConvBlock.fuse_bn_`` / ``GTConvBlock.fuse_bn_`` do (Export_GTCRN.py:171-194, 244-267), pre-transpose the ERB
matrices (``ERB.prepare_for_export_`` :109-114), and write the tensors under the reference's own names.

    python -m audio_denoiser_onnx_amd.export <checkpoint.tar|state_dict.npz> <out_dir> [--length 16000] [--dynamic] [--in-rate 48000] [--out-rate 8000]
                                                                                       [--in-dtype F32] [--out-dtype F32]
    python -m audio_denoiser_onnx_amd.export --family mel_band_roformer <MelBandRoformer.ckpt> <out_dir> [--length 66150] [--fold] [--dynamic [--in-rate 48000] [--out-rate 48000]]
    python -m audio_denoiser_onnx_amd.export --family mossformer2_ss <checkpoint> <out_dir> [--length 24000] [--fold] [--dynamic [--in-rate 8000] [--out-rate 48000]]
    python -m audio_denoiser_onnx_amd.export --family ul_unas <model_trained_on_dns3.tar> <out_dir> [--length 16000]
    python -m audio_denoiser_onnx_amd.export --family zipenhancer <pytorch_model.bin> <out_dir> [--length 32000] [--fold]

The other two families fold their checkpoints the way their export constructors do (``melband.fuse_checkpoint`` =
Export_MelBandRoformer.py:455-538; ``mossformer.fuse_checkpoint`` = Export_MossFormer2_SS_16K.py:130-395); both folds are
pinned against the reference's own constructors (tests/test_melband.py, tests/test_mossformer.py).
"""
from __future__ import annotations

import sys
from collections import OrderedDict
from pathlib import Path
from typing import Dict, Mapping

import numpy as np

from .metadata import build_audio_metadata, write_metadata
from .weights import save_blob

BN_EPS = 1e-5   # nn.BatchNorm2d default, as constructed at Export_GTCRN.py:167,214,221,225

# (conv, bn) pairs of the two block flavours; transposed = nn.ConvTranspose2d (decoder)
_CONVBLOCKS = [("encoder.en_convs.0.", False, 1), ("encoder.en_convs.1.", False, 2),
               ("decoder.de_convs.3.", True, 2), ("decoder.de_convs.4.", True, 1)]
_GTBLOCKS = [(f"encoder.en_convs.{i}.", False) for i in (2, 3, 4)] + [(f"decoder.de_convs.{i}.", True) for i in (0, 1, 2)]


def _fold(w: np.ndarray, b, gamma, beta, mean, var, transposed: bool, groups: int):
    """BN(conv(x)) == conv'(x): scale per OUTPUT channel; ConvTranspose2d weights are (Cin, Cout/groups, kh, kw)."""
    scale = (gamma / np.sqrt(var + BN_EPS)).astype(np.float32)
    if transposed:
        cin, og = w.shape[0], w.shape[1]
        wv = w.reshape(groups, cin // groups, og, w.shape[2], w.shape[3])
        w2 = (wv * scale.reshape(groups, 1, og, 1, 1)).reshape(w.shape)
    else:
        w2 = w * scale.reshape(-1, 1, 1, 1)
    b2 = beta - mean * scale if b is None else (b - mean) * scale + beta
    return w2.astype(np.float32), b2.astype(np.float32)


def fold_gtcrn_state_dict(sd: Mapping[str, np.ndarray]) -> "OrderedDict[str, np.ndarray]":
    """Checkpoint-format GTCRN state_dict (numpy arrays) -> the BN-folded tensor set libade loads."""
    sd = {k: np.asarray(v, dtype=np.float32) for k, v in sd.items() if "num_batches_tracked" not in k}
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()

    def bn(prefix):
        return sd[prefix + "weight"], sd[prefix + "bias"], sd[prefix + "running_mean"], sd[prefix + "running_var"]

    for p, transposed, groups in _CONVBLOCKS:
        w, b = _fold(sd[p + "conv.weight"], sd.get(p + "conv.bias"), *bn(p + "bn."), transposed, groups)
        out[p + "conv.weight"], out[p + "conv.bias"] = w, b
        if p + "act.weight" in sd:
            out[p + "act.weight"] = sd[p + "act.weight"]
    for p, transposed in _GTBLOCKS:
        for conv, bnn, groups in (("point_conv1", "point_bn1", 1), ("depth_conv", "depth_bn", 16), ("point_conv2", "point_bn2", 1)):
            w, b = _fold(sd[p + conv + ".weight"], sd.get(p + conv + ".bias"), *bn(p + bnn + "."), transposed, groups)
            out[p + conv + ".weight"], out[p + conv + ".bias"] = w, b
        out[p + "point_act.weight"] = sd[p + "point_act.weight"]
        out[p + "depth_act.weight"] = sd[p + "depth_act.weight"]
        for leaf in ("tra.att_gru.weight_ih_l0", "tra.att_gru.weight_hh_l0", "tra.att_gru.bias_ih_l0", "tra.att_gru.bias_hh_l0",
                     "tra.att_fc.weight", "tra.att_fc.bias"):
            out[p + leaf] = sd[p + leaf]
    for k, v in sd.items():
        if k.startswith("dpgrnn"):
            out[k] = v
    out["erb.erb_weight_t"] = np.ascontiguousarray(sd["erb.erb_fc.weight"].T)     # (192, 64)
    out["erb.ierb_weight_t"] = np.ascontiguousarray(sd["erb.ierb_fc.weight"].T)   # (64, 192)
    return out


def load_state_dict(path) -> Dict[str, np.ndarray]:
    path = Path(path)
    if path.suffix == ".npz":
        return dict(np.load(path))
    import torch
    ckpt = torch.load(str(path), map_location="cpu", weights_only=False)
    sd = ckpt["model"] if isinstance(ckpt, dict) and "model" in ckpt else ckpt
    return {k: v.detach().cpu().numpy() for k, v in sd.items()}


def export_gtcrn(checkpoint, out_dir, input_audio_length: int = 16000, name: str = "GTCRN", in_sample_rate: int = 16000, out_sample_rate: int = 16000,
                 dynamic_axes: bool = False, input_audio_dtype: str = "INT16", output_audio_dtype: str = "INT16") -> Path:
    """The export's I/O switches (Export_GTCRN.py:26-30, 47-48): other input / output sample rates need ``dynamic_axes`` (the static graph sizes its frame count
    from the input-rate length, :45); float audio tensors are "F32" / "F16"."""
    out_dir = Path(out_dir)
    out_dir.mkdir(parents=True, exist_ok=True)
    model_path = out_dir / f"{name}.adew"
    save_blob(model_path, fold_gtcrn_state_dict(load_state_dict(checkpoint)))
    meta = build_audio_metadata(producer=Path(__file__).name, model_name=name, task="denoise", model_family="gtcrn",
                                input_audio_length=input_audio_length, in_sample_rate=in_sample_rate, out_sample_rate=out_sample_rate, model_sample_rate=16000,
                                dynamic_axes=dynamic_axes, input_audio_dtype=input_audio_dtype, output_audio_dtype=output_audio_dtype, extra={"n_mels": 100})
    write_metadata(model_path, meta)
    return model_path


def export_melband(checkpoint, out_dir, input_audio_length: int = 66150, use_batch_fold: bool = False, heads: int = 8, dim_head: int = 64,
                   name: str = "MelBandRoformer", dynamic_axes: bool = False, in_sample_rate: int = 44100, out_sample_rate: int = 44100) -> Path:
    """Upstream Mel-Band-Roformer ``.ckpt`` -> ``<name>.adew`` + manifest (the role of Export_MelBandRoformer.py:684-737 minus ONNX).
    ``dynamic_axes`` / other sample rates: the reference's DYNAMIC_AXES export (:33, :50-53); the engine serves ``input_audio_length`` per handle."""
    from . import melband
    out_dir = Path(out_dir)
    out_dir.mkdir(parents=True, exist_ok=True)
    model_path = out_dir / f"{name}.adew"
    sd = {k: v for k, v in load_state_dict(checkpoint).items() if not k.endswith("rotary_embed.freqs")}
    save_blob(model_path, melband.model_tensors(melband.fuse_checkpoint(sd, heads=heads, dim_head=dim_head)))
    write_metadata(model_path, melband.metadata(input_audio_length, use_batch_fold=use_batch_fold, dynamic_axes=dynamic_axes, in_sample_rate=in_sample_rate,
                                                out_sample_rate=out_sample_rate))
    return model_path


def export_mossformer(checkpoint, out_dir, input_audio_length: int = 24000, use_batch_fold: bool = False, name: str = "MossFormer2_SS_16K", dynamic_axes: bool = False,
                      in_sample_rate: int = 16000, out_sample_rate: int = 16000) -> Path:
    """clearvoice ``MossFormer2_SS_16K`` checkpoint -> ``<name>.adew`` + manifest (Export_MossFormer2_SS_16K.py:672-720 minus ONNX).
    ``dynamic_axes``: the DYNAMIC_AXES export (:24): scale-factor edges, 1 / frames left out of the fused OffsetScale row and applied at run time (:183)."""
    from . import mossformer
    out_dir = Path(out_dir)
    out_dir.mkdir(parents=True, exist_ok=True)
    model_path = out_dir / f"{name}.adew"
    meta = mossformer.metadata(input_audio_length, use_batch_fold=use_batch_fold, dynamic_axes=dynamic_axes, in_sample_rate=in_sample_rate, out_sample_rate=out_sample_rate)
    model_len = int(input_audio_length * (16000 / in_sample_rate)) if dynamic_axes else int(round(input_audio_length * 16000 / in_sample_rate))      # (:36; floor for scale_factor)
    window = int(meta["fold_window_length"]) if use_batch_fold else model_len
    sd = {k[len("module."):] if k.startswith("module.") else k: v for k, v in load_state_dict(checkpoint).items()}
    fused, scalars = mossformer.fuse_checkpoint(sd, mossformer.frames_of(window), fold_inv_n=not dynamic_axes)
    save_blob(model_path, mossformer.model_tensors(fused, scalars, window))
    write_metadata(model_path, meta)
    return model_path


def export_ulunas(checkpoint, out_dir, input_audio_length: int = 16000, name: str = "UL_UNAS", dynamic_axes: bool = False, in_sample_rate: int = 16000,
                  out_sample_rate: int = 16000) -> Path:
    """UL-UNAS checkpoint (``ckpt['model']``, upstream or optimised key names) -> ``<name>.adew`` + manifest (Export_UL_UNAS.py:959-962 minus ONNX).
    ``dynamic_axes`` / other sample rates: the reference's DYNAMIC_AXES export (:26, :41-43); the engine serves ``input_audio_length`` per handle."""
    from . import ulunas
    out_dir = Path(out_dir)
    out_dir.mkdir(parents=True, exist_ok=True)
    model_path = out_dir / f"{name}.adew"
    save_blob(model_path, ulunas.fold_state_dict(ulunas.convert_state_dict(load_state_dict(checkpoint))))
    write_metadata(model_path, ulunas.metadata(input_audio_length, dynamic_axes=dynamic_axes, in_sample_rate=in_sample_rate, out_sample_rate=out_sample_rate))
    return model_path


def export_hgtcrn(checkpoint, out_dir, input_audio_length: int = 32000, use_batch_fold: bool = False, name: str = "H_GTCRN", dynamic_axes: bool = False,
                  in_sample_rate: int = 16000, out_sample_rate: int = 16000) -> Path:
    """H-GTCRN checkpoint (``ckpt['model']`` of GTCRN_IVA) -> ``<name>.adew`` + manifest (Export_H_GTCRN.py:1119-1186 minus ONNX).  ``dynamic_axes``: the DYNAMIC_AXES
    export (:27): the ISTFT's dynamic trim, 256 model-rate samples more out than in."""
    from . import hgtcrn
    out_dir = Path(out_dir)
    out_dir.mkdir(parents=True, exist_ok=True)
    model_path = out_dir / f"{name}.adew"
    save_blob(model_path, hgtcrn.fold_state_dict(load_state_dict(checkpoint)))
    write_metadata(model_path, hgtcrn.metadata(input_audio_length, use_batch_fold, in_sample_rate=in_sample_rate, out_sample_rate=out_sample_rate, dynamic_axes=dynamic_axes))
    return model_path


def export_zipenhancer(checkpoint, out_dir, input_audio_length: int = 32000, use_batch_fold: bool = True, name: str = "ZipEnhancer", dynamic_axes: bool = False,
                       in_sample_rate: int = 16000, out_sample_rate: int = 16000) -> Path:
    """ModelScope ``speech_zipenhancer_ans_multiloss_16k_base`` state dict -> ``<name>.adew`` + manifest (the constructor folds of
    ZipEnhancer/Export_ZipEnhancer.py:437-664 minus ONNX; the reference's default export folds 1.5 s windows, :57-60).  The geometry is read from the
    tensor shapes; wrapper prefixes (``module.``, ``model.``, ``generator.``) are dropped; training-only tensors (balancers, whiteners) are ignored."""
    from . import zipenhancer as zp
    out_dir = Path(out_dir)
    out_dir.mkdir(parents=True, exist_ok=True)
    model_path = out_dir / f"{name}.adew"
    sd = {}
    for k, v in load_state_dict(checkpoint).items():
        for pre in ("module.", "model.", "generator."):
            if k.startswith(pre):
                k = k[len(pre):]
        sd[k] = v
    save_blob(model_path, zp.fuse_state_dict(sd, zp.config_from_state_dict(sd)))
    write_metadata(model_path, zp.metadata(input_audio_length, use_batch_fold=use_batch_fold and not dynamic_axes, dynamic_axes=dynamic_axes,    # (DYNAMIC_AXES, :31)
                                           in_sample_rate=in_sample_rate, out_sample_rate=out_sample_rate))
    return model_path


def main(argv=None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    length, family, fold = None, "gtcrn", False
    if "--length" in argv:
        i = argv.index("--length")
        length = int(argv[i + 1])
        del argv[i:i + 2]
    if "--family" in argv:
        i = argv.index("--family")
        family = argv[i + 1]
        del argv[i:i + 2]
    if "--fold" in argv:
        argv.remove("--fold")
        fold = True
    gt = {"dynamic_axes": False, "in_sample_rate": None, "out_sample_rate": None, "input_audio_dtype": "INT16", "output_audio_dtype": "INT16"}     # GTCRN's / Mel-Band's I/O switches
    if "--dynamic" in argv:
        argv.remove("--dynamic")
        gt["dynamic_axes"] = True
    for flag, key, conv in (("--in-rate", "in_sample_rate", int), ("--out-rate", "out_sample_rate", int), ("--in-dtype", "input_audio_dtype", str.upper),
                            ("--out-dtype", "output_audio_dtype", str.upper)):
        if flag in argv:
            i = argv.index(flag)
            gt[key] = conv(argv[i + 1])
            del argv[i:i + 2]
    if len(argv) != 2 or family not in ("gtcrn", "h_gtcrn", "mel_band_roformer", "mossformer2_ss", "ul_unas", "zipenhancer"):
        print(__doc__)
        return 2
    model_rate = 44100 if family == "mel_band_roformer" else 16000
    gt["in_sample_rate"] = gt["in_sample_rate"] or model_rate
    gt["out_sample_rate"] = gt["out_sample_rate"] or model_rate
    if family == "mel_band_roformer":
        path = export_melband(argv[0], argv[1], length or 66150, fold, dynamic_axes=gt["dynamic_axes"], in_sample_rate=gt["in_sample_rate"],
                              out_sample_rate=gt["out_sample_rate"])
    elif family == "mossformer2_ss":
        path = export_mossformer(argv[0], argv[1], length or 24000, fold, dynamic_axes=gt["dynamic_axes"], in_sample_rate=gt["in_sample_rate"] or 16000,
                                 out_sample_rate=gt["out_sample_rate"] or 16000)
    elif family == "ul_unas":
        path = export_ulunas(argv[0], argv[1], length or 16000, dynamic_axes=gt["dynamic_axes"], in_sample_rate=gt["in_sample_rate"], out_sample_rate=gt["out_sample_rate"])
    elif family == "h_gtcrn":
        path = export_hgtcrn(argv[0], argv[1], length or 32000, fold, dynamic_axes=gt["dynamic_axes"], in_sample_rate=gt["in_sample_rate"], out_sample_rate=gt["out_sample_rate"])
    elif family == "zipenhancer":
        path = export_zipenhancer(argv[0], argv[1], length or 32000, fold, dynamic_axes=gt["dynamic_axes"], in_sample_rate=gt["in_sample_rate"],
                                  out_sample_rate=gt["out_sample_rate"])
    else:
        path = export_gtcrn(argv[0], argv[1], length or 16000, **gt)
    print(f"Export done: {path} (+ {path.with_name(path.stem + '_Metadata.json').name})")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
