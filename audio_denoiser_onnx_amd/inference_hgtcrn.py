#!/usr/bin/env python3
"""The reference's ``H-GTCRN/Inference_H_GTCRN_ONNX.py`` call surface on the MI355X engine.

    python -m audio_denoiser_onnx_amd.inference_hgtcrn <model_dir_or_.adew> [noisy_stereo_16k.wav] [denoised.wav]

Life-cycle of the reference driver (:266-384): open the session, validate the metadata, load the file with the graph's channel count
as (1, 2, n) int16, cut it into static slices of the graph's input length, pad the tail -- zeros when the graph folds windows, else a
REFLECTION of the signal's own end (``pad_audio_tail_with_context`` :138-151; deterministic, unlike the Gaussian tails of the other
stereo driver) -- run, concatenate the mono outputs, trim to the input length, write PCM_16.  The reference makes one ORT call per slice;
here ALL slices of the file go to the GPU as one batch (each slice is an independent call of the graph: its own DC mean, its own WPE /
AuxIVA statistics).
"""
from __future__ import annotations

import sys
import time
from pathlib import Path

import numpy as np

from .inference_gtcrn import example_audio, normalise_audio, output_length, session_rates, plan_slices
from .inference_melband import load_stereo
from .metadata import runtime_config_from_metadata
from .session import InferenceSession
from .wavio import write_pcm16


def pad_tail(audio: np.ndarray, target: int, fold_active: bool) -> np.ndarray:
    """(C, n) -> (C, target): zeros in fold mode (:158-166), otherwise the reflect-context padding (:138-151)."""
    n = audio.shape[-1]
    if n >= target:
        return audio
    pad = target - n
    if fold_active or n == 0:
        block = np.zeros((audio.shape[0], pad), audio.dtype)
    elif n == 1:
        block = np.repeat(audio[:, -1:], pad, axis=-1)
    else:
        block = np.pad(audio, ((0, 0), (0, pad)), mode="reflect")[:, n:]
    return np.concatenate((audio, block.astype(audio.dtype, copy=False)), axis=-1)


def cut_slices(audio: np.ndarray, in_len: int, fold_active: bool, out_len: int = 0, rates_equal: bool = True) -> np.ndarray:
    """(2, n) -> (n_slices, 2, in_len).  Stride = the graph's OUTPUT length when it is a hop-truncated copy of the input
    length at the same rate, else the input length.  (The reference's H-GTCRN driver takes the output-length stride
    whenever the two lengths differ, :341-343, without the GTCRN driver's IN == OUT rate condition; with resampling
    edges that would skip input samples, so the GTCRN driver's condition is applied here.)"""
    stride, n_slices, total = plan_slices(audio.shape[1], in_len, out_len or in_len, out_stride=rates_equal)
    audio = pad_tail(audio, total, fold_active)
    idx = np.arange(n_slices)[:, None] * stride + np.arange(in_len)[None, :]
    return np.ascontiguousarray(audio[:, idx].transpose(1, 0, 2))


def denoise(session: InferenceSession, audio: np.ndarray, fold_active: bool, rank: int = 0, world: int = 1, group=None) -> np.ndarray:
    """(2, n) int16 -> int16 mono of the input's duration at the OUTPUT rate (``int(n * OUT / IN)`` samples, :352):
    every slice of the file in one batched call."""
    in_rate, out_rate = session_rates(session)
    # A dynamic-length export returns MORE than its input's duration (the ISTFT keeps the last frame's tail: in_len + 256 samples at equal rates).  The reference
    # driver sees no static output length for such a graph, so it steps by the INPUT length (:341-343 only take the output stride when both shapes are integers) and
    # binds an output of round(in_len * OUT / IN) samples per slice (:350): the tail never reaches the stitched file.
    meta = getattr(session, "metadata", None)
    dynamic = bool(meta and meta.optional_bool("dynamic_axes", False))
    ratio = out_rate / in_rate if in_rate > 0 and out_rate > 0 else 1.0
    keep = min(session.out_len, int(round(session.in_len * ratio))) if dynamic else session.out_len
    slices = cut_slices(audio, session.in_len, fold_active, session.out_len, in_rate == out_rate and not dynamic)
    from .distributed import run_rows
    out = run_rows(session, slices, rank, world, group)[0]                                  # (n_slices, 1, out_len)
    n_out = output_length(audio.shape[1], in_rate, out_rate)
    return np.ascontiguousarray(out.reshape(out.shape[0], -1)[:, :keep].reshape(-1)[:n_out])


def main(argv=None) -> int:
    argv = [a for a in (sys.argv[1:] if argv is None else argv) if not a.startswith("--")]
    if not argv:
        print(__doc__)
        return 2
    here = Path(__file__).resolve().parent
    noisy = Path(argv[1]) if len(argv) > 1 else example_audio("denoise", "h_gtcrn_noisy.wav")
    out_path = Path(argv[2]) if len(argv) > 2 else here / "denoised_hgtcrn.wav"
    from .distributed import init_from_env, shutdown
    rank, world, local = init_from_env()
    session = InferenceSession(argv[0], device_id=local)
    if session.metadata.metadata.get("model_family") != "h_gtcrn":
        raise ValueError("this driver expects a model_family=h_gtcrn manifest")
    cfg = runtime_config_from_metadata(session.metadata)
    fold_active = bool(session.metadata.optional_bool("use_batch_fold", False))
    print(f"\nUsable Providers: {session.get_providers()}\n\nTest Input Audio: {noisy}")
    audio = load_stereo(noisy, cfg["IN_SAMPLE_RATE"], session.channels)
    audio = normalise_audio(audio, cfg["NORMALIZE_AUDIO"], cfg["NORMALIZE_TARGET_RMS"])
    print("\nRunning the H-GTCRN on the MI355X engine.")
    session.reserve(max(1, -(-audio.shape[1] // session.in_len)))
    t0 = time.time()
    denoised = denoise(session, audio, fold_active, rank, world)
    elapsed = time.time() - t0
    shutdown()
    if rank != 0:
        return 0
    write_pcm16(out_path, denoised[None], cfg["OUT_SAMPLE_RATE"])
    duration = denoised.shape[0] / cfg["OUT_SAMPLE_RATE"]
    print(f"\nDenoise Process Complete.\n\nSaving to: {out_path}.\n\nReal-Time Factor (RTF): {elapsed / duration:.6f}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
