#!/usr/bin/env python3
"""The reference's ``Mel_Band_Roformer/Stereo/Inference_MelBandRoformer_ONNX.py`` call surface on the MI355X engine.

    python -m audio_denoiser_onnx_amd.inference_melband <model_dir_or_.adew> [noisy_44k1.wav] [denoised.wav] [--seed N]

Life-cycle of the reference driver (:206-357): open the session, validate the metadata, load the file as (1, C, n) int16
(mono is duplicated to two channels :270-274, extra channels are dropped / missing ones repeated :281-286), cut it into
static slices of the graph's input length, pad the tail -- zeros when the graph folds windows (:299-300, 307-308), else
Gaussian noise scaled to the RMS of the tail (:301-303, unseeded there; ``--seed`` here) -- run, concatenate per channel,
trim to the input length, write WAVEX PCM_16 (:353).  The reference makes one ORT call per slice; here ALL slices of the file
go to the GPU as one batch (each slice is an independent call of the graph, so the result is the same).
"""
from __future__ import annotations

import sys
import time
from pathlib import Path

import numpy as np

from .inference_gtcrn import example_audio, normalise_audio
from .metadata import runtime_config_from_metadata
from .session import InferenceSession
from .wavio import read_pcm16, write_pcm16


def load_stereo(path, sample_rate: int, channels: int = 2) -> np.ndarray:
    """int16 (channels, n): mono duplicated, surplus channels dropped, missing ones repeated from the last (:270-286)."""
    pcm, sr = read_pcm16(path)
    if sr != sample_rate:
        raise NotImplementedError(f"{path}: sample rate {sr} != model input rate {sample_rate} (resampling is not implemented)")
    if pcm.shape[0] == 1:
        pcm = np.concatenate((pcm, pcm), axis=0)
    if pcm.shape[0] < channels:
        pcm = np.concatenate((pcm, np.repeat(pcm[-1:], channels - pcm.shape[0], axis=0)), axis=0)
    return np.ascontiguousarray(pcm[:channels])


def cut_slices(audio: np.ndarray, in_len: int, fold_active: bool, rng=None) -> np.ndarray:
    """(C, n) -> (n_slices, C, in_len) with stride in_len; tail policy as above (:291-314)."""
    C, n = audio.shape
    n_slices = max(1, -(-n // in_len))
    pad = n_slices * in_len - n
    if pad:
        if fold_active:
            block = np.zeros((C, pad), audio.dtype)
        else:
            tail = (audio[:, -pad:] if n > in_len else audio).astype(np.float32)
            rms = np.sqrt(np.mean(tail * tail, dtype=np.float32), dtype=np.float32)
            rng = rng or np.random.default_rng()
            block = (rms * rng.normal(0.0, 1.0, size=(C, pad))).astype(audio.dtype)
        audio = np.concatenate((audio, block), axis=1)
    return np.ascontiguousarray(audio.reshape(C, n_slices, in_len).transpose(1, 0, 2))


def denoise(session: InferenceSession, audio: np.ndarray, fold_active: bool, rng=None, rank: int = 0, world: int = 1, group=None) -> np.ndarray:
    """(C, n) int16 -> (C, round(n * out_rate / in_rate)) int16: every slice of the file in one batched call.  A dynamic-length export returns more than its input's
    duration (the ISTFT keeps the last frame's tail); the reference driver binds an output of round(INPUT_AUDIO_LENGTH * scale) samples for it (:322-323), so each
    slice's output is cut there before the stitch."""
    slices = cut_slices(audio, session.in_len, fold_active, rng)
    from .distributed import run_rows
    out = run_rows(session, slices, rank, world, group)[0]                                  # (n_slices, C, out_len); world > 1: each rank runs a block, one all-gather
    in_rate, out_rate = getattr(session, "in_sample_rate", 0), getattr(session, "out_sample_rate", 0)
    scale = out_rate / in_rate if in_rate > 0 and out_rate > 0 else 1.0           # (a session-like object without rates counts as equal-rate)
    keep = min(session.out_len, int(round(session.in_len * scale)))
    out = out[:, :, :keep]
    return np.ascontiguousarray(out.transpose(1, 0, 2).reshape(out.shape[1], -1)[:, :int(round(audio.shape[1] * scale))])


def main(argv=None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    seed = None
    if "--seed" in argv:
        i = argv.index("--seed")
        seed = int(argv[i + 1])
        del argv[i:i + 2]
    argv = [a for a in argv if not a.startswith("--")]
    if not argv:
        print(__doc__)
        return 2
    here = Path(__file__).resolve().parent
    noisy = Path(argv[1]) if len(argv) > 1 else example_audio("denoise", "mel_band_roformer.wav")
    out_path = Path(argv[2]) if len(argv) > 2 else here / "denoised_melband.wav"
    from .distributed import init_from_env, shutdown
    rank, world, local = init_from_env()                                                    # torchrun: one process per GPU, slices dealt in blocks (BASELINE configs[3])
    session = InferenceSession(argv[0], device_id=local)
    if session.metadata.metadata.get("model_family") != "mel_band_roformer":
        raise ValueError("this driver expects a model_family=mel_band_roformer manifest")
    cfg = runtime_config_from_metadata(session.metadata)
    fold_active = bool(session.metadata.optional_bool("use_batch_fold", False))
    print(f"\nUsable Providers: {session.get_providers()}\n\nTest Input Audio: {noisy}")
    audio = load_stereo(noisy, cfg["IN_SAMPLE_RATE"], session.channels)
    audio = normalise_audio(audio, cfg["NORMALIZE_AUDIO"], cfg["NORMALIZE_TARGET_RMS"])
    print("\nRunning the MelBandRoformer on the MI355X engine.")
    session.reserve(max(1, -(-audio.shape[1] // session.in_len)))
    t0 = time.time()
    denoised = denoise(session, audio, fold_active, np.random.default_rng(seed), rank, world)
    elapsed = time.time() - t0
    shutdown()
    if rank != 0:
        return 0
    write_pcm16(out_path, denoised, cfg["OUT_SAMPLE_RATE"], extensible=True)
    duration = denoised.shape[1] / cfg["OUT_SAMPLE_RATE"]
    print(f"\nDenoise Process Complete.\n\nSaving to: {out_path}.\n\nReal-Time Factor (RTF): {elapsed / duration:.6f}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
