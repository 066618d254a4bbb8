#!/usr/bin/env python3
"""The reference's ``DFSMN/Inference_DFSMN_ONNX.py`` call surface on the MI355X engine.

    python -m audio_denoiser_onnx_amd.inference_dfsmn <model_dir_or_.adew> [noisy_48k.wav] [denoised.wav] [--seed N]

Same life-cycle as the GTCRN driver (``inference_gtcrn.py``): open the session, validate the metadata, cut the file into
static slices, run ALL slices as one batch, concatenate, trim.  What differs in the reference's DFSMN driver is the tail
policy when batch-fold is inactive: the last partial slice is padded with Gaussian noise scaled to the RMS of the tail
(DFSMN/Inference_DFSMN_ONNX.py:292-305), unseeded there; ``--seed`` makes it reproducible here.
"""
from __future__ import annotations

import sys
import time
from pathlib import Path

import numpy as np

from .inference_gtcrn import denoise, example_audio, normalise_audio, plan_slices, read_wav_int16, write_wav_int16
from .metadata import runtime_config_from_metadata
from .session import InferenceSession


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
    noisy = Path(argv[1]) if len(argv) > 1 else example_audio("denoise", "speech_with_noise_48k.wav")
    out_path = Path(argv[2]) if len(argv) > 2 else here / "denoised_dfsmn.wav"
    from .distributed import init_from_env, shutdown
    rank, world, local = init_from_env()
    session = InferenceSession(argv[0], device_id=local)
    if session.metadata.metadata.get("model_family") != "dfsmn":
        raise ValueError("this driver expects a model_family=dfsmn manifest")
    cfg = runtime_config_from_metadata(session.metadata)
    print(f"\nUsable Providers: {session.get_providers()}\n\nTest Input Audio: {noisy}")
    audio = normalise_audio(read_wav_int16(noisy, cfg["IN_SAMPLE_RATE"]), cfg["NORMALIZE_AUDIO"], cfg["NORMALIZE_TARGET_RMS"])
    print("\nRunning the DFSMN on the MI355X engine.")
    session.reserve(plan_slices(len(audio), session.in_len, session.out_len, out_stride=False)[1])
    fold_active = bool(session.metadata.optional_bool("use_batch_fold", False))      # zeros under batch-fold (:292-295, :300-302)
    t0 = time.time()
    denoised = denoise(session, audio, tail_pad="zeros" if fold_active else "noise", rng=np.random.default_rng(seed), family="dfsmn", rank=rank, world=world)
    elapsed = time.time() - t0
    shutdown()
    if rank != 0:
        return 0
    write_wav_int16(out_path, denoised, cfg["OUT_SAMPLE_RATE"])
    duration = len(denoised) / cfg["OUT_SAMPLE_RATE"]
    print(f"\nDenoise Process Complete.\n\nSaving to: {out_path}.\n\nReal-Time Factor (RTF): {elapsed / duration:.6f}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
