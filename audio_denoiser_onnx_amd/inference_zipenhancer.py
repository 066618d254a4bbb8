#!/usr/bin/env python3
"""The reference's ``ZipEnhancer/Inference_ZipEnhancer_ONNX.py`` call surface on the MI355X engine.

    python -m audio_denoiser_onnx_amd.inference_zipenhancer <model_dir_or_.adew> [noisy_16k.wav] [denoised.wav]

Same life-cycle as the reference driver (:268-352): open the session, validate the metadata, read the file as mono int16 (the int16 amplitudes go
to the model as they are -- its per-window RMS normalisation expects [-32768, 32767], :203-206), cut static slices at a stride of the INPUT length
(:296), zero-pad the tail (:297-305), run, concatenate, trim to ``int(round(n * INPUT_TO_OUTPUT_SCALE))`` output samples (:308), write PCM_16.  The
reference makes one ORT call per slice; here ALL slices of the file go to the GPU as one batch (a slice is an independent call of the graph: its
own fold windows, each with its own RMS norm factor).
"""
from __future__ import annotations

import sys
import time
from pathlib import Path

from .inference_gtcrn import denoise, example_audio, normalise_audio, plan_slices, read_wav_int16, write_wav_int16
from .metadata import runtime_config_from_metadata
from .session import InferenceSession


def main(argv=None) -> int:
    argv = [a for a in (sys.argv[1:] if argv is None else argv) if not a.startswith("--")]
    if not argv:
        print(__doc__)
        return 2
    here = Path(__file__).resolve().parent
    noisy = Path(argv[1]) if len(argv) > 1 else example_audio("denoise", "speech_with_noise1.wav")      # Example_Audio.py registry entry "zipenhancer"
    out_path = Path(argv[2]) if len(argv) > 2 else here / "denoised_zipenhancer.wav"
    from .distributed import init_from_env, shutdown
    rank, world, local = init_from_env()
    session = InferenceSession(argv[0], device_id=local)
    if session.metadata.metadata.get("model_family") != "zipenhancer":
        raise ValueError("this driver expects a model_family=zipenhancer manifest")
    cfg = runtime_config_from_metadata(session.metadata)
    print(f"\nUsable Providers: {session.get_providers()}\n\nTest Input Audio: {noisy}")
    audio = normalise_audio(read_wav_int16(noisy, cfg["IN_SAMPLE_RATE"]), cfg["NORMALIZE_AUDIO"], cfg["NORMALIZE_TARGET_RMS"])
    print("\nRunning the ZipEnhancer on the MI355X engine.")
    session.reserve(plan_slices(len(audio), session.in_len, session.out_len, out_stride=False)[1])
    t0 = time.time()
    denoised = denoise(session, audio, tail_pad="zeros", family="dfsmn", rank=rank, world=world)       # input-length stride, rounded output length: the same rules as the DFSMN driver
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
