#!/usr/bin/env python3
"""The reference's ``MossFormer2_SS_16K/Inference_MossFormer_SS_ONNX.py`` call surface on the MI355X engine.

    python -m audio_denoiser_onnx_amd.inference_mossformer <model_dir_or_.adew> [mixture_16k.wav] [out_prefix] [--seed N]

Life-cycle of the reference driver (:206-357): open the session, validate the metadata, load the file as mono int16, PREPEND
``pad_head`` zeros (:275, metadata key ``pad_head`` = 8000: the model's first half second is a warm-up that is cut again after
separation :337-338), cut into static slices of the graph's input length, pad the tail (zeros under batch-fold, else Gaussian
noise at the tail's RMS, unseeded there; ``--seed`` here), run, concatenate each speaker's slices, drop the head padding and trim
to the input length, write two PCM_16 wavs.  All slices go to the GPU as ONE batch (each is an independent call of the graph).
"""
from __future__ import annotations

import sys
import time
from pathlib import Path

import numpy as np

from .inference_gtcrn import example_audio, normalise_audio, output_length, session_rates, read_wav_int16
from .metadata import runtime_config_from_metadata
from .session import InferenceSession
from .wavio import write_pcm16


def cut_slices(audio: np.ndarray, in_len: int, fold_active: bool, rng=None) -> np.ndarray:
    """(n,) -> (n_slices, in_len), stride in_len; tail policy of the reference (:289-306)."""
    n = len(audio)
    n_slices = max(1, -(-n // in_len))
    pad = n_slices * in_len - n
    if pad:
        if fold_active:
            block = np.zeros(pad, audio.dtype)
        else:
            tail = (audio[-pad:] if n > in_len else audio).astype(np.float32)
            rms = np.sqrt(np.mean(tail * tail, dtype=np.float32), dtype=np.float32)
            rng = rng or np.random.default_rng()
            block = (rms * rng.normal(0.0, 1.0, size=pad)).astype(audio.dtype)
        audio = np.concatenate((audio, block))
    return np.ascontiguousarray(audio.reshape(n_slices, in_len))


def separate(session: InferenceSession, audio: np.ndarray, pad_head: int, fold_active: bool, rng=None, rank: int = 0, world: int = 1, group=None):
    """mono int16 (n,) -> [speaker_0 (n,), speaker_1 (n,)] int16: head padding, one batched call, head drop + trim (:275, :337-338)."""
    padded = np.concatenate((np.zeros(pad_head, audio.dtype), audio))
    slices = cut_slices(padded, session.in_len, fold_active, rng)
    from .distributed import run_rows
    outs = run_rows(session, slices[:, None, :], rank, world, group)                     # world > 1: each rank runs a block of slices, one all-gather (both outputs)
    in_rate, out_rate = session_rates(session)
    head_out = output_length(pad_head, in_rate, out_rate, rounded=True)          # pad_head_out, out_audio_len (:308-309)
    end_out = output_length(len(padded), in_rate, out_rate, rounded=True)
    return [np.ascontiguousarray(o.reshape(-1)[head_out:end_out]) for o in outs]


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
    mix = Path(argv[1]) if len(argv) > 1 else example_audio("separation", "mixed_speech.wav")
    prefix = Path(argv[2]) if len(argv) > 2 else here / "separated"
    from .distributed import init_from_env, shutdown
    rank, world, local = init_from_env()                                                  # torchrun: one process per GPU (BASELINE configs[4])
    session = InferenceSession(argv[0], device_id=local)
    if session.metadata.metadata.get("model_family") != "mossformer2_ss":
        raise ValueError("this driver expects a model_family=mossformer2_ss manifest")
    cfg = runtime_config_from_metadata(session.metadata)
    pad_head = int(session.metadata.optional_int("pad_head", 8000))
    fold_active = bool(session.metadata.optional_bool("use_batch_fold", False))
    print(f"\nUsable Providers: {session.get_providers()}\n\nTest Input Audio: {mix}")
    audio = normalise_audio(read_wav_int16(mix, cfg["IN_SAMPLE_RATE"]), cfg["NORMALIZE_AUDIO"], cfg["NORMALIZE_TARGET_RMS"])
    print("\nRunning the MossFormer_SS on the MI355X engine.")
    session.reserve(max(1, -(-(len(audio) + pad_head) // session.in_len)))
    t0 = time.time()
    spk = separate(session, audio, pad_head, fold_active, np.random.default_rng(seed), rank, world)
    elapsed = time.time() - t0
    shutdown()
    if rank != 0:
        return 0
    paths = [Path(f"{prefix}_{i}.wav") for i in range(len(spk))]
    for p, x in zip(paths, spk):
        write_pcm16(p, x, cfg["OUT_SAMPLE_RATE"])
    duration = len(spk[0]) / cfg["OUT_SAMPLE_RATE"]
    print(f"\nDenoise Process Complete.\n\nSaving to: {' & '.join(str(p) for p in paths)}.\n\nReal-Time Factor (RTF): {elapsed / duration:.6f}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
