#!/usr/bin/env python3
"""The reference's ``GTCRN/Inference_GTCRN_ONNX.py`` call surface on the MI355X engine.

    python -m audio_denoiser_onnx_amd.inference_gtcrn <model_dir_or_.adew> [noisy.wav] [denoised.wav] [--sequential | --stream FRAMES]

Same life-cycle as the reference script (Inference_GTCRN_ONNX.py:237-344): open the session, load + validate the
metadata, read the wav as mono int16 at IN_SAMPLE_RATE, optional RMS normalisation, cut fixed-length slices (stride =
the model's OUTPUT length when it differs from the input length, :287-290), zero-pad the tail (:291-298), run, concat,
trim to the original length, write PCM_16, print the RTF.

What differs is only the hot loop: the reference runs one ``run_with_iobinding`` per slice (:326-330); slices carry
no state from one to the next, so here ALL slices of the file go through the engine as ONE batch (``--sequential``
reproduces the one-slice-per-call loop for comparison).  With ``torch.distributed`` initialised, slices are sharded
across ranks and the outputs stitched with an all-gather (distributed.py).  ``--stream FRAMES`` instead runs the whole file through one
stateful stream (``ade_stream_*``) in pushes of FRAMES hops: no slice edges at all.
"""
from __future__ import annotations

import sys
import time
import wave
from pathlib import Path
from typing import Optional, Tuple

import numpy as np

from .distributed import shard_bounds, stitch_rows
from .metadata import runtime_config_from_metadata
from .session import InferenceSession

NORMALIZE_TARGET_RMS = 4096.0


def example_audio(task: str, name: str) -> Path:
    """Default test clip of a driver, like the reference's ``Example_Audio.py`` registry: ``<ADE_TEST_EXAMPLES or
    ./Test_Examples>/<task>/<name>``.  The clips belong to the reference checkout; nothing of it is bundled here, so a
    missing file is a plain ``FileNotFoundError`` asking for an explicit path."""
    import os
    root = Path(os.environ.get("ADE_TEST_EXAMPLES", "Test_Examples"))
    p = root / task / name
    if not p.exists():
        raise FileNotFoundError(f"default example clip {p} not found: pass the wav path explicitly or set ADE_TEST_EXAMPLES "
                                f"to the reference's Test_Examples directory")
    return p


def read_wav_int16(path, sample_rate: int) -> np.ndarray:
    """Mono int16 at ``sample_rate`` — what ``AudioSegment.from_file(..).set_channels(1).set_frame_rate(sr)`` yields
    (Inference_GTCRN_ONNX.py:272).  Multi-channel files are averaged; other rates are not resampled here."""
    with wave.open(str(path), "rb") as w:
        if w.getsampwidth() != 2:
            raise ValueError(f"{path}: only 16-bit PCM wav is supported, got {8 * w.getsampwidth()}-bit")
        sr, ch = w.getframerate(), w.getnchannels()
        data = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")
    if ch > 1:
        data = (data.reshape(-1, ch).astype(np.int32).sum(axis=1) // ch).astype(np.int16)
    if sr != sample_rate:
        raise NotImplementedError(f"{path}: sample rate {sr} != model input rate {sample_rate} (resampling is not implemented)")
    return np.ascontiguousarray(data, dtype=np.int16)


def write_wav_int16(path, pcm: np.ndarray, sample_rate: int) -> None:
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(sample_rate)
        w.writeframes(np.ascontiguousarray(pcm, dtype="<i2").tobytes())


def write_wav_float32(path, x: np.ndarray, sample_rate: int) -> None:
    """Mono IEEE-float wav: what ``sf.write(..., subtype='FLOAT')`` produces for a float model output (Inference_GTCRN_ONNX.py:340)."""
    import struct
    data = np.ascontiguousarray(x, dtype="<f4").tobytes()
    fmt = struct.pack("<HHIIHH", 3, 1, int(sample_rate), int(sample_rate) * 4, 4, 32)          # WAVE_FORMAT_IEEE_FLOAT, mono
    fact = struct.pack("<I", len(data) // 4)
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"fact" + struct.pack("<I", 4) + fact + b"data" + struct.pack("<I", len(data)) + data
    with open(str(path), "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", len(body)) + body)


def normalise_audio(audio: np.ndarray, enable: bool, target_rms: float = NORMALIZE_TARGET_RMS) -> np.ndarray:
    """Optional RMS normalisation to ``target_rms`` with int16 clipping (Inference_GTCRN_ONNX.py:115-135)."""
    if not enable:
        return audio
    x = audio.astype(np.float32)
    rms = np.sqrt(np.mean(x * x, dtype=np.float32), dtype=np.float32)
    if rms > 0.0:
        x *= target_rms / (rms + 1e-7)
    np.clip(x, -32768.0, 32767.0, out=x)
    return x.astype(np.int16)


def plan_slices(audio_len: int, in_len: int, out_len: int, out_stride: bool = True) -> Tuple[int, int, int]:
    """(stride, number of slices, zero-padded length) exactly as Inference_GTCRN_ONNX.py:287-299 computes them.
    ``out_stride``: whether a graph whose output is shorter than its input is stepped by the OUTPUT length.  The GTCRN
    driver does that only when IN_SAMPLE_RATE == OUT_SAMPLE_RATE (:289: a hop-truncated output, not a resampled one);
    the DFSMN / MossFormer2 drivers always step by the input length (DFSMN/Inference_DFSMN_ONNX.py:289-291)."""
    stride = in_len
    if audio_len > in_len:
        if in_len != out_len and out_stride:
            stride = out_len
        n = int(np.ceil((audio_len - in_len) / stride)) + 1
        return stride, n, (n - 1) * stride + in_len
    return stride, 1, in_len


def output_length(audio_len: int, in_rate: int, out_rate: int, rounded: bool = False) -> int:
    """Length the concatenated output is trimmed to, in OUTPUT samples: ``int(audio_len * OUT / IN)`` in the GTCRN /
    H-GTCRN drivers (Inference_GTCRN_ONNX.py:303), ``int(round(audio_len * INPUT_TO_OUTPUT_SCALE))`` in the DFSMN /
    MossFormer2 drivers (DFSMN/Inference_DFSMN_ONNX.py:312)."""
    if in_rate == out_rate or in_rate <= 0:
        return int(audio_len)
    x = audio_len * (float(out_rate) / float(in_rate)) if rounded else audio_len * out_rate / in_rate
    return int(round(x)) if rounded else int(x)


def session_rates(session) -> Tuple[int, int]:
    """(IN_SAMPLE_RATE, OUT_SAMPLE_RATE) of a session; objects without the attributes count as equal-rate."""
    return int(getattr(session, "in_sample_rate", 0)), int(getattr(session, "out_sample_rate", 0))


def cut_slices(audio: np.ndarray, in_len: int, out_len: int, tail_pad: str = "zeros", rng=None,
               out_stride: bool = True) -> Tuple[np.ndarray, int]:
    """Slices exactly as the reference drivers cut them.  ``tail_pad``: ``"zeros"`` (GTCRN / ZipEnhancer,
    Inference_GTCRN_ONNX.py:291-299) or ``"noise"`` -- Gaussian noise scaled to the RMS of the last ``pad`` samples (of
    the whole file when it is shorter than one slice), the reference's policy for DFSMN / Mel-Band / MossFormer2 when
    batch-fold is inactive (DFSMN/Inference_DFSMN_ONNX.py:292-305).  The reference draws that noise unseeded; pass a
    ``numpy.random.Generator`` as ``rng`` for a reproducible run (it only affects the last, partial slice)."""
    stride, n, total = plan_slices(len(audio), in_len, out_len, out_stride)
    padded = np.zeros(total, np.int16)
    padded[: len(audio)] = audio
    pad = total - len(audio)
    if tail_pad == "noise" and pad > 0 and len(audio) > 0:
        ref = (audio[-pad:] if len(audio) > in_len else audio).astype(np.float32)
        rng = rng or np.random.default_rng()
        padded[len(audio):] = (np.sqrt(np.mean(ref * ref)) * rng.normal(0.0, 1.0, pad)).astype(np.int16)
    elif tail_pad not in ("zeros", "noise"):
        raise ValueError(f"tail_pad must be 'zeros' or 'noise', got {tail_pad!r}")
    idx = np.arange(n)[:, None] * stride + np.arange(in_len)[None, :]
    return padded[idx], stride


PIPELINE_BATCH = 256      # rows per submission of a file that is larger than one batch (BASELINE's batch: one 1 s chunk per CU)


def process_rows(session, rows: np.ndarray, batch: int = PIPELINE_BATCH) -> np.ndarray:
    """All slices of a file through the engine.  Up to ``batch`` rows: ONE ``process`` call.  More: the reference's loop over its slices
    (Inference_GTCRN_ONNX.py:314-333) as a pipeline of ``batch``-row submissions (``ade_submit`` / ``ade_wait``, three in flight): the copy-in of
    batch k + 1 and the copy-out of batch k - 1 run under the kernels of batch k.  Same bits either way (rows are independent calls)."""
    n = len(rows)
    if n <= batch or not hasattr(session, "submit"):
        return session.process(rows)[0]
    rows = np.ascontiguousarray(rows, dtype=np.int16)
    out = np.empty((n, session.row_out), np.int16)
    tickets = []
    for i in range(0, n, batch):
        if len(tickets) >= 3:
            session.wait(tickets.pop(0))
        tickets.append(session.submit(rows[i:i + batch], out[i:i + batch]))
    for t in tickets:
        session.wait(t)
    return out


def denoise(session: InferenceSession, audio: np.ndarray, sequential: bool = False, rank: int = 0, world: int = 1,
            group=None, tail_pad: str = "zeros", rng=None, family: str = "gtcrn") -> np.ndarray:
    """int16 mono waveform in -> int16 denoised waveform out: the input's duration at the OUTPUT sample rate.
    ``family`` selects the reference driver whose stride / trim rules apply (``"gtcrn"`` or ``"dfsmn"``).
    A model whose audio tensors are float (input_audio_dtype / output_audio_dtype F32 or F16) gets the int16 samples cast straight to its input dtype and
    returns its output dtype, as the reference driver does (Inference_GTCRN_ONNX.py:133-135, 336-340)."""
    in_rate, out_rate = session_rates(session)
    in_dt, out_dt = getattr(session, "in_dtype", np.int16), getattr(session, "out_dtype", np.int16)
    float_io = in_dt != np.int16 or out_dt != np.int16
    dfsmn = family == "dfsmn"
    audio_len = output_length(len(audio), in_rate, out_rate, rounded=dfsmn)
    # A dynamic-length export returns MORE than its input's duration (the ISTFT keeps the last frame's tail).  The reference driver binds an output of
    # round(INPUT_AUDIO_LENGTH * scale) samples for such a model (Inference_GTCRN_ONNX.py:300-304), i.e. it only ever keeps that many: slices are stepped by the
    # input length and each slice's output is cut there before the stitch.
    dynamic = bool(getattr(session, "metadata", None) and session.metadata.optional_bool("dynamic_axes", False))
    # (a session-like object without rate attributes counts as equal-rate: session_rates() returns 0 for it)
    ratio = out_rate / in_rate if in_rate > 0 and out_rate > 0 else 1.0
    keep = min(session.out_len, int(round(session.in_len * ratio))) if dynamic else session.out_len
    slices, _ = cut_slices(audio, session.in_len, session.out_len, tail_pad, rng,
                           out_stride=(not dfsmn) and in_rate == out_rate and not dynamic)
    if world > 1 and not sequential and hasattr(session, "run_device") and not float_io and not dynamic:
        from .distributed import sharded_run
        return sharded_run(session, slices, world, rank, group).reshape(-1)[:audio_len]      # device block -> all-gather -> one D2H
    lo, hi = shard_bounds(len(slices), world, rank)
    mine = slices[lo:hi]
    if float_io:
        name = session.get_inputs()[0].name
        local = (session.run(None, {name: mine.astype(in_dt)[:, None, :]})[0][:, 0] if len(mine) else np.zeros((0, session.out_len), out_dt))
        if out_dt == np.float16:
            local = local.astype(np.float32)                                                      # (:336-337)
    elif sequential:
        outs = [session.run(None, {"noisy_audio": s.reshape(1, 1, -1)})[0].reshape(1, -1) for s in mine]
        local = np.concatenate(outs, axis=0) if outs else np.zeros((0, session.out_len), np.int16)
    else:
        local = process_rows(session, mine)
    full = stitch_rows(local, len(slices), world, rank, group) if world > 1 else local
    return np.ascontiguousarray(full[:, :keep]).reshape(-1)[:audio_len]            # np.concatenate(saved).reshape(-1)[:audio_len]  (:332)


def denoise_streaming(session: InferenceSession, audio: np.ndarray, frames_per_push: int = 62) -> np.ndarray:
    """int16 mono waveform -> int16 denoised waveform of the same length through ONE stateful stream (``--stream N``): no slice edges --
    the result is what the reference's graph would give on the whole file in one call (without its whole-call DC removal), which its
    static export cannot do for files longer than the graph input.  The file is zero-padded to whole pushes; the stream's one-hop latency
    is removed again (pushes + flush, first hop dropped)."""
    from .session import StreamingSession
    P = frames_per_push * 256
    n = max(1, -(-len(audio) // P))
    padded = np.zeros(n * P, np.int16)
    padded[:len(audio)] = audio
    with StreamingSession(session, 1, frames_per_push) as st:
        parts = [st.push(padded[None, i * P:(i + 1) * P]) for i in range(n)]
        parts.append(st.flush())
    return np.ascontiguousarray(np.concatenate(parts, axis=1)[0, 256:256 + len(audio)])


def main(argv=None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    sequential = "--sequential" in argv
    stream_frames = 0
    if "--stream" in argv:
        i = argv.index("--stream")
        stream_frames = int(argv[i + 1])
        del argv[i:i + 2]
    argv = [a for a in argv if not a.startswith("--")]
    if not argv:
        print(__doc__)
        return 2
    model = argv[0]
    here = Path(__file__).resolve().parent
    noisy = Path(argv[1]) if len(argv) > 1 else example_audio("denoise", "gtcrn_mix.wav")
    out_path = Path(argv[2]) if len(argv) > 2 else here / "denoised.wav"

    from .distributed import init_from_env, shutdown
    rank, world, local = init_from_env()                 # torchrun: one process per GPU; the file's slices are dealt in contiguous blocks, one all-gather stitches them
    session = InferenceSession(model, device_id=local)
    cfg = runtime_config_from_metadata(session.metadata)
    print(f"\nUsable Providers: {session.get_providers()}")
    print(f"\nTest Input Audio: {noisy}")
    audio = read_wav_int16(noisy, cfg["IN_SAMPLE_RATE"])
    audio = normalise_audio(audio, cfg["NORMALIZE_AUDIO"], cfg["NORMALIZE_TARGET_RMS"])
    print("\nRunning the GTCRN on the MI355X engine.")
    session.reserve(plan_slices(len(audio), session.in_len, session.out_len, cfg["IN_SAMPLE_RATE"] == cfg["OUT_SAMPLE_RATE"])[1])
    t0 = time.time()
    denoised = denoise_streaming(session, audio, stream_frames) if stream_frames else denoise(session, audio, sequential=sequential, rank=rank, world=world)
    elapsed = time.time() - t0
    shutdown()
    if rank != 0:
        return 0
    print("Complete: 100.00%")
    (write_wav_int16 if denoised.dtype == np.int16 else write_wav_float32)(out_path, denoised, cfg["OUT_SAMPLE_RATE"])      # PCM_16 / FLOAT (:340)
    duration = len(denoised) / cfg["OUT_SAMPLE_RATE"] if cfg["OUT_SAMPLE_RATE"] > 0 else 0.0
    rtf = elapsed / duration if duration > 0 else float("inf")
    print(f"\nDenoise Process Complete.\n\nSaving to: {out_path}.\n\nReal-Time Factor (RTF): {rtf:.6f}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
