"""Deterministic synthetic noisy speech for benchmarks and full-size tests.

Per chunk i: seed 1234 + i; a 20-harmonic voiced source (f0 ~ U(90,250) Hz, a_k = 1/k) under a 4 Hz raised-cosine
envelope, peak 0.25 FS, plus white Gaussian noise sigma = 0.05 FS -> x32768 -> round -> clamp -> int16.
"""
from __future__ import annotations

import numpy as np


def synth_chunk(index: int, length: int = 16000, sample_rate: int = 16000) -> np.ndarray:
    rng = np.random.default_rng(1234 + int(index))
    t = np.arange(length, dtype=np.float64) / sample_rate
    f0 = rng.uniform(90.0, 250.0)
    phases = rng.uniform(0.0, 2.0 * np.pi, 20)
    voiced = np.zeros(length)
    for k in range(1, 21):
        if k * f0 < sample_rate / 2:
            voiced += np.sin(2.0 * np.pi * k * f0 * t + phases[k - 1]) / k
    env = 0.5 - 0.5 * np.cos(2.0 * np.pi * 4.0 * t + rng.uniform(0.0, 2.0 * np.pi))
    voiced *= env
    voiced *= 0.25 / max(1e-9, np.abs(voiced).max())
    x = voiced + rng.normal(0.0, 0.05, length)
    return np.clip(np.round(x * 32768.0), -32768, 32767).astype(np.int16)


def synth_batch(batch: int, length: int = 16000, first_index: int = 0, sample_rate: int = 16000) -> np.ndarray:
    return np.stack([synth_chunk(first_index + i, length, sample_rate) for i in range(batch)]) if batch else \
        np.zeros((0, length), np.int16)


def synth_stereo(index: int, length: int, sample_rate: int = 44100) -> np.ndarray:
    """(2, length) int16: right = left delayed by 7 samples + independent noise (the Mel-Band-Roformer workload)."""
    left = synth_chunk(index, length, sample_rate)
    rng = np.random.default_rng(99991 + int(index))
    right = np.roll(left.astype(np.float64), 7) + rng.normal(0.0, 0.02 * 32768.0, length)
    right[:7] = 0.0
    return np.stack((left, np.clip(np.round(right), -32768, 32767).astype(np.int16)))
